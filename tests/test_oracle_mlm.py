"""Pin the oracle's BertContextMLMSwitch restatement (oracle/bert_oracle.py::mlm_switch_forward) against the
reference class itself (sentclf/bert_model.py:106-186; goldens by oracle/gen_golden.py::make_mlm_case)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers"))
from mlm_case import batches  # noqa: E402
from clip_amd.weights import procedural_state_dict, state_dict_keys  # noqa: E402
from oracle import bert_oracle as O  # noqa: E402


@pytest.mark.parametrize("name", ["tiny_mlm", "base_mlm"])
def test_oracle_mlm_switch_matches_reference(golden, name):
    g = golden(name)
    torch.set_num_threads(8)
    sd = procedural_state_dict(g["cfg"], g["seed"], mlm_head=True)
    assert len(sd) == len(state_dict_keys(g["cfg"])) + 7
    r0 = 0
    for b, (ids, mask, tt, sw, mlm, sl) in enumerate(batches(g)):
        with torch.no_grad():
            loss, logits, scores = O.mlm_switch_forward(sd, g["cfg"], ids, mask, tt, sw, mlm, 102, g["n_ctx"])
        assert abs(float(loss) - g["ref_batch_loss"][b]) < 2e-5
        assert np.abs(logits.numpy() - g["ref_logits"][r0:r0 + len(ids)]).max() < 1e-5
        sc = scores[mlm != -100]
        assert np.abs(torch.logsumexp(sc, -1).numpy() - g["ref_lse"][sl]).max() < 2e-5
        assert np.abs(sc[:, torch.from_numpy(g["sample_cols"])].numpy() - g["ref_sampled"][sl]).max() < 1e-5
        assert np.array_equal(sc.argmax(-1).numpy(), g["ref_argmax"][sl])
        r0 += len(ids)
