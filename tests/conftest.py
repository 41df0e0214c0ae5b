import ast
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    """tests/golden/<name>.npz -> dict with cfg (BertConfig), rows (list of int32 arrays), ..."""
    from clip_amd.config import BertConfig
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    g = {k: z[k] for k in z.files}
    g["cfg"] = BertConfig(**ast.literal_eval(str(z["cfg"][0])))
    off = np.concatenate(([0], np.cumsum(g["lens"])))
    g["rows"] = [g["ids"][off[i]:off[i + 1]] for i in range(len(g["lens"]))]
    g["seed"], g["n_ctx"], g["batch_size"] = int(g["seed"]), int(g["n_ctx"]), int(g["batch_size"])
    g["weights"] = str(z["weights"][0]) if "weights" in z.files else "procedural"
    return g


def golden_state_dict(g):
    """The weights a golden case was generated with (oracle/gen_golden.py): seeds travel, tensors do not."""
    from clip_amd.weights import procedural_state_dict, trained_like_state_dict
    make = trained_like_state_dict if g["weights"] == "trained_like" else procedural_state_dict
    return make(g["cfg"], g["seed"])


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get
