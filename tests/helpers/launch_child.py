"""Child program of tests/test_launch.py: starts its own ranks (clip_amd.launch), merges a sharded
top-k over the process group and lets rank 0 print one JSON line."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clip_amd import launch, topk  # noqa: E402


def main():
    n = int(sys.argv[1])
    rc = launch.spawn_ranks_if_needed(n)
    if rc is not None:
        sys.exit(rc)
    dist, rank, world, device = launch.init_distributed(timeout_s=120)
    if len(sys.argv) > 2 and sys.argv[2] in ("fail", "hang") and rank == 1:
        sys.exit(7)
    if len(sys.argv) > 2 and sys.argv[2] == "hang" and rank == 0:      # a rank that does not act on SIGTERM
        import signal
        import time
        signal.signal(signal.SIGTERM, signal.SIG_IGN)
        time.sleep(600)
    rng = np.random.default_rng(5)
    doc = np.sort(rng.integers(0, 200, 2000)).astype(np.int64)
    sent = np.arange(2000, dtype=np.int64)
    score = (rng.integers(0, 40, 2000) / 40).astype(np.float32)
    a, b = topk.shard_documents(doc, np.full(2000, 9), world, rank)
    glob = topk.global_topk(topk.local_topk(doc[a:b], sent[a:b], score[a:b], 100), 100)
    if rank == 0:
        ref = topk.local_topk(doc, sent, score, 100)
        print(json.dumps({"world": world, "backend": dist.get_backend() if dist else None,
                          "equal": bool(np.array_equal(glob, ref))}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
