"""`--gpus N` starts N ranks itself (clip_amd/launch.py): world size, one all-gather, exit status."""
import json
import os
import subprocess
import sys

CHILD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "launch_child.py")


def _run(*args, **extra_env):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(extra_env)
    return subprocess.run([sys.executable, CHILD, *args], env=env, capture_output=True, text=True, timeout=300)


def test_self_launch_two_ranks_merge_equals_global():
    p = _run("2")
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out == {"world": 2, "backend": "gloo", "equal": True}


def test_single_rank_needs_no_process_group():
    p = _run("1")
    assert p.returncode == 0, p.stderr[-2000:]
    assert json.loads(p.stdout.strip().splitlines()[-1]) == {"world": 1, "backend": None, "equal": True}


def test_failing_rank_fails_the_launch():
    p = _run("2", "fail")
    assert p.returncode != 0


def test_rank_that_ignores_sigterm_is_killed_after_the_grace_period():
    import time
    t0 = time.monotonic()
    p = _run("2", "hang", CLIP_AMD_KILL_GRACE_S="2")
    assert p.returncode == 7
    assert time.monotonic() - t0 < 120          # (the hanging rank sleeps 600 s)
