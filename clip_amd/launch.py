"""One process per GPU, started by the program itself.

The reference scores on ONE device in ONE process (/root/reference/sentclf/apply_bert_unlabeled.py:71).
Here `--gpus N` means N ranks, one per MI355X, each a FRESH process (nothing is forked or exec'ed
after a GPU call: the parent starts its children before it touches HIP and only waits for them).
Under `torch.distributed.run` the ranks already exist (RANK / WORLD_SIZE in the environment) and
nothing is spawned.

Backend: "nccl" (= RCCL over xGMI on ROCm) when the node shows at least N GPUs.  With fewer GPUs than
ranks the ranks share cards; RCCL refuses two ranks on one device, so that rehearsal runs over gloo and
switches the fused-LayerNorm flow off (its workgroup exchange assumes the card is not shared).
"""
import os
import socket
import subprocess
import sys


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_gpus():
    """Device count for the PARENT, which must not initialise HIP (its children are started afterwards and the
    parent only waits).  torch counts through amdsmi when it can and falls back to hipGetDeviceCount when it
    cannot, so the count is taken in a short-lived child process and the parent stays clean either way."""
    try:
        out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                             capture_output=True, text=True, timeout=300)
        return int(out.stdout.strip().splitlines()[-1])
    except Exception:
        return 0


def spawn_ranks_if_needed(n_gpus, argv=None, module=None):
    """Called first thing in main().  Returns None when this process IS a rank (or n_gpus <= 1);
    otherwise starts n_gpus rank processes running the same program, waits, and returns the exit
    code the caller must exit with (0 only if every rank returned 0)."""
    if n_gpus <= 1 or "RANK" in os.environ:
        if "WORLD_SIZE" in os.environ and n_gpus > 1 and int(os.environ["WORLD_SIZE"]) != n_gpus:
            raise SystemExit(f"--gpus {n_gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}")
        return None
    argv = list(sys.argv if argv is None else argv)
    cmd = [sys.executable, "-u"] + (["-m", module] + argv[1:] if module else argv)
    n_dev = visible_gpus()
    env0 = dict(os.environ)
    env0.setdefault("MASTER_ADDR", "127.0.0.1")
    env0.setdefault("MASTER_PORT", str(_free_port()))
    env0["WORLD_SIZE"] = env0["LOCAL_WORLD_SIZE"] = str(n_gpus)
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "CLIP_DIST_BACKEND" not in env0:
        env0["CLIP_DIST_BACKEND"] = "nccl" if n_dev >= n_gpus else "gloo"
    if n_dev < n_gpus:
        env0.setdefault("CLIP_AMD_FUSED_LN", "0")
    procs = []
    for r in range(n_gpus):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=env))
    # wait for all; a rank that fails takes the others down (they would wait in a collective for it):
    # SIGTERM first, SIGKILL for whatever is still there after the grace period (a rank stuck inside a kernel
    # or a collective does not act on SIGTERM)
    import time
    rc = 0
    live = list(procs)
    kill_at = None
    while live:
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            if r != 0 and rc == 0:
                rc = r if r > 0 else 128 - r
                for q in live:
                    q.terminate()
                kill_at = time.monotonic() + float(os.environ.get("CLIP_AMD_KILL_GRACE_S", "30"))
        if kill_at is not None and live and time.monotonic() > kill_at:
            for q in live:
                q.kill()
            kill_at = None
        time.sleep(0.1)
    return rc


def init_distributed(timeout_s=7200):
    """(dist module or None, rank, world, device) for this rank; sets the current device.  The timeout
    covers the longest wait of the job: ranks > 0 at the barrier while rank 0 tokenises the corpus."""
    import datetime
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_gpu = torch.cuda.device_count()
    n_dev = max(n_gpu, 1)
    dev_index = local_rank % n_dev
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("CLIP_DIST_BACKEND", "nccl" if n_dev >= int(os.environ.get("LOCAL_WORLD_SIZE", world)) else "gloo")
        kw = {"device_id": torch.device("cuda", dev_index)} if backend == "nccl" else {}
        if n_dev < int(os.environ.get("LOCAL_WORLD_SIZE", world)):
            os.environ.setdefault("CLIP_AMD_FUSED_LN", "0")      # ranks share a card (rehearsal only)
        dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=timeout_s), **kw)
    if n_gpu == 0:      # host-logic tests only: every compute entry point refuses a CPU device
        return dist, rank, world, torch.device("cpu")
    torch.cuda.set_device(dev_index)
    return dist, rank, world, torch.device("cuda", dev_index)
