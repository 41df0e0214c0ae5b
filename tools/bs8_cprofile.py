"""cProfile of the host side of tools/bs8_latency.py's loop (which Python / torch calls the per-batch time goes to):
    python tools/bs8_cprofile.py"""
import cProfile, pstats, sys, os, io
sys.argv = ["bs8", "1024", "8"]
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/tools")
src = open(os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/tools/bs8_latency.py").read()
src = src.split("sc = scoring.BulkScorer")[0]
g = {"__name__": "__main__", "__file__": os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/tools/bs8_latency.py"}
exec(compile(src, "bs8_latency.py", "exec"), g)
pr = cProfile.Profile(); pr.enable(); g["loop"](); g["torch"].cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue())
