import sys, os, time, cProfile, pstats
sys.path.insert(0, "/root/repo")
sys.argv = ["bench.py", "--no-cpu-baseline", "--steps", "30", "--no-kernel-timing"]
import bench
pr = cProfile.Profile()
orig = bench.run_steps
calls = {"n": 0}
def wrapped(*a, **k):
    calls["n"] += 1
    if calls["n"] == 2:
        pr.enable(); r = orig(*a, **k); pr.disable(); return r
    return orig(*a, **k)
bench.run_steps = wrapped
bench.main()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
