"""How many cores does this box really give us?  cgroup quota + a pure-compute scaling probe (no memory traffic)."""
import os, sys, time, threading, ctypes
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try:
        print(f, "=", open(f).read().strip())
    except Exception as e:
        print(f, "n/a")
print("affinity:", len(os.sched_getaffinity(0)), "cpu_count:", os.cpu_count())
import numpy as np
a = np.random.rand(200000)
def work(n):
    for _ in range(n):
        np.sort(a)          # releases the GIL
for nt in (1, 8, 16, 32, 64, 128, 256):
    th = [threading.Thread(target=work, args=(20,)) for _ in range(nt)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print("threads %3d: %.3f s  -> %.1f sorts/s (%.2fx of 1 thread ideal)" % (nt, dt, nt * 20 / dt, 0), flush=True)
