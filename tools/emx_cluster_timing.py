import os, sys, time
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hisatgenotype_amd
from hisatgenotype_amd import engine
from test_gpu_emx import _random_problem
engine.test_switch("emx_stamps", "1")
A, n_used, C_, dens = (7000, 4549, 16098, 0.3)
rng = np.random.RandomState(5)
a_pad, name_rank, classes, rows, counts, lengths = _random_problem(rng, A, n_used, C_, dens)
cl = engine.Classes.from_host(rows, counts, a_pad)
cl.set_allele_rank(name_rank)
engine.em_set_fast(-1)
for rep in range(2):
    t0 = time.perf_counter(); p, it = cl.em(A, True, None); dt = time.perf_counter() - t0
print("cluster: iters", it, "call %.2f ms" % (dt * 1e3), flush=True)
