"""Phase profile of k_emx (test switch emx_stamps) on panel-sized random problems, and wall time per call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hisatgenotype_amd
from hisatgenotype_amd import engine
from hisatgenotype_amd import capi
capi.use_lab()            # (drives comparison kernels that live in the lab build since round 4)
from test_gpu_emx import _random_problem
engine.test_switch("emx_stamps", "1")
for case in [(7000, 4549, 1600, 0.25), (3000, 1949, 1340, 0.27), (500, 323, 704, 0.3), (2000, 1100, 500, 0.05)]:
    A, n_used, C_, dens = case
    rng = np.random.RandomState(5)
    a_pad, name_rank, classes, rows, counts, lengths = _random_problem(rng, A, n_used, C_, dens)
    cl = engine.Classes.from_host(rows, counts, a_pad)
    cl.set_allele_rank(name_rank)
    for rep in range(2):
        t0 = time.perf_counter()
        p, it = cl.em(A, True, None)
        dt = time.perf_counter() - t0
    print(case, "iters", it, "exact", engine.em_last_exact(), "call %.2f ms" % (dt * 1e3), flush=True)
    engine.em_set_fast(True)                                        # the same kernel skeleton with table lookups (hgx_type_opts.em_fast)
    for rep in range(2):
        t0 = time.perf_counter()
        pf, itf = cl.em(A, True, None)
        dt = time.perf_counter() - t0
    engine.em_set_fast(False)
    print("   fast mode: iters", itf, "call %.2f ms" % (dt * 1e3), "max |diff| %.3g" % float(np.max(np.abs(p - pf))), flush=True)
    engine.test_switch("em_skip", "emx"); engine.test_switch("em_no_mid", "1")
    for rep in range(2):
        t0 = time.perf_counter()
        p2, it2 = cl.em(A, True, None)
        dt = time.perf_counter() - t0
    engine.test_switch("em_skip", None); engine.test_switch("em_no_mid", None)
    print("   table-lookup path: iters", it2, "call %.2f ms" % (dt * 1e3), "max |diff| %.3g" % float(np.max(np.abs(p - p2))), flush=True)
