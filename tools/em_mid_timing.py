"""Mid-size EM: k_em_ref (reference order, one workgroup) against the table-lookup path on the same random problems."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hisatgenotype_amd  # noqa
from hisatgenotype_amd import engine, capi
capi.use_lab()            # (drives comparison kernels that live in the lab build since round 4)
rng = np.random.RandomState(5)
for A, n_used, C_, dens in ((700, 300, 200, 0.1), (700, 600, 600, 0.1), (2000, 1000, 1000, 0.1), (7000, 1024, 2048, 0.05), (7000, 1024, 2048, 0.3)):
    a_pad = capi.a_pad(A)
    used = np.sort(rng.choice(A, n_used, replace=False))
    fam = rng.rand(6, n_used) < dens * rng.choice([0.5, 1.0, 3.0], size=6)[:, None]
    rows = np.zeros((C_, a_pad // 64), np.uint64)
    for c in range(C_):
        m = fam[rng.randint(6)] ^ (rng.rand(n_used) < 0.02)
        m[rng.randint(n_used)] = True
        for a in used[m]:
            rows[c, a >> 6] |= np.uint64(1) << np.uint64(a & 63)
    counts = rng.randint(1, 300, C_).astype(np.int64)
    cl = engine.Classes.from_host(rows, counts, a_pad)
    cl.set_allele_rank(np.arange(A, dtype=np.int32))
    out = []
    for env in (None, "1"):
        engine.test_switch("em_no_mid", env)
        for _ in range(3):
            p, it = cl.em(A, True, None)
        t0 = time.perf_counter()
        for _ in range(20):
            p, it = cl.em(A, True, None)
        out.append(((time.perf_counter() - t0) / 20 * 1e3, it))
    engine.test_switch("em_no_mid", None)
    print("C=%5d alleles=%5d density~%.2f: reference order %.3f ms (%d iterations) | table lookup %.3f ms (%d iterations)" % (
        C_, n_used, dens, out[0][0], out[0][1], out[1][0], out[1][1]))
