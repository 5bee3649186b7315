import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hisatgenotype_amd
from hisatgenotype_amd import capi, engine
capi.use_lab()            # (drives comparison kernels that live in the lab build since round 4)
rng = np.random.RandomState(1)
A, Cn = 7000, 16098
ap = capi.a_pad(A); w64 = ap // 64
bits = rng.randint(0, 2**63, size=(Cn, w64), dtype=np.int64).astype(np.uint64)
counts = rng.randint(1, 500, Cn).astype(np.int64)
cl = engine.Classes.from_host(bits, counts, ap)
x = np.zeros(ap); x[:A] = rng.rand(A)
xc = rng.rand(Cn)
engine.test_switch("dbg_reps", "50")
for backend in [int(b) for b in os.environ.get('BACKENDS', '1,3').split(',')]:
    for which, xx, ny in ((0, x, Cn), (1, xc, ap)):
        y = np.zeros(ny)
        capi.check(capi.lib().hgx_debug_matvec(cl.h, which, backend, capi.ptr(xx), capi.ptr(y)))
