import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hisatgenotype_amd
from hisatgenotype_amd import capi, engine
capi.use_lab()            # (drives comparison kernels that live in the lab build since round 4)
rng = np.random.RandomState(1)
A, Cn = 7000, 900
ap = capi.a_pad(A); w64 = ap // 64
bits = np.zeros((Cn, w64), np.uint64)
dense = np.zeros((Cn, ap), bool)
for k in range(Cn):
    m = rng.rand(A) < rng.choice([0.002, 0.05, 0.6])
    m[rng.randint(A)] = True
    dense[k, :A] = m
    bits[k] = np.packbits(dense[k], bitorder="little").view(np.uint64)
counts = rng.randint(1, 500, Cn).astype(np.int64)
cl = engine.Classes.from_host(bits, counts, ap)
x = np.zeros(ap); x[:A] = rng.rand(A) ** 8 * rng.choice([1e-12, 1e-6, 1.0], A)
exact = counts / (dense.astype(np.float64) @ x)
for backend in (1, 2):
    y = np.zeros(Cn)
    capi.check(capi.lib().hgx_debug_matvec(cl.h, 0, backend, capi.ptr(x), capi.ptr(y)))
    print("rows backend", backend, "max rel err", np.max(np.abs(y - exact) / exact), y[:3], exact[:3])
xc = rng.rand(Cn) * 10.0 ** rng.randint(-3, 12, Cn)
exact = dense.astype(np.float64).T @ xc
for backend in (1, 2):
    y = np.zeros(ap)
    capi.check(capi.lib().hgx_debug_matvec(cl.h, 1, backend, capi.ptr(xc), capi.ptr(y)))
    nz = exact > 0
    print("cols backend", backend, "max rel err", np.max(np.abs(y[nz] - exact[nz]) / exact[nz]), y[:3], exact[:3])
