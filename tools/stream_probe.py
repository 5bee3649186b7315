"""Which streams share a hardware queue?  n fresh streams, every pair probed with two 150 us spin kernels (hgx_stream_probe_matrix)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hisatgenotype_amd  # noqa: E402,F401
from hisatgenotype_amd import capi  # noqa: E402

capi.set_device(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
pattern = sys.argv[2] if len(sys.argv) > 2 else "H" * n
prio = np.array([1 if c == "H" else 0 for c in pattern[:n].ljust(n, "H")], np.int32)
us = np.zeros((n, n), np.float64)
capi.check(capi.lib().hgx_stream_probe_matrix(C.c_int32(n), capi.ptr(prio), capi.ptr(us)))
print("priorities:", "".join("H" if p else "L" for p in prio))
np.set_printoptions(linewidth=250, precision=0, suppress=True)
print(us)
print("serial (>= 240 us):")
print((us >= 240).astype(int))
