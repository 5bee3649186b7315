"""N device front-end passes over the configs[0] fixture (10 000 records): what a small input's 1.4 ms is made of (under rocprofv3)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu
import hisatgenotype_amd  # noqa
from hisatgenotype_amd import capi, locus as hl
capi.set_device(0)
fx = gu.load(sys.argv[2] if len(sys.argv) > 2 else "hla_7000_10k")
pl = hl.PackedLocus.from_synth(fx["_locus"])
pl.index()
sam = fx["sam"].encode()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
import time
for _ in range(3):
    pl.parse_sam_dev(sam).close()
t0 = time.perf_counter()
for _ in range(n):
    pl.parse_sam_dev(sam).close()
print("%.3f ms per parse" % ((time.perf_counter() - t0) / n * 1e3))
