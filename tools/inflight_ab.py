"""N samples in flight, as bench.py times them (run from the root of the tree to measure: python tools/inflight_ab.py)."""
import os
import sys
import time

ROOT = os.getcwd()
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from hisatgenotype_amd import capi, synth, locus as hl  # noqa: E402

from hisatgenotype_amd import engine  # noqa: E402
if os.environ.get("HGX_STREAMS"):
    engine.test_switch("streams", os.environ["HGX_STREAMS"])
capi.set_device(0)
loc = synth.make_hla_like_locus(n_alleles=7000, n_vars=2500, seed=101)
pl = hl.PackedLocus.from_synth(loc)
pl.index()
sam = synth.simulate_sam_fast(loc, synth.pick_sample(loc, 101), 500000, err_rate=0.002, seed=100)
batch = pl.parse_sam(sam)
db = pl.parse_sam_dev(sam)
del sam
out = []
for n, steps in ((1, 20), (2, 20), (3, 30), (2, 20), (1, 20)):
    bench.run_steps(pl, batch, db, n, 3 * n, None, False, 0)
    capi.sync()
    t0 = time.perf_counter()
    bench.run_steps(pl, batch, db, n, steps, None, False, 0)
    capi.sync()
    out.append("%d: %.3f" % (n, (time.perf_counter() - t0) / steps * 1e3))
print(os.path.basename(ROOT), os.environ.get("HGX_STREAMS", ""), "ms per step with n in flight:", "  ".join(out))
