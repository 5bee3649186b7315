"""EM #1 of the bench workload alone on the GPU: the 8-bit slab form of rounds 1-3 (k_lutmatvec, switch em_lut8) against the
narrow-table form (k_lut4, lab build: switch em_lut4) at several rows-per-workgroup settings.  usage: python tools/em_forms.py [pairs]
Measured on MI355X (round 4): 1.015 ms per EM call for the product form against 2.33 ms for the narrow form at its best."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hisatgenotype_amd  # noqa
from hisatgenotype_amd import synth, locus as hl, engine, capi
capi.use_lab()
hgx = sys.modules["hisatgenotype_amd.typing"]
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
loc = synth.make_hla_like_locus(n_alleles=7000, n_vars=2500, seed=101)
pl = hl.PackedLocus.from_synth(loc)
sample = synth.pick_sample(loc, 101)
sam = synth.simulate_sam_fast(loc, sample, n_pairs, err_rate=0.002, seed=100)
res = hgx.type_locus(pl, sam, keep_classes=True)
bits, cnt = res.exon_classes
print("exon classes %d, alleles %d, EM #1 iterations %d" % (len(cnt), pl.n_alleles, res.em[0]["n_iter"]))
ref = None
for label, sw in (("8-bit slabs (product)", {}), ("narrow, default rows", {"em_lut4": 1}), ("narrow, 16 rows", {"em_lut4": 1, "l4_rows": 16}),
                  ("narrow, 32 rows", {"em_lut4": 1, "l4_rows": 32}), ("narrow, 64 rows", {"em_lut4": 1, "l4_rows": 64}),
                  ("narrow, 128 rows", {"em_lut4": 1, "l4_rows": 128})):
    with engine.test_switches(**sw):
        cl = engine.Classes.from_host(bits, cnt, pl.a_pad)
        for _ in range(3):
            p, it = cl.em(pl.n_alleles, True, None)
        capi.sync(None)
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            p, it = cl.em(pl.n_alleles, True, None)
        dt = (time.perf_counter() - t0) / n
    if ref is None:
        ref = p
    dev = float(np.max(np.abs(np.where(p >= 0, p, 0) - np.where(ref >= 0, ref, 0))))
    print("%-24s %.3f ms per EM call (%d iterations), max |p - p(8-bit)| = %.2e" % (label, dt * 1e3, it, dev))
