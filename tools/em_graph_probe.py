"""EM #1 of the bench workload alone: one launch per pass as queued by the host, against the same launches captured into a hipGraph per
batch (test switch em_graph; capture + instantiation + launch inside the call).  usage: python tools/em_graph_probe.py [pairs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hisatgenotype_amd  # noqa
from hisatgenotype_amd import synth, locus as hl, engine, capi
hgx = sys.modules["hisatgenotype_amd.typing"]
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
loc = synth.make_hla_like_locus(n_alleles=7000, n_vars=2500, seed=101)
pl = hl.PackedLocus.from_synth(loc)
sample = synth.pick_sample(loc, 101)
sam = synth.simulate_sam_fast(loc, sample, n_pairs, err_rate=0.002, seed=100)
res = hgx.type_locus(pl, sam, keep_classes=True)
bits, cnt = res.exon_classes
print("exon classes %d, alleles %d, EM #1 iterations %d" % (len(cnt), pl.n_alleles, res.em[0]["n_iter"]))
cl = engine.Classes.from_host(bits, cnt, pl.a_pad)
st = capi.get_stream(2)            # (the legacy default stream cannot be captured)
ref = None
for label, sw in (("one launch per pass", None), ("hipGraph per batch", "1"), ("one launch per pass", None), ("hipGraph per batch", "1")):
    engine.test_switch("em_graph", sw)
    for _ in range(3):
        p, it = cl.em(pl.n_alleles, True, None, st)
    capi.sync(st)
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        p, it = cl.em(pl.n_alleles, True, None, st)
    dt = (time.perf_counter() - t0) / n
    if ref is None:
        ref = p
    print("%-22s %.3f ms per EM call (%d iterations) identical results: %s" % (label, dt * 1e3, it, np.array_equal(p, ref)))
engine.test_switch("em_graph", None)
