"""EM #1 of the bench workload alone on the GPU (no gene side beside it): resident-block kernel vs one launch per pass.
usage: python tools/em_alone.py [pairs]      (--stamps prints the phase profile of k_em_grid)
The resident-block kernel is lab code: this tool binds libhgx_lab.so (build.build_lab())."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hisatgenotype_amd  # noqa
from hisatgenotype_amd import synth, locus as hl, engine, capi
capi.use_lab()
hgx = sys.modules["hisatgenotype_amd.typing"]
want_stamps = "--stamps" in sys.argv
if want_stamps:
    sys.argv.remove("--stamps")
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
loc = synth.make_hla_like_locus(n_alleles=7000, n_vars=2500, seed=101)
pl = hl.PackedLocus.from_synth(loc)
sample = synth.pick_sample(loc, 101)
sam = synth.simulate_sam_fast(loc, sample, n_pairs, err_rate=0.002, seed=100)
res = hgx.type_locus(pl, sam, keep_classes=True)
bits, cnt = res.exon_classes
print("exon classes %d, alleles %d, EM #1 iterations %d" % (len(cnt), pl.n_alleles, res.em[0]["n_iter"]))
cl = engine.Classes.from_host(bits, cnt, pl.a_pad)
for label, env in (("per pass", None), ("resident blocks", "1")):
    engine.test_switch("em_grid", env)
    engine.test_switch("grid_stamps", None)
    for _ in range(3):
        p, it = cl.em(pl.n_alleles, True, None)
    capi.sync(None)
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        p, it = cl.em(pl.n_alleles, True, None)
    dt = (time.perf_counter() - t0) / n
    print("%-16s %.3f ms per EM call (%d iterations)" % (label, dt * 1e3, it))
    if want_stamps and env:
        engine.test_switch("grid_stamps", "1")
        cl.em(pl.n_alleles, True, None)
