#!/usr/bin/env python3
"""Per-task EM #1 sizes and iteration counts of the configs[3] panel (which tasks make the tail of the one-launch EM), and the launch
time with the table-lookup problems on 1 / 2 workgroups (test switch emx_fast_wg).  usage: tools/panel_em_profile.py [samples]"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hisatgenotype_amd as hgx
from hisatgenotype_amd import engine, locus as hl, synth
import bench
htyping = sys.modules["hisatgenotype_amd.typing"]
n_s = int(sys.argv[1]) if len(sys.argv) > 1 else 64
loci = [synth.make_hla_like_locus(gene=g, n_alleles=a, length=ln, n_vars=v, seed=500 + i, var_id_base=10000 * i) for i, (g, a, ln, v) in enumerate(bench.PANEL)]
pls, manies = [], []
for k, loc in enumerate(loci):
    pl = hl.PackedLocus.from_synth(loc); pl.index()
    bs = [pl.parse_sam(synth.simulate_sam_fast(loc, synth.pick_sample(loc, 1000 * s + k), 5000, err_rate=0.002, seed=100 * s + k)) for s in range(n_s)]
    pls.append(pl); manies.append(engine.ManyBatch(pl, bs))
rows = htyping.type_many_loci(pls, manies)
for k, row in enumerate(rows):
    its = [r.em[0]["n_iter"] for r in row]; cs = [r.em[0]["n_classes"] for r in row]
    print("%-5s alleles %5d  classes %d..%d (mean %.0f)  EM #1 iterations: min %d mean %.1f max %d  hist %s" % (
        loci[k].gene, len(loci[k].allele_names), min(cs), max(cs), sum(cs) / len(cs), min(its), sum(its) / len(its), max(its),
        sorted(collections.Counter(its).items())))
for sw in (dict(emx_fast_wg=1), dict(emx_fast_wg=2), dict()):
    with engine.test_switches(**sw):
        htyping.type_many_loci(pls, manies, light=True)
        engine.emx_set_timing(True)
        t0 = time.perf_counter()
        for _ in range(5):
            htyping.type_many_loci(pls, manies, light=True)
        dt = (time.perf_counter() - t0) / 5
        ms, nl, nj, na, nb = engine.emx_get_timing(True)
        engine.emx_set_timing(False)
    print(sw, "step %.2f ms; k_emx table-lookup launches: %.2f ms each" % (dt * 1e3, ms / max(nl, 1)))
