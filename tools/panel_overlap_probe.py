#!/usr/bin/env python3
"""configs[3] panel step with the gene level of the HLA loci scored BESIDE the launch of EM #1 (the default) and BEFORE it
(test switch many=rest_first = the form of rounds 3-4), results compared task by task.  usage: tools/panel_overlap_probe.py [samples]"""
import os, sys, time
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
def sig(rows):
    return [[(r.num_reads, r.counts.tolist(), r.gene_prob, [(e["n_iter"], e["n_classes"], e["result"]) for e in r.em]) for r in row] for row in rows]
ref = None
for rnd in range(2):
    for sw in (dict(many="rest_first"), dict()):
        with engine.test_switches(**sw):
            rows = htyping.type_many_loci(pls, manies)
            s = sig(rows)
            if ref is None: ref = s
            assert s == ref, "results differ"
            htyping.type_many_loci(pls, manies, light=True)
            ts = []
            for _ in range(8):
                t0 = time.perf_counter()
                htyping.type_many_loci(pls, manies, light=True)
                ts.append((time.perf_counter() - t0) * 1e3)
        print(sw or "default (gene level beside EM #1)", "step median %.2f ms  min %.2f" % (sorted(ts)[len(ts) // 2], min(ts)), flush=True)
os.environ["HGX_TYPE_PROFILE"] = "1"
for sw in (dict(many="rest_first"), dict()):
    with engine.test_switches(**sw):
        htyping.type_many_loci(pls, manies, light=True)
