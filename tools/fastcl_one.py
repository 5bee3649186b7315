#!/usr/bin/env python3
"""ONE table-lookup EM problem of panel size on 1 / 2 / 4 workgroups (test switch emx_fast_wg): kernel time of the k_emx launch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hisatgenotype_amd import engine, locus as hl, synth
htyping = sys.modules["hisatgenotype_amd.typing"]
loc = synth.make_hla_like_locus(gene="A", n_alleles=7000, length=3569, n_vars=2500, seed=500)
pl = hl.PackedLocus.from_synth(loc); pl.index()
for seed in (3, 17):
    b = pl.parse_sam(synth.simulate_sam_fast(loc, synth.pick_sample(loc, 1000 * seed), 5000, err_rate=0.002, seed=100 * seed))
    many = engine.ManyBatch(pl, [b])
    for wgs in (1, 2, 3, 4):
        with engine.test_switches(emx_fast_wg=wgs):
            r = htyping.type_many(pl, many)
            engine.emx_set_timing(True)
            for _ in range(5):
                htyping.type_many(pl, many)
            ms, nl, nj, na, nb = engine.emx_get_timing(True)
            engine.emx_set_timing(False)
        print("seed %d  classes %d  iterations %d  wgs %d: %.3f ms per table-lookup launch (%d applications)" % (seed, r[0].em[0]["n_classes"], r[0].em[0]["n_iter"], wgs, ms / nl, na // nl))
