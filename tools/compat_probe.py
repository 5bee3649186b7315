#!/usr/bin/env python3
"""k_piece_compat_pat alone at several numbers of distinct pieces (one sample; the merged batch of 16 / 64 samples of a locus),
(the pieces-per-workgroup sweep of round 4 -- 256 / 512 / 1024 -- was made with a temporary knob: 256 won at every size).  usage: tools/compat_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hisatgenotype_amd import capi, engine, locus as hl, synth
loc = synth.make_hla_like_locus(gene="A", n_alleles=7000, length=3569, n_vars=2500, seed=500)
pl = hl.PackedLocus.from_synth(loc); pl.index()
def batch_of(n_samples, pairs):
    bs = [pl.parse_sam(synth.simulate_sam_fast(loc, synth.pick_sample(loc, 1000 * s), pairs, err_rate=0.002, seed=100 * s)) for s in range(n_samples)]
    if n_samples == 1:
        return engine.DeviceBatch(bs[0])
    m = engine.ManyBatch(pl, bs)
    return engine.DeviceBatch(m.merged())
for n_samples, pairs in ((1, 500000), (16, 5000), (64, 5000)):
    db = batch_of(n_samples, pairs)
    bufs = engine.ScoreBuffers(pl, db)
    line = "%2d sample(s) x %6d pairs: %6d distinct pieces:" % (n_samples, pairs, db.n_pieces)
    engine.piece_compat(pl, db, bufs); capi.sync()
    t0 = time.perf_counter()
    for _ in range(20): engine.piece_compat(pl, db, bufs)
    capi.sync()
    line += "  %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3)
    print(line, flush=True)
