#!/usr/bin/env python3
"""Wall-clock breakdown of one bench step (host + device), to find host-side overheads."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import hisatgenotype_amd as hgx
from hisatgenotype_amd import capi, engine, synth, locus as hl
ht = sys.modules["hisatgenotype_amd.typing"]

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
loc = synth.make_hla_like_locus(n_alleles=7000, n_vars=2500, seed=101)
pl = hl.PackedLocus.from_synth(loc)
sample = synth.pick_sample(loc, 101)
sam = synth.simulate_sam_fast(loc, sample, pairs, err_rate=0.002, seed=100)
batch = pl.parse_sam(sam)
db = engine.DeviceBatch(batch)
bufs = engine.ScoreBuffers(pl, db, exon=True)

class T:
    def __init__(self): self.t = {}; self.last = None
    def start(self): capi.sync(); self.last = time.perf_counter()
    def lap(self, name):
        capi.sync(); now = time.perf_counter(); self.t[name] = self.t.get(name, 0.0) + now - self.last; self.last = now

def step(T):
    T.start()
    engine.score_pairs(pl, db, bufs); T.lap("score")
    gcl = engine.Classes.dedup(bufs.gene_bits, db.n_pairs, pl.a_pad, hashes=bufs.gene_hash); T.lap("dedup_gene")
    cnt, first = gcl.allele_counts(); T.lap("allele_counts")
    fr = np.zeros(gcl.n_classes, np.int64)
    capi.check(capi.lib().hgx_classes_to_host(gcl.h, None, None, capi.ptr(fr)))
    A = pl.n_alleles
    counted = [a for a in range(A) if cnt[a] > 0]
    counted.sort(key=lambda a: (fr[first[a]], a)); counted.sort(key=lambda a: -cnt[a]); T.lap("py_counts_sort")
    ecl = engine.Classes.dedup(bufs.exon_bits, db.n_pairs, pl.a_pad, hashes=bufs.exon_hash); T.lap("dedup_exon")
    prob, it = ecl.em(A, True, None); T.lap("em1")
    _, f2 = ecl.allele_counts(); T.lap("em1_first")
    order = engine.em_order(f2[:A], pl.name_rank, prob >= 0.0); res = ht._sorted_result(prob, order); T.lap("py_em_order")
    groups = pl.rep_groups(); T.lap("py_rep_groups")
    return gcl.n_classes, ecl.n_classes, it

tm = T()
for k in range(6):
    if k == 1: tm = T()
    out = step(tm)
print(out)
for k, v in tm.t.items():
    print("%-16s %8.3f ms" % (k, v / 5 * 1e3))
print("sum %.3f ms" % (sum(tm.t.values()) / 5 * 1e3))
