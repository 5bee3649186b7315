import sys, numpy as np, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import hisatgenotype_amd as hgx
from hisatgenotype_amd import engine, locus as hl, synth
htyping = sys.modules["hisatgenotype_amd.typing"]
loc = synth.make_hla_like_locus(n_alleles=7000, n_vars=2500, seed=31)
pl = hl.PackedLocus.from_synth(loc)
batches=[]
for s_,n in enumerate([5000,3000,800,5000,200,5000]):
    sam = synth.simulate_sam_fast(loc, synth.pick_sample(loc, 50+s_), n, err_rate=0.003, seed=7*s_+1)
    batches.append(pl.parse_sam(sam))
many = engine.ManyBatch(pl, batches)
def key(r): return (r.num_reads, [ (e["n_iter"], e["result"]) for e in r.em], r.gene_prob)
with engine.test_switches(emx_fast_wg=1):
    base = htyping.type_many(pl, many)
for wgs in (2,3,4):
    with engine.test_switches(emx_fast_wg=wgs):
        got = htyping.type_many(pl, many)
    print(wgs, all(key(a)==key(b) for a,b in zip(base,got)), engine.emx_cluster_stats())
got = htyping.type_many(pl, many)
print('default', all(key(a)==key(b) for a,b in zip(base,got)), engine.emx_cluster_stats())
for wgs in (1,2,4,None):
    sw = {} if wgs is None else dict(emx_fast_wg=wgs)
    with engine.test_switches(**sw):
        htyping.type_many(pl, many)
        t0=time.perf_counter()
        for _ in range(10): htyping.type_many(pl, many)
        print('wgs',wgs,'ms per call', (time.perf_counter()-t0)*100)
