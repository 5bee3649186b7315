"""How many DISTINCT columns does the exon-level class matrix of the bench workload have?  (Alleles that occur in exactly the
same classes are indistinguishable to the EM and can be merged with a multiplicity.)  Dumps the compact matrix for offline
experiments: gpurun_out/em_matrix.npz."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hisatgenotype_amd  # noqa
from hisatgenotype_amd import synth, locus as hl
ht = sys.modules["hisatgenotype_amd.typing"]
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
loc = synth.make_hla_like_locus(n_alleles=7000, n_vars=2500, seed=101)
pl = hl.PackedLocus.from_synth(loc)
pl.index()
sample = synth.pick_sample(loc, 101)
sam = synth.simulate_sam_fast(loc, sample, n_pairs, err_rate=0.002, seed=100)
res = ht.type_locus(pl, sam, keep_classes=True)
for lvl, (bits, cnt) in (("exon", res.exon_classes), ("gene", res.gene_classes)):
    C = bits.shape[0]
    b = np.unpackbits(bits.view(np.uint8), axis=1, bitorder="little")          # [C][a_pad]
    act = np.nonzero(b.any(axis=0))[0]
    cols = np.ascontiguousarray(b[:, act].T)                                     # [A'][C]
    packed = np.packbits(cols, axis=1)
    uniq, inv, mult = np.unique(packed, axis=0, return_inverse=True, return_counts=True)
    print("%s level: C = %d classes, A' = %d active alleles, distinct columns G = %d (largest group %d, singletons %d)" % (
        lvl, C, len(act), len(uniq), mult.max(), int((mult == 1).sum())))
    sizes = b.sum(axis=1)
    print("   class sizes: min %d median %d max %d; nnz %d" % (sizes.min(), np.median(sizes), sizes.max(), sizes.sum()))
    if lvl == "exon":
        os.makedirs("gpurun_out", exist_ok=True)
        np.savez_compressed("gpurun_out/em_matrix.npz", bits=bits, cnt=cnt, act=act)
print("EM:", [(e["n_classes"], e["n_iter"]) for e in res.em], res.gene_prob[:3])
