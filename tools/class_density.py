"""Density structure of the EM's class matrices (exon level of the bench workload, compact active alleles): how much would storing
the COMPLEMENT of dense rows buy the table-lookup passes (which skip 64 x 64 all-zero tiles)?  usage: python tools/class_density.py [pairs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hisatgenotype_amd  # noqa
from hisatgenotype_amd import synth, locus as hl
hgx = sys.modules["hisatgenotype_amd.typing"]
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
for (g, a, ln, v, sd, pairs) in [("A", 7000, 3569, 2500, 101, n_pairs), ("A", 7000, 3569, 2500, 500, 5000)]:
    loc = synth.make_hla_like_locus(gene=g, n_alleles=a, length=ln, n_vars=v, seed=sd)
    pl = hl.PackedLocus.from_synth(loc)
    sam = synth.simulate_sam_fast(loc, synth.pick_sample(loc, 101), pairs, err_rate=0.002, seed=100)
    res = hgx.type_locus(pl, sam, keep_classes=True)
    bits, cnt = res.exon_classes
    C = len(cnt)
    m = np.unpackbits(bits.view(np.uint8), axis=1, bitorder="little")[:, :pl.n_alleles].astype(bool)
    act = m.any(axis=0)
    m = m[:, act]
    A1 = m.shape[1]
    dens = m.mean(axis=1)
    w = cnt / cnt.sum()
    print("%s %d pairs: %d classes x %d active alleles; nnz %.1f%%; rows denser than 50%%: %.1f%% of the rows, %.1f%% of the pairs; mean density %.3f"
          % (g, pairs, C, A1, 100 * m.mean(), 100 * (dens > 0.5).mean(), 100 * w[dens > 0.5].sum(), dens.mean()))
    def zero_tiles(mat):
        Cp, Ap = (mat.shape[0] + 63) // 64 * 64, (mat.shape[1] + 63) // 64 * 64
        z = np.zeros((Cp, Ap), bool); z[:mat.shape[0], :mat.shape[1]] = mat
        t = z.reshape(Cp // 64, 64, Ap // 64, 64).any(axis=(1, 3))
        return 1.0 - t.mean()
    order = np.argsort(-dens, kind="stable")
    mc = m.copy()
    mc[dens > 0.5] = ~mc[dens > 0.5]
    print("   all-zero 64x64 tiles: as stored %.1f%%; dense rows complemented %.1f%%; complemented AND rows sorted by density %.1f%%; nnz after complement %.2f%%"
          % (100 * zero_tiles(m), 100 * zero_tiles(mc), 100 * zero_tiles(mc[order]), 100 * mc.mean()))
