"""CPU pipeline on the oracles (TEST INFRASTRUCTURE): integer haplotypes -> classes -> EM -> abundances,
following the reference's orchestration (typing_core.py:1650-1789).  Used by tests and by bench.py's
cpu_baseline leg only."""
import time

import numpy as np


def bits_to_indices(row):
    out = []
    for w, word in enumerate(row):
        word = int(word)
        while word:
            b = (word & -word).bit_length() - 1
            out.append(64 * w + b)
            word &= word - 1
    return out


def em_sorted(orc, t, bits, counts, remove_low, lengths):
    """single_abundance on a class bit matrix; alleles inside a class in key (sorted-name) order."""
    names = t["names"]
    rank = {a: r for r, a in enumerate(sorted(range(len(names)), key=lambda i: names[i]))}
    classes = [sorted(bits_to_indices(r), key=lambda a: rank[a]) for r in bits]
    oa, op, it = orc.single_abundance(t["n_alleles"], classes, counts, remove_low, lengths)
    return list(zip(oa.tolist(), op.tolist())), it


def run(orc, t, arrs, hla, lengths, remove_low=True):
    """Returns dict(times, n_iter, gene_prob [(allele index, prob)])."""
    out = {}
    L = orc.make_locus(t)
    t0 = time.perf_counter()
    eb, gb, gc, fp = orc.score_pairs(L, t["exon_keys"], t["gene_keys"], *arrs)
    t1 = time.perf_counter()
    gub, guc, _ = orc.dedup(gb)
    n_iter = 0
    ems = []                     # per single_abundance call: (n_classes, n_iter, [(allele index, prob)])
    out["gene_classes"] = (gub, guc)
    if hla:
        eub, euc, _ = orc.dedup(eb)
        out["exon_classes"] = (eub, euc)
        t2 = time.perf_counter()
        exon_prob, it = em_sorted(orc, t, eub, euc, remove_low, None)
        ems.append((len(euc), it, exon_prob))
        n_iter += it
        gene_prob = exon_prob
        aidx, groups = t["aidx"], t["rep_groups"]
        exon_alleles, psum = set(), 0.0
        for i, (a, p) in enumerate(exon_prob):
            if i >= 10 and p < 0.03:
                break
            g = groups[t["names"][a]]
            if len(g) <= 1:
                continue
            psum += p
            exon_alleles |= {aidx[x] for x in g}
        if exon_alleles:
            mask = np.zeros(gub.shape[1], np.uint64)
            for a in exon_alleles:
                mask[a >> 6] |= np.uint64(1) << np.uint64(a & 63)
            g2b, g2c, _ = orc.dedup(gub, weight=guc, and_mask=mask)
            gp, it = em_sorted(orc, t, g2b, g2c, True, lengths)
            ems.append((len(g2c), it, gp))
            n_iter += it
            comb = {}
            for a, p in exon_prob:
                if a not in exon_alleles:
                    comb[a] = p
            for a, p in gp:
                comb[a] = p * psum
            gene_prob = sorted(comb.items(), key=lambda x: x[1], reverse=True)
    else:
        t2 = time.perf_counter()
        gene_prob, it = em_sorted(orc, t, gub, guc, False, None)
        ems.append((len(guc), it, gene_prob))
        n_iter += it
    t3 = time.perf_counter()
    out.update(t_score=t1 - t0, t_dedup=t2 - t1, t_em=t3 - t2, n_iter=n_iter, gene_prob=list(gene_prob),
               gene_counts=gc, first_pair=fp, em=ems)
    return out
