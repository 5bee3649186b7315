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


def classes_in_key_order(t, bits):
    """Class bit matrix -> (offsets [C + 1], alleles) with the alleles of a class in key order ('-'.join(sorted(names)): the order
    single_abundance walks them in, common:1300-1305) -- numpy, for class tables of tens of thousands of rows."""
    names = t["names"]
    A = len(names)
    perm = np.array(sorted(range(A), key=lambda i: names[i]), dtype=np.int64)          # name order -> allele index
    bits = np.ascontiguousarray(bits, dtype=np.uint64)
    off = np.zeros(len(bits) + 1, dtype=np.int64)
    flat = []
    for c0 in range(0, len(bits), 2048):
        u = np.unpackbits(bits[c0:c0 + 2048].view(np.uint8), axis=1, bitorder="little")[:, :A][:, perm]
        off[c0 + 1:c0 + 1 + len(u)] = u.sum(axis=1)
        flat.append(perm[np.nonzero(u)[1]])
    np.cumsum(off, out=off)
    return off.astype(np.int32), (np.concatenate(flat) if flat else np.zeros(0, np.int64)).astype(np.int32)


def em_sorted(orc, t, bits, counts, remove_low, lengths):
    """single_abundance on a class bit matrix; alleles inside a class in key (sorted-name) order."""
    off, flat = classes_in_key_order(t, bits)
    oa, op, it = orc.single_abundance_flat(t["n_alleles"], off, flat, counts, remove_low, lengths)
    return list(zip(oa.tolist(), op.tolist())), it


def run(orc, t, arrs, hla, lengths, remove_low=True):
    """Returns dict(times, n_iter, gene_prob [(allele index, prob)])."""
    L = orc.make_locus(t)
    t0 = time.perf_counter()
    eb, gb, gc, fp = orc.score_pairs(L, t["exon_keys"], t["gene_keys"], *arrs)
    t1 = time.perf_counter()
    gub, guc, _ = orc.dedup(gb)
    eub = euc = None
    if hla:
        eub, euc, _ = orc.dedup(eb)
    t2 = time.perf_counter()
    out = finish(orc, t, eub, euc, gub, guc, gc, fp, hla, lengths, remove_low)
    out.update(t_score=t1 - t0, t_dedup=t2 - t1)
    return out


def finish(orc, t, eub, euc, gub, guc, gc, fp, hla, lengths, remove_low=True):
    """Everything behind the class tables (typing_core.py:1650-1789): EM #1, the exon -> gene hand-off, EM #2, the combined
    abundances.  `run` calls it on one process' tables; the sharded runs of tests/oracle_util.py on the merged tables of the
    shards (class dicts merged in stream order are the dicts of the whole stream)."""
    out = {}
    t2 = time.perf_counter()
    n_iter = 0
    ems = []                     # per single_abundance call: (n_classes, n_iter, [(allele index, prob)])
    out["gene_classes"] = (gub, guc)
    if hla:
        out["exon_classes"] = (eub, euc)
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
        gene_prob, it = em_sorted(orc, t, gub, guc, False, None)
        ems.append((len(guc), it, gene_prob))
        n_iter += it
    t3 = time.perf_counter()
    out.update(t_em=t3 - t2, n_iter=n_iter, gene_prob=list(gene_prob), gene_counts=gc, first_pair=fp, em=ems)
    return out
