"""Integer tables of a locus for the C oracle, derived with the Python oracle.  TEST INFRASTRUCTURE."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pyref  # noqa: E402


def oracle_tables(locus):
    names = [n for n in locus.allele_names if n.find("BACKBONE") == -1]
    aidx = {n: i for i, n in enumerate(names)}
    V = len(locus.var_ids)
    var_pos = np.array(locus.var_pos, dtype=np.int32)
    var_right = np.array([p + int(d) - 1 if t == "deletion" else p
                          for t, p, d in zip(locus.var_type, locus.var_pos, locus.var_data)], dtype=np.int32)
    linked = np.array([1 if v in locus.links else 0 for v in locus.var_ids], dtype=np.uint8)
    off = np.zeros(V + 1, dtype=np.int32)
    flat = []
    for i, v in enumerate(locus.var_ids):
        al = [aidx[a] for a in locus.links.get(v, []) if a in aidx]
        flat += al
        off[i + 1] = len(flat)
    d = locus.reference_dicts()
    ev = pyref.exonic_vars(d["Vars"][locus.gene], locus.exons)
    reps, groups = pyref.rep_alleles(d["Links"], ev)
    rep_set = set(reps.values())
    exon_keys = np.array([1 if n in rep_set else 0 for n in names], dtype=np.uint8)
    gene_keys = np.ones(len(names), dtype=np.uint8)
    return dict(n_alleles=len(names), names=names, aidx=aidx, var_pos=var_pos, var_right=var_right,
                var_linked=linked, link_off=off, link_allele=np.array(flat or [0], dtype=np.int32),
                exon_keys=exon_keys, gene_keys=gene_keys, var_index={v: i for i, v in enumerate(locus.var_ids)},
                rep_groups=groups, reps=reps)


def pieces_from_pairs(pairs, var_index):
    """golden/pyref pair log -> flat piece arrays for orc_score_pairs."""
    def parse_ht(ht, var_index):
        f = ht.split("-")
        return int(f[0]), int(f[-1]), [var_index.get(v, -1) if v.startswith("hv") else -1 for v in f[1:-1]]
    pair_off, level, left, right, id_off, ids = [0], [], [], [], [0], []
    for p in pairs:
        for lvl, key in ((0, "exon"), (1, "gene")):
            for ht in p[key]:
                l, r, vs = parse_ht(ht, var_index)
                level.append(lvl); left.append(l); right.append(r)
                ids += vs
                id_off.append(len(ids))
        pair_off.append(len(level))
    return (np.array(pair_off, np.int32), np.array(level, np.uint8), np.array(left, np.int32),
            np.array(right, np.int32), np.array(id_off, np.int32), np.array(ids or [0], np.int32))
