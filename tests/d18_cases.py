"""CODIS D18S51 streams for the choose_pairs tests (tests/lab_front_cases.py on the CPU emulation, tests/test_gpu_front.py on the kernels)."""
from hisatgenotype_amd import synth


def d18s51_cases(n_seeds=4, pairs=400, step=9):
    """CODIS D18S51 samples with EVERY step-th pair moved to the end of the stream in turn: choose_pairs (typing_core.py:680-716) is
    applied to the stream's last pair only (1547-1552), and a pair of reads inside the repeat run carries several haplotypes per
    mate.  Yields (locus, the same locus under another base name -- no choose_pairs --, text)."""
    for seed in range(n_seeds):
        kw = dict(gene="D18S51", unit="AGAA", max_repeats=22, min_repeats=9, flank=180, seed=71 + seed)
        d18, plain = synth.make_str_like_locus(**kw), synth.make_str_like_locus(**kw)
        d18.base_fname, plain.base_fname = "codis", "notcodis"
        dn = [a for a in d18.allele_names if "BACKBONE" not in a]
        sam = synth.simulate_sam_fast(d18, [dn[3 + seed], dn[-3 - seed]], pairs, read_len=100, frag_len=(200, 280), err_rate=0.002, seed=17 + seed)
        groups = []
        for l in sam.split("\n"):
            if not l:
                continue
            q = l.split("\t")[0]
            if groups and groups[-1][0] == q:
                groups[-1][1].append(l)
            else:
                groups.append((q, [l]))
        for k in range(seed, len(groups), step):
            g = groups[:k] + groups[k + 1:] + [groups[k]]
            yield d18, plain, "\n".join(l for _, ls in g for l in ls) + "\n"
