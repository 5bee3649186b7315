"""Oracle runs for the config-sized GPU tests (TEST INFRASTRUCTURE): the pure-Python front-end restatement (oracle/pyref.py,
pinned to the reference per record) feeding the C oracle (oracle/hgx_oracle.c, pinned to the reference's classes and EM
floats) -- add_count / add_stat / dict accumulation / single_abundance / hand-off in the reference's own order.

`oracle_type` is a module-level function of plain arguments so that a spawn-ed process pool can run many (sample, locus)
tasks beside each other on the GPU box's host cores; it never touches the GPU or libhgx."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def oracle_type(locus_json, sam, simulation=False, allow_discordant=False, remove_low=True):
    """-> dict(num_reads, num_pairs, gene_counts [A] int64, em [(n_classes, n_iter, [(allele name, prob)])],
    gene_prob [(name, prob)], exon_classes / gene_classes (bits [C][w64], counts [C]) in dict order)."""
    import numpy as np
    import hisatgenotype_amd  # noqa: F401  (registers the hyphenated package)
    from hisatgenotype_amd import synth
    import orc_pipeline
    import orclib
    import pyref
    import tables
    loc = synth.Locus.from_json(locus_json)
    rl = pyref.RefLocus(loc, allow_discordant=allow_discordant, simulation=simulation)
    rl.score = False                                       # front-end only: every pair's add_count arguments
    fe = rl.run(sam)
    t = tables.oracle_tables(loc)
    arrs = tables.pieces_from_pairs(fe["pairs"], t["var_index"])
    lengths = np.array([loc.allele_length(n) for n in t["names"]], dtype=np.int32)
    hla = loc.base_fname == "hla"
    out = {"num_reads": fe["num_reads"], "num_pairs": fe["num_pairs"]}
    if fe["num_reads"] == 0:
        return out
    r = orc_pipeline.run(orclib.load(), t, arrs, hla, lengths, remove_low=remove_low)
    names = t["names"]
    out.update(gene_counts=r["gene_counts"], first_pair=r["first_pair"],
               em=[(c, it, [(names[a], p) for a, p in res]) for c, it, res in r["em"]],
               gene_prob=[(names[a], p) for a, p in r["gene_prob"]],
               gene_classes=r["gene_classes"], exon_classes=r.get("exon_classes"))
    return out


# ---- the same chain over ONE big sample, shard by shard on the host cores (BASELINE configs[1] at its stated size) ---------------
# pyref does ~600 reads/s per core, ~2 000 with its per-key memo: a million reads are tens of core-minutes.  The stream is cut at
# read-id boundaries; what crosses shards in the reference's loop is (1) the pileup every record is corrected against -- counted per
# shard, added up, handed to every shard (pileup_tables) -- and (2) the class dicts, whose first-seen order and counts over the
# whole stream are the shards' dicts merged in stream order.  Gene_counts add up; "first pair that counted an allele" is the minimum.

def _shard_pileup(locus_json, sam, allow_discordant):
    import hisatgenotype_amd  # noqa: F401
    from hisatgenotype_amd import synth
    import pyref
    loc = synth.Locus.from_json(locus_json)
    rl = pyref.RefLocus(loc, allow_discordant=allow_discordant)
    return pyref.pileup_counts(rl.pileup_records(sam), len(rl.ref_seq), allow_discordant)


def _shard_classes(locus_json, sam, counts, simulation, allow_discordant):
    import numpy as np
    import hisatgenotype_amd  # noqa: F401
    from hisatgenotype_amd import synth
    import orclib
    import pyref
    import tables
    loc = synth.Locus.from_json(locus_json)
    rl = pyref.RefLocus(loc, allow_discordant=allow_discordant, simulation=simulation)
    rl.score = False
    rl.memo = {}
    fe = rl.run(sam, pileup_tables=(counts, pyref.nt_sets_from_counts(counts)))
    t = tables.oracle_tables(loc)
    arrs = tables.pieces_from_pairs(fe["pairs"], t["var_index"])
    orc = orclib.load()
    hla = loc.base_fname == "hla"
    eb, gb, gc, fp = orc.score_pairs(orc.make_locus(t), t["exon_keys"], t["gene_keys"], *arrs)
    out = {"num_reads": fe["num_reads"], "num_pairs": fe["num_pairs"], "gene_counts": gc, "first_pair": fp,
           "gene": orc.dedup(gb)[:2], "exon": orc.dedup(eb)[:2] if hla else None}
    return out


def split_name_grouped(sam, n):
    """`sam` cut into at most n runs of whole lines, never inside a run of equal read names."""
    lines = [l for l in sam.split("\n") if l and not l.startswith("@")]
    cuts = [0]
    for k in range(1, n):
        i = max(cuts[-1], len(lines) * k // n)
        while 0 < i < len(lines) and lines[i].split("\t", 1)[0] == lines[i - 1].split("\t", 1)[0]:
            i += 1
        cuts.append(i)
    cuts.append(len(lines))
    return ["\n".join(lines[a:b]) + "\n" for a, b in zip(cuts, cuts[1:]) if b > a]


def oracle_type_sharded(loc, sam, n_shards=None, simulation=False, allow_discordant=False, remove_low=True, executor=None):
    """oracle_type over a big sample: same keys in the result (plus nothing else), the work spread over a process pool."""
    import numpy as np
    import orc_pipeline
    import orclib
    import tables
    n_shards = n_shards or max(1, min((os.cpu_count() or 2) - 1, 48))      # one shard per worker: the per-key memo pays best on long shards
    shards = split_name_grouped(sam, n_shards)
    lj = loc.to_json()
    ex = executor or pool(len(shards))
    try:
        parts = list(ex.map(_shard_pileup, [lj] * len(shards), shards, [allow_discordant] * len(shards)))
        counts = parts[0]
        for p in parts[1:]:
            for tot, d in zip(counts, p):
                for nt, c in d.items():
                    tot[nt] = tot.get(nt, 0) + c
        res = list(ex.map(_shard_classes, [lj] * len(shards), shards, [counts] * len(shards), [simulation] * len(shards),
                          [allow_discordant] * len(shards)))
    finally:
        if executor is None:
            ex.shutdown(cancel_futures=True)
    orc = orclib.load()
    t = tables.oracle_tables(loc)
    hla = loc.base_fname == "hla"
    out = {"num_reads": sum(r["num_reads"] for r in res), "num_pairs": sum(r["num_pairs"] for r in res)}
    if out["num_reads"] == 0:
        return out
    gc = np.sum([r["gene_counts"] for r in res], axis=0)
    fp = np.full(len(gc), -1, np.int64)
    base = 0
    for r in res:                                                    # first pair (stream order) whose class held the allele
        f = r["first_pair"].astype(np.int64)
        new = (fp < 0) & (f >= 0)
        fp[new] = f[new] + base
        base += r["num_pairs"]

    def merged(key):                                                 # the shards' class dicts, merged in stream order
        bits = np.concatenate([r[key][0] for r in res])
        cnt = np.concatenate([r[key][1] for r in res])
        ub, uc, _ = orc.dedup(bits, weight=cnt)
        return ub, uc
    gub, guc = merged("gene")
    eub, euc = merged("exon") if hla else (None, None)
    lengths = np.array([loc.allele_length(n) for n in t["names"]], dtype=np.int32)
    r = orc_pipeline.finish(orc, t, eub, euc, gub, guc, gc, fp.astype(np.int32), hla, lengths, remove_low=remove_low)
    names = t["names"]
    out.update(gene_counts=r["gene_counts"], first_pair=r["first_pair"],
               em=[(c, it, [(names[a], p) for a, p in res_]) for c, it, res_ in r["em"]],
               gene_prob=[(names[a], p) for a, p in r["gene_prob"]],
               gene_classes=r["gene_classes"], exon_classes=r.get("exon_classes"))
    return out


def pool(n_tasks):
    """A spawn-context process pool (fork after HIP initialisation is unsafe) sized to the host."""
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    n = max(1, min(n_tasks, (os.cpu_count() or 2) - 1, 48))
    return ProcessPoolExecutor(max_workers=n, mp_context=mp.get_context("spawn"))
