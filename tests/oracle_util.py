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


def pool(n_tasks):
    """A spawn-context process pool (fork after HIP initialisation is unsafe) sized to the host."""
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    n = max(1, min(n_tasks, (os.cpu_count() or 2) - 1, 48))
    return ProcessPoolExecutor(max_workers=n, mp_context=mp.get_context("spawn"))
