"""hgx_type_many (many samples of one locus behind one launch chain) against the one-task path: every field of every task's
result must be EQUAL -- counts and their order, class counts, iteration counts, and the EM abundances as doubles."""
import numpy as np
import pytest

import hisatgenotype_amd as hgx
from hisatgenotype_amd import engine, locus as hl, synth
import sys
htyping = sys.modules["hisatgenotype_amd.typing"]

pytestmark = pytest.mark.gpu


def _same(a, b):
    assert a.num_reads == b.num_reads and a.num_pairs == b.num_pairs
    if a.num_reads == 0:
        return
    assert np.array_equal(a.counts_order, b.counts_order)
    assert np.array_equal(a.counts, b.counts)
    assert len(a.em) == len(b.em)
    for x, y in zip(a.em, b.em):
        assert x["n_classes"] == y["n_classes"] and x["n_iter"] == y["n_iter"]
        assert x["remove_low"] == y["remove_low"] and x["use_length"] == y["use_length"]
        assert x["result"] == y["result"], (x["result"][:3], y["result"][:3])
    assert a.gene_prob == b.gene_prob


def _one(pl, batch, remove_low=True, em_fast=False):
    res = htyping.LocusResult()
    res.num_reads, res.num_pairs = batch.n_reads, batch.n_pairs
    if batch.n_reads <= 0:
        return res
    return htyping._type_batch(pl, batch, res, remove_low, em_fast=em_fast)


@pytest.mark.parametrize("n_alleles,n_vars,pairs", [(300, 500, [400, 900, 0, 1500, 37, 600]), (2500, 1500, [1200, 2500, 800]),
                                                     (7000, 2500, [3000, 1000])])
def test_many_equals_one_by_one_hla(n_alleles, n_vars, pairs):
    loc = synth.make_hla_like_locus(n_alleles=n_alleles, n_vars=n_vars, seed=31 + n_alleles)
    pl = hl.PackedLocus.from_synth(loc)
    batches = []
    for s, n in enumerate(pairs):
        sample = synth.pick_sample(loc, 50 + s)
        sam = synth.simulate_sam_fast(loc, sample, n, err_rate=0.003, seed=7 * s + 1) if n else ""
        batches.append(pl.parse_sam(sam))
    many = engine.ManyBatch(pl, batches)
    assert many.n_pairs == sum(b.n_pairs for b in batches) and many.n_pieces <= sum(b.n_pieces for b in batches)
    for low in (True, False):
        got = htyping.type_many(pl, many, remove_low=low, em_fast=False)        # the reference's order of operations (em_fast = 2)
        assert len(got) == len(batches)
        for g, b in zip(got, batches):
            _same(g, _one(pl, b, low))
    # throughput arithmetic (the many-task calls' DEFAULT: hgx_type_opts.em_fast = 0 there means table lookups): task for task
    # what the one-task path gives with em_fast = 1, and within 1e-8 of the reference-order results
    got = htyping.type_many(pl, many)
    for g, b in zip(got, batches):
        _same(g, _one(pl, b, True, em_fast=True))
    for g, h in zip(got, htyping.type_many(pl, many, em_fast=True)):
        _same(g, h)
    for g, b in zip(got, batches):
        want = _one(pl, b, True)
        assert [e["n_iter"] for e in g.em] == [e["n_iter"] for e in want.em]
        assert [a for a, _ in g.gene_prob] == [a for a, _ in want.gene_prob] or max(abs(p - q) for (_, p), (_, q) in zip(g.gene_prob, want.gene_prob)) <= 1e-8
        for (_, p), (_, q) in zip(sorted(g.gene_prob), sorted(want.gene_prob)):
            assert abs(p - q) <= 1e-8
    # the same call again on the same resident batch (buffers recycled, scratch reused): identical
    again = htyping.type_many(pl, many)
    for g, h in zip(again, htyping.type_many(pl, many)):
        _same(g, h)


def test_many_equals_one_by_one_str_locus():
    """CODIS-like STR loci (gene level only, EM without pruning; a task with ONE class is the reference's quirk Q3)."""
    loc = synth.make_str_like_locus(gene="TH01", unit="AATG", max_repeats=12, min_repeats=4, seed=5)
    pl = hl.PackedLocus.from_synth(loc)
    batches = []
    for s, n in enumerate([300, 50, 700, 2]):
        names = [a for a in loc.allele_names if "BACKBONE" not in a]
        sample = [names[2 + s], names[-2 - s]] if n > 2 else [names[1]]
        sam = synth.simulate_sam_fast(loc, sample, n, read_len=100, frag_len=(200, 300), err_rate=0.002, seed=3 * s + 2)
        batches.append(pl.parse_sam(sam))
    many = engine.ManyBatch(pl, batches)
    got = htyping.type_many(pl, many, return_errors=True, em_fast=False)
    for g, b in zip(got, batches):
        try:
            want = _one(pl, b)
        except TypeError:
            assert isinstance(g, TypeError)
            continue
        _same(g, want)


def _fuzz_tool():
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("fuzz_many", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_many.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("seed", [800012, 800028, 800044, 800020, 800058, 800101])
def test_many_fuzz_cases_with_tiny_tasks(seed):
    """Cases tools/fuzz_many.py found: a task of ONE pair whose hand-off leaves no class (the reference's loop over the empty dict
    still makes one pass: n_iter 1, empty result) typed beside ordinary tasks; and a CODIS task on which the reference raises
    (quirk Q3) that must fail alone.  Compared with the one-task path AND with oracle/pyref.py, `==`."""
    import pyref
    fm = _fuzz_tool()
    loc, sams = fm.make_case(seed)
    pl = hl.PackedLocus.from_synth(loc)
    batches = [pl.parse_sam(s) for s in sams]
    many = engine.ManyBatch(pl, batches)
    got = htyping.type_many(pl, many, return_errors=True, em_fast=False)
    for t, sam in enumerate(sams):
        try:
            one = htyping.type_locus(pl, sam) if sam else None
        except Exception as e:
            one = e
        try:
            exp = pyref.RefLocus(loc).run(sam) if sam else {"num_reads": 0}
        except Exception as e:
            exp = e
        s_many = fm.summary(got[t])
        assert fm.first_difference(s_many, (0, 0) if one is None else fm.summary(one)) is None, (seed, t)
        assert fm.first_difference(s_many, fm.summary(exp)) is None, (seed, t)
    many.close()


def test_many_with_only_empty_tasks():
    """No task has a class: no EM job at all (the launch is skipped, not an argument error)."""
    loc = synth.make_hla_like_locus(n_alleles=120, n_vars=200, seed=11)
    pl = hl.PackedLocus.from_synth(loc)
    batches = [pl.parse_sam(""), pl.parse_sam("")]
    many = engine.ManyBatch(pl, batches)
    got = htyping.type_many(pl, many)
    assert [g.num_reads for g in got] == [0, 0]
    rows = htyping.type_many_loci([pl], [many], light=True)
    assert [r[0] for r in rows[0]] == [0, 0]
    many.close()


def test_many_loci_in_the_any_size_mode():
    """em_fast = -1 through the many-task path with deep samples (more than 4 096 exon-level classes): each such EM gets a cluster
    launch of its own behind the ordinary launch; results `==` the one-task path in the same mode (itself `==` the C oracle: test_gpu_emx.py)."""
    pls, manies, batches = [], [], []
    for k, (A, V) in enumerate([(7000, 2500), (5000, 1800)]):
        loc = synth.make_hla_like_locus(n_alleles=A, n_vars=V, seed=60 + k, var_id_base=100000 * k)
        pl = hl.PackedLocus.from_synth(loc)
        sample = synth.pick_sample(loc, 3 + k)
        deep = pl.parse_sam(synth.simulate_sam_fast(loc, sample, 60000, err_rate=0.002, seed=19 + k))
        small = pl.parse_sam(synth.simulate_sam_fast(loc, sample, 700, err_rate=0.002, seed=29 + k))
        pls.append(pl); batches.append([deep, small]); manies.append(engine.ManyBatch(pl, [deep, small]))
    rows = htyping.type_many_loci(pls, manies, em_fast=-1)
    big = 0
    for pl, bs, row in zip(pls, batches, rows):
        for b, got in zip(bs, row):
            want = _one(pl, b, em_fast=-1)
            big += want.em[0]["n_classes"] > 4096
            _same(got, want)
    assert big >= 2
    for m in manies:
        m.close()


def test_gene_level_beside_em1_equals_gene_level_before_it():
    """The many-task calls score the gene level of an HLA locus and copy the Gene_counts while the launch of EM #1 runs
    (hgx_type.hip run_many); the test switch many=rest_first scores everything before the launch, as rounds 3-4 did: every field
    of every task equal, for an HLA locus and a locus of another base side by side (the latter has no second half)."""
    pls, manies = [], []
    loc = synth.make_hla_like_locus(n_alleles=2500, n_vars=1500, seed=77)
    pl = hl.PackedLocus.from_synth(loc)
    bs = [pl.parse_sam(synth.simulate_sam_fast(loc, synth.pick_sample(loc, 9 + s), n, err_rate=0.003, seed=5 * s + 2) if n else "")
          for s, n in enumerate([1500, 0, 700, 2600, 90])]
    pls.append(pl); manies.append(engine.ManyBatch(pl, bs))
    sloc = synth.make_str_like_locus(gene="TH01", unit="AATG", max_repeats=12, min_repeats=4, seed=5)
    spl = hl.PackedLocus.from_synth(sloc)
    names = [a for a in sloc.allele_names if "BACKBONE" not in a]
    sbs = [spl.parse_sam(synth.simulate_sam_fast(sloc, [names[2 + s], names[-2 - s]], 300 + 100 * s, read_len=100, frag_len=(200, 300), err_rate=0.002, seed=3 * s + 2))
           for s in range(3)]
    pls.append(spl); manies.append(engine.ManyBatch(spl, sbs))
    for kw in (dict(), dict(em_fast=False)):
        rows = htyping.type_many_loci(pls, manies, **kw)
        with engine.test_switches(many="rest_first"):
            rows0 = htyping.type_many_loci(pls, manies, **kw)
        for row, row0 in zip(rows, rows0):
            assert len(row) == len(row0)
            for g, h in zip(row, row0):
                _same(g, h)
        one = htyping.type_many(pls[0], manies[0], **kw)                   # (the one-locus entry point goes the same way)
        for g, h in zip(one, rows[0]):
            _same(g, h)
    for m in manies:
        m.close()
