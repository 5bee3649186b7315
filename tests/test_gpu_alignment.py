"""One alignment file, many loci (typing_core.py:370 loops `locus_list` over ONE alignment file): hgx_alignment_open reads the file once
-- its bytes stay in HBM, a BAM inflated on the device -- and hgx_alignment_parse_dev pulls one locus at a time out of it by region,
as kernels.  Every golden fixture recorded from the real reference, concatenated into ONE multi-reference file (SAM text, a
name-grouped BAM, a coordinate-sorted BAM), must come out of that file with the reference's recorded EM results and report text, the
device front end asserted; typing() on a three-locus file writes the report the per-locus calls write, side by side or one by one."""
import os
import sys
import threading

import numpy as np
import pytest

import golden_util as gu
import hisatgenotype_amd as hgx
from hisatgenotype_amd import bamio, capi, engine, locus as hl, synth

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import pyref  # noqa: E402

pytestmark = pytest.mark.gpu
EM_TOL = 1e-9
NAMES = gu.ALL + gu.LEAN


def _rname(k):
    return "L%02d*BACKBONE" % k


def _multi_locus_sam():
    """Every fixture's records with RNAME = its own reference: one stream, the fixtures one after the other."""
    lines, refs = [], []
    for k, name in enumerate(NAMES):
        fx = gu.load(name)
        refs.append((_rname(k), len(fx["_locus"].backbone)))
        for l in fx["sam"].split("\n"):
            if l and not l.startswith("@"):
                f = l.split("\t")
                f[2] = _rname(k)
                lines.append("\t".join(f))
    return "\n".join(lines) + "\n", refs


def _check_fixture(name, res):
    fx = gu.load(name)
    o = fx["options"]
    assert len(res.em) == len(fx["em"]), name
    for got, exp in zip(res.em, fx["em"]):
        assert got["n_classes"] == len(exp["cmpt"]) and got["n_iter"] == exp["n_iter"], name
        assert [a for a, _ in got["result"]] == [a for a, _ in exp["result"]], name
        for (_, p), (_, q) in zip(got["result"], exp["result"]):
            assert abs(p - float(q)) <= EM_TOL, name
    lines, _ = hgx.report_lines(res, o["simulation"], o["sample"] if o["simulation"] else (), True)
    keep = lambda ls: [l for l in ls if "aligned" in l or "ranked" in l or "(count:" in l]
    assert keep(lines) == keep(fx["report"].split("\n")), name


@pytest.fixture(scope="module")
def multi_files(tmp_path_factory):
    d = tmp_path_factory.mktemp("multi")
    sam, refs = _multi_locus_sam()
    paths = {"sam": str(d / "all.sam"), "bam": str(d / "all.bam"), "sorted_bam": str(d / "all.sorted.bam")}
    with open(paths["sam"], "w") as f:
        f.write("".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs) + sam)
    bamio.write_bam_native(paths["bam"], sam.encode(), refs)
    bamio.write_bam_native(paths["sorted_bam"], sam.encode(), refs, sort_by_coordinate=True)
    return paths


@pytest.mark.parametrize("kind", ["sam", "bam", "sorted_bam"])
def test_every_fixture_out_of_one_multi_locus_file(multi_files, kind):
    capi.set_device(0)
    with engine.Alignment(multi_files[kind]) as al:
        assert al.resident and al.is_text == (kind == "sam") and al.stream_bytes > 3_000_000
        sent = al.bytes_to_device
        assert sent > 0 and (kind == "sam" or sent < al.stream_bytes / 2)          # a BAM travels deflated and is inflated on the device
        for k, name in enumerate(NAMES):
            fx = gu.load(name)
            o = fx["options"]
            pl = hl.PackedLocus.from_synth(fx["_locus"])
            res = hgx.type_locus(pl, None, alignment=al, regions=[_rname(k)], num_editdist=o["num_editdist"], error_correction=o["error_correction"],
                                 allow_discordant=o["allow_discordant"], remove_low_abundance_alleles=o["remove_low"], simulation=o["simulation"])
            assert engine.front_last() == (2, 0), (name, engine.front_last())
            _check_fixture(name, res)
            pl.close()


def test_loci_side_by_side_from_threads(multi_files):
    """The resident bytes are read-only: every fixture's locus parsed and typed at the same time, a thread and stream each."""
    capi.set_device(0)
    out, errs = {}, []
    with engine.Alignment(multi_files["sorted_bam"]) as al:
        def work(k, name):
            try:
                capi.set_device(0)
                capi.set_stream_slot(("alignment test", k))
                st = capi.get_stream(2)
                fx = gu.load(name)
                o = fx["options"]
                pl = hl.PackedLocus.from_synth(fx["_locus"])
                out[name] = hgx.type_locus(pl, None, alignment=al, regions=[_rname(k)], num_editdist=o["num_editdist"],
                                           error_correction=o["error_correction"], allow_discordant=o["allow_discordant"],
                                           remove_low_abundance_alleles=o["remove_low"], simulation=o["simulation"], stream=st)
                capi.sync(st)
                pl.close()
            except BaseException as e:      # noqa: BLE001 (re-raised below)
                errs.append((name, e))
        ths = [threading.Thread(target=work, args=(k, n)) for k, n in enumerate(NAMES)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
    assert not errs, errs
    for name in NAMES:
        _check_fixture(name, out[name])


def test_small_files_and_odd_requests_take_the_per_path_call(tmp_path):
    capi.set_device(0)
    fx = gu.load("hla_small_pair")
    pl = hl.PackedLocus.from_synth(fx["_locus"])
    p = str(tmp_path / "small.sam")
    open(p, "w").write(fx["sam"])
    with engine.Alignment(p) as al:
        assert not al.resident                                      # below the size gate: not even read at open
        res = hgx.type_locus(pl, None, alignment=al, regions=[pl.ref_allele], simulation=True)
        assert engine.front_last() == (0, 6)
    ref = hgx.type_locus(pl, fx["sam"], simulation=True)
    assert res.gene_prob == ref.gene_prob and res.counts_sorted == ref.counts_sorted
    with engine.test_switches(front="device"):                      # forced: resident, the kernels take it
        with engine.Alignment(p) as al:
            assert al.resident
            res = hgx.type_locus(pl, None, alignment=al, regions=[pl.ref_allele], simulation=True)
            assert engine.front_last() == (2, 0)
            # two regions in one request are not deferrable: the per-path call (which reads the file the ordinary way)
            res2 = hgx.type_locus(pl, None, alignment=al, regions=[pl.ref_allele, "nowhere"], simulation=True)
    assert res.gene_prob == ref.gene_prob and res.counts_sorted == ref.counts_sorted and res2.gene_prob == ref.gene_prob
    with pytest.raises(capi.HgxError):
        engine.Alignment(str(tmp_path / "missing.bam"))


import functools  # noqa: E402


@functools.lru_cache(maxsize=None)
def _three_loci_data(n_pairs):
    specs = [("A", 600, 3569, 1300, 0), ("B", 800, 4081, 1500, 5000), ("C", 500, 4305, 1200, 10000)]
    loci, sams, expect = [], [], {}
    for k, (gene, n_all, length, n_vars, base) in enumerate(specs):
        loc = synth.make_hla_like_locus(gene=gene, n_alleles=n_all, length=length, n_vars=n_vars, seed=300 + k, var_id_base=base)
        sample = synth.pick_sample(loc, 40 + k)
        sam = synth.simulate_sam_fast(loc, sample, n_pairs + 100 * k, err_rate=0.003, seed=50 + k)
        expect[gene] = pyref.RefLocus(loc).run(sam)
        sams.append(sam)
        loci.append(loc)
    return loci, sams, expect


def _three_loci(tmp_path, n_pairs=1500):
    loci, sams, expect = _three_loci_data(n_pairs)
    refs = [(loc.ref_allele, len(loc.backbone)) for loc in loci]
    bam = str(tmp_path / "sample.bam")
    bamio.write_bam_native(bam, "".join(sams).encode(), refs, sort_by_coordinate=True)
    d = {k: {} for k in ("refGenes", "Genes", "Gene_names", "Gene_lengths", "refGene_loci", "Vars", "Var_list", "Links")}
    for loc in loci:
        for k, v in loc.reference_dicts().items():
            d[k].update(v)
    return loci, bam, d, expect


@pytest.mark.parametrize("side_by_side", [True, False, "together"])
def test_typing_three_loci_from_one_file(tmp_path, side_by_side):
    """typing() (38 arguments, typing_core.py:249-286) with locus_list = [A, B, C] on ONE coordinate-sorted BAM of all three loci: the
    file is opened once, every locus goes through the device front end, and the report holds the three sections in locus_list order
    -- counts, abundances and read totals equal to the oracle's per-locus results."""
    capi.set_device(0)
    loci, bam, d, expect = _three_loci(tmp_path)
    T = sys.modules["hisatgenotype_amd.typing"]
    old = T.typing_options.loci_side_by_side
    T.typing_options.loci_side_by_side = bool(side_by_side)
    T.typing_options.loci_together = side_by_side == "together"      # (parsed side by side, typed by ONE hgx_type_many_loci call: hgx_many_from_dbatch)
    try:
        hgx.typing(False, str(tmp_path / "hla"), ["A", "B", "C"], "", True, set(), d["refGenes"], d["Genes"], d["Gene_names"], d["Gene_lengths"],
                   d["refGene_loci"], d["Vars"], d["Var_list"], d["Links"], [["hisat2", "graph"]], 2, False, "assembly_graph", True, True, False,
                   False, True, [], False, ["sample.fq"], bam, [], 150, 400, 1, False, 0, False, str(tmp_path), "NONE", True, 0)
    finally:
        T.typing_options.loci_side_by_side = old
        T.typing_options.loci_together = False
    assert [p["gene"] for p in T.last_profile] == ["A", "B", "C"]
    assert all(p["front_end_route"] == [2, 0] for p in T.last_profile), T.last_profile
    rep = open(str(tmp_path / "assembly_graph-hla.sample.report")).read().split("\n")
    got_counts = [l.strip() for l in rep if "(count:" in l]
    want_counts = ["%d %s (count: %d)" % (i + 1, n, c) for g in "ABC" for i, (n, c) in enumerate(expect[g]["counts_sorted"])]
    assert got_counts == want_counts
    assert [l.strip() for l in rep if "aligned" in l] == ["%d reads and %d pairs are aligned" % (expect[g]["num_reads"], expect[g]["num_pairs"]) for g in "ABC"]
    want_ab = []
    for g in "ABC":
        for i, (n, p) in enumerate(expect[g]["gene_prob"]):
            if p < 0.01 or i >= 10:
                break
            want_ab.append("%d ranked %s (abundance: %.2f%%)" % (i + 1, n, p * 100.0))
    assert [l.strip() for l in rep if "abundance" in l] == want_ab


def test_open_once_is_the_same_batch_as_the_per_path_call(tmp_path):
    """Array for array: the batch a locus gets out of the resident file is the batch hgx_parse_alignment_file_dev builds from the path."""
    capi.set_device(0)
    loci, bam, d, _ = _three_loci(tmp_path)
    with engine.Alignment(bam) as al:
        for loc in loci:
            pl = hl.PackedLocus.from_synth(loc)
            a = al.parse_dev(pl, [loc.ref_allele])
            assert engine.front_last() == (2, 0)
            b = pl.parse_alignment_file_dev(bam, [loc.ref_allele])
            ha, hb = a.to_host(), b.to_host()
            for k in ("pieces", "masks", "pair_off", "pair_ref"):
                assert np.array_equal(getattr(ha, k), getattr(hb, k)), (loc.gene, k)
            assert (ha.n_reads, ha.n_pairs) == (hb.n_reads, hb.n_pairs)
            pl.close()


def test_typing_options_table_lookup_em_writes_the_same_report(tmp_path):
    """hgx.typing_options.em_fast = True (table-lookup EM, <= 1e-8 of the reference's doubles): the report -- counts, ranks, abundances at the
    reference's two decimals -- is the default arithmetic's, text for text."""
    capi.set_device(0)
    loci, bam, d, expect = _three_loci(tmp_path)
    T = sys.modules["hisatgenotype_amd.typing"]
    reps = []
    for fast in (False, True):
        T.typing_options.em_fast = fast
        try:
            hgx.typing(False, str(tmp_path / "hla"), ["A", "B", "C"], "", True, set(), d["refGenes"], d["Genes"], d["Gene_names"], d["Gene_lengths"],
                       d["refGene_loci"], d["Vars"], d["Var_list"], d["Links"], [["hisat2", "graph"]], 2, False, "assembly_graph", True, True, False,
                       False, True, [], False, ["sample.fq"], bam, [], 150, 400, 1, False, 0, False, str(tmp_path), "NONE", True, 0)
        finally:
            T.typing_options.em_fast = False
        reps.append([l for l in open(str(tmp_path / "assembly_graph-hla.sample.report")).read().split("\n") if not l.startswith("#")])
    assert reps[0] == reps[1] and sum("abundance" in l for l in reps[0]) >= 6


@pytest.mark.parametrize("side_by_side", [True, False])
def test_typing_raises_at_the_failing_locus_after_the_reports_before_it(tmp_path, side_by_side):
    """Three CODIS loci in one file; the SECOND sample is homozygous with identical reads: ONE compatibility class, on which the reference
    raises TypeError (quirk Q3, typing_core.py:1784-1787).  Side by side or locus after locus, typing() raises that TypeError -- and the
    report holds the first locus' section, not the third's (the reference's loop never got there)."""
    capi.set_device(0)
    specs = [("D8S1179", "TCTA", 19, 7), ("TH01", "AATG", 12, 4), ("FGA", "CTTT", 30, 16)]
    loci, sams = [], []
    for k, (gene, unit, mx, mn) in enumerate(specs):
        loc = synth.make_str_like_locus(gene=gene, unit=unit, max_repeats=mx, min_repeats=mn, flank=170, seed=60 + k, var_id_base=100 * k)
        names = [a for a in loc.allele_names if "BACKBONE" not in a]
        sample = [names[1]] if k == 1 else [names[2], names[-3]]
        sams.append(synth.simulate_sam_fast(loc, sample, 1 if k == 1 else 900, read_len=100, frag_len=(200, 300), err_rate=0.0 if k == 1 else 0.002, seed=3 * k + 2))
        loci.append(loc)
    one = hl.PackedLocus.from_synth(loci[1])
    with pytest.raises(TypeError):
        hgx.type_locus(one, sams[1])                            # (the case is what it is meant to be)
    refs = [(loc.ref_allele, len(loc.backbone)) for loc in loci]
    sam_path = str(tmp_path / "s.sam")
    with open(sam_path, "w") as f:
        f.write("".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs) + "".join(sams))
    d = {k: {} for k in ("refGenes", "Genes", "Gene_names", "Gene_lengths", "refGene_loci", "Vars", "Var_list", "Links")}
    for loc in loci:
        for k, v in loc.reference_dicts().items():
            d[k].update(v)
    T = sys.modules["hisatgenotype_amd.typing"]
    T.typing_options.loci_side_by_side = side_by_side
    try:
        with engine.test_switches(front="device"):              # (the file is small: forced resident, so that the side-by-side form is what runs)
            with pytest.raises(TypeError):
                hgx.typing(False, str(tmp_path / "codis"), [loc.gene for loc in loci], "", True, set(), d["refGenes"], d["Genes"], d["Gene_names"],
                           d["Gene_lengths"], d["refGene_loci"], d["Vars"], d["Var_list"], d["Links"], [["hisat2", "graph"]], 2, False, "assembly_graph",
                           True, True, False, False, True, [], False, ["sample.fq"], sam_path, [], 100, 250, 1, False, 0, False, str(tmp_path), "NONE", True, 0)
    finally:
        T.typing_options.loci_side_by_side = True
    rep = open(str(tmp_path / "assembly_graph-codis.sample.report")).read()
    assert "D8S1179" in rep and "FGA*" not in rep and rep.count("pairs are aligned") == 1
