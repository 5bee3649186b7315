"""End-to-end on the GPU: SAM text -> report, against the reference's recorded outputs and the Python oracle."""
import os
import sys

import numpy as np
import pytest

import golden_util as gu
import hisatgenotype_amd as hgx
from hisatgenotype_amd import engine, locus as hl, synth

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import pyref  # noqa: E402

pytestmark = pytest.mark.gpu

EM_TOL = 1e-5      # north_star: EM float weights within 1e-5 (observed: ~1e-15)


def _check_em(got, exp_result, exp_iter, exact=False):
    assert got["n_iter"] == exp_iter
    assert [a for a, _ in got["result"]] == [a for a, _ in exp_result]
    for (a, p), (_, q) in zip(got["result"], exp_result):
        assert abs(p - float(q)) <= EM_TOL
        assert abs(p - float(q)) <= 1e-9
        if exact:      # the hand-off EM follows the reference's summation order: the same bits
            assert p == float(q), (a, repr(p), q)


# The front-end routes a result is asked through: None = the library's choice (these fixtures are below the 20 000-record
# gate: the host stages), "device" = the RECORD route (fields, filters, key grouping, pileup, decode, piece table, pair protocol all as
# kernels: k_fe_*), "device,keys" = the KEY route (the host tokenises and groups, the kernels take the distinct keys).  The reference's
# recorded class dicts, EM doubles and report text are compared with what came out of the route asked for -- `engine.front_last()` is
# asserted -- not with the host front end's batch.
FRONTS = [None, "device", "device,keys"]
FRONT_IDS = ["default", "record_route", "key_route"]


class _front:
    """Force a front-end route for the calls inside the block and check afterwards that it was the one that ran."""
    def __init__(self, front, fixture_name=None):
        self.front, self.name = front, fixture_name
        self.sw = engine.test_switches(front=front) if front else None

    def __enter__(self):
        if self.sw:
            self.sw.__enter__()
        return self

    def check(self):
        if not self.front:
            return
        assert engine.front_last() == (2 if self.front == "device" else 1, 0), engine.front_last()      # (codis_d18s51 too, since round 6)

    def __exit__(self, *exc):
        if self.sw:
            self.sw.__exit__(*exc)
        return False


@pytest.mark.parametrize("front", FRONTS, ids=FRONT_IDS)
@pytest.mark.parametrize("name", gu.ALL)
def test_type_locus_matches_reference(name, front):
    fx = gu.load(name)
    o = fx["options"]
    pl = hl.PackedLocus.from_synth(fx["_locus"])
    with _front(front, name) as f:
        res = hgx.type_locus(pl, fx["sam"], num_editdist=o["num_editdist"], error_correction=o["error_correction"],
                             allow_discordant=o["allow_discordant"], remove_low_abundance_alleles=o["remove_low"],
                             simulation=o["simulation"])
        f.check()
    assert res.num_reads == len(fx["records"]) and res.num_pairs == len(fx["pairs"])
    assert len(res.em) == len(fx["em"])
    for k, (got, exp) in enumerate(zip(res.em, fx["em"])):
        assert got["n_classes"] == len(exp["cmpt"])
        assert got["remove_low"] == exp["remove_low"] and got["use_length"] == exp["use_length"]
        small = len(exp["cmpt"]) <= 64 and len({a for cid, _ in exp["cmpt"] for a in gu.class_key(fx, cid).split("-")}) <= 64
        _check_em(got, exp["result"], exp["n_iter"], exact=(k == 1 and got["use_length"]) or small)
    lines, _ = hgx.report_lines(res, o["simulation"], o["sample"] if o["simulation"] else (), True)
    keep = lambda ls: [l for l in ls if "aligned" in l or "ranked" in l or "(count:" in l]
    assert keep(lines) == keep(fx["report"].split("\n"))


@pytest.mark.parametrize("front", FRONTS, ids=FRONT_IDS)
def test_config0_ten_thousand_reads_matches_reference(front):
    """BASELINE configs[0]: HLA-A-like, 7 000 alleles, 10 k reads -- the run the reference itself needed 109 s for
    (fixture `hla_7000_10k`, recorded from the real reference with its wall time).  Same classes and counts going into both
    EM calls, same iteration counts, allele order and abundances, and the same report: with output_allele_counts the
    report lists EVERY allele with a non-zero count, i.e. the complete integer Gene_counts table."""
    fx = gu.load("hla_7000_10k")
    o = fx["options"]
    pl = hl.PackedLocus.from_synth(fx["_locus"])
    with _front(front) as f:
        res = hgx.type_locus(pl, fx["sam"], num_editdist=o["num_editdist"], error_correction=o["error_correction"],
                             allow_discordant=o["allow_discordant"], remove_low_abundance_alleles=o["remove_low"],
                             simulation=o["simulation"], keep_classes=True)
        f.check()
    assert len(res.em) == len(fx["em"]) == 2
    A = pl.n_alleles
    for k, (got, exp) in enumerate(zip(res.em, fx["em"])):
        assert got["n_classes"] == len(exp["cmpt"])
        assert got["remove_low"] == exp["remove_low"] and got["use_length"] == exp["use_length"]
        _check_em(got, exp["result"], exp["n_iter"], exact=got["use_length"])
        if k == 0:                     # EM #1's input: the exon-level class dict, bit rows + counts in dict order
            bits, cnt = res.exon_classes
            want = np.stack([gu.class_bits(fx, cid, A) for cid, _ in exp["cmpt"]])
            assert np.array_equal(bits[:, :want.shape[1]], want) and not bits[:, want.shape[1]:].any()
            assert cnt.tolist() == [n for _, n in exp["cmpt"]]
    lines, _ = hgx.report_lines(res, False, (), True)
    keep = lambda ls: [l for l in ls if "aligned" in l or "ranked" in l or "(count:" in l]
    assert keep(lines) == keep(fx["report"].split("\n"))
    assert len([l for l in lines if "(count:" in l]) > 1000


@pytest.mark.parametrize("front", FRONTS, ids=FRONT_IDS)
def test_codis_ten_thousand_reads_matches_reference(front):
    """BASELINE configs[4]'s shape against the REAL reference (fixture `codis_10k`: one CODIS STR ladder, 10 k reads, recorded
    with the reference's wall time): same class dict into the EM, bit-identical abundances (13 alleles: one wavefront in the
    reference's order), same iteration count and the same report lines."""
    fx = gu.load("codis_10k")
    o = fx["options"]
    pl = hl.PackedLocus.from_synth(fx["_locus"])
    with _front(front) as f:
        res = hgx.type_locus(pl, fx["sam"], num_editdist=o["num_editdist"], error_correction=o["error_correction"],
                             allow_discordant=o["allow_discordant"], remove_low_abundance_alleles=o["remove_low"],
                             simulation=o["simulation"], keep_classes=True)
        f.check()
    assert len(res.em) == len(fx["em"])
    for got, exp in zip(res.em, fx["em"]):
        assert got["n_classes"] == len(exp["cmpt"])
        _check_em(got, exp["result"], exp["n_iter"], exact=True)
    lines, _ = hgx.report_lines(res, False, (), True)
    keep = lambda ls: [l for l in ls if "aligned" in l or "ranked" in l or "(count:" in l]
    assert keep(lines) == keep(fx["report"].split("\n"))


@pytest.mark.parametrize("name", ["hla_small_pair", "hla_mid_real", "codis_like"])
def test_single_abundance_dropin(name):
    """hgx.single_abundance takes the reference's dict-of-strings and returns its list-of-lists."""
    fx = gu.load(name)
    loc = fx["_locus"]
    lengths = {n: loc.allele_length(n) for n in loc.allele_names[1:]}
    for em in fx["em"]:
        cmpt = {gu.class_key(fx, cid): n for cid, n in em["cmpt"]}
        out = hgx.single_abundance(cmpt, em["remove_low"], lengths if em["use_length"] else {})
        assert [a for a, _ in out] == [a for a, _ in em["result"]]
        small = len(cmpt) <= 64 and len({a for key in cmpt for a in key.split("-")}) <= 64
        exact = engine.em_last_exact()                   # ran in the reference's own order (k_em_wave / k_em_ref)
        assert exact or not small
        for (a, p), (_, q) in zip(out, em["result"]):
            assert abs(p - float(q)) <= 1e-9
            if exact:      # one wavefront or one workgroup, the reference's own summation order
                assert p == float(q)


def test_seeded_mid_size_against_python_oracle():
    """Bigger than the fixtures (the reference cannot run on the GPU box): compare with oracle/pyref.py."""
    loc = synth.make_hla_like_locus(n_alleles=900, n_vars=1500, seed=202, unlinked_vars=3)
    sample = synth.pick_sample(loc, 7)
    al = synth.simulate_pairs(loc, sample, 1200, err_rate=0.004, seed=11, softclip_frac=0.03, novel_del_frac=0.02,
                              multi_hit_frac=0.01, dup_frac=0.01)
    sam = synth.sam_text(loc, al)
    rl = pyref.RefLocus(loc)
    exp = rl.run(sam)
    pl = hl.PackedLocus.from_synth(loc)
    res = hgx.type_locus(pl, sam)
    assert (res.num_reads, res.num_pairs) == (exp["num_reads"], exp["num_pairs"])
    assert res.counts_sorted == exp["counts_sorted"]
    assert len(res.em) == len(exp["em"])
    for got, e in zip(res.em, exp["em"]):
        _check_em(got, e["result"], e["n_iter"])
    assert [a for a, _ in res.gene_prob[:2]] == [a for a, _ in exp["gene_prob"][:2]]
    for (a, p), (b, q) in zip(res.gene_prob, exp["gene_prob"]):
        assert a == b and abs(p - q) <= 1e-9


def test_typing_signature_writes_report(tmp_path):
    """typing() keeps the reference's 38-parameter signature and report file naming."""
    fx = gu.load("hla_small_pair")
    loc = fx["_locus"]
    d = loc.reference_dicts()
    bam = tmp_path / "syn.sam"
    bam.write_text(fx["sam"])
    passed = hgx.typing(True, str(tmp_path / "hla"), [fx["options"]["sample"]], "", True, set(), d["refGenes"],
                        d["Genes"], d["Gene_names"], d["Gene_lengths"], d["refGene_loci"], d["Vars"], d["Var_list"],
                        d["Links"], [["hisat2", "graph"]], 2, False, "assembly_graph", True, True, False, False, True,
                        [], False, ["r1.fa", "r2.fa"], str(bam), [], 100, 350, 1, False, 0, False, str(tmp_path),
                        "NONE", True, 0)
    rep = (tmp_path / "assembly_graph-hla.test-1.report").read_text()
    keep = lambda ls: [l for l in ls if "aligned" in l or "ranked" in l or "(count:" in l]
    assert keep(rep.split("\n")) == keep(fx["report"].split("\n"))
    assert passed == {"hisat2 graph": 2}


def _compare_with_pyref(loc, sam, **opts):
    rl = pyref.RefLocus(loc, **opts)
    exp = rl.run(sam)
    pl = hl.PackedLocus.from_synth(loc)
    res = hgx.type_locus(pl, sam, simulation=opts.get("simulation", False))
    assert (res.num_reads, res.num_pairs) == (exp["num_reads"], exp["num_pairs"])
    assert res.counts_sorted == exp["counts_sorted"]            # integer compatibility counts: bit-exact, same tie order
    assert len(res.em) == len(exp["em"])
    for got, e in zip(res.em, exp["em"]):
        _check_em(got, e["result"], e["n_iter"])
    assert [a for a, _ in res.gene_prob] == [a for a, _ in exp["gene_prob"]]
    for (a, p), (b, q) in zip(res.gene_prob, exp["gene_prob"]):
        assert abs(p - q) <= 1e-9
    return res


def test_class1_panel_three_loci():
    """BASELINE configs[2] shape at test size: three class-I-like loci of different sizes, each typed independently."""
    specs = [("A", 600, 3569, 1300, 0), ("B", 800, 4081, 1500, 5000), ("C", 500, 4305, 1200, 10000)]
    for k, (gene, n_all, length, n_vars, base) in enumerate(specs):
        loc = synth.make_hla_like_locus(gene=gene, n_alleles=n_all, length=length, n_vars=n_vars, seed=300 + k, var_id_base=base)
        sample = synth.pick_sample(loc, 40 + k)
        al = synth.simulate_pairs(loc, sample, 500, err_rate=0.003, seed=50 + k)
        res = _compare_with_pyref(loc, synth.sam_text(loc, al))
        assert {a for a, _ in res.gene_prob[:2]} == set(sample)      # identical (and correct) top-2 calls


def test_codis_panel_bit_exact():
    """BASELINE configs[4] shape at test size: several STR loci (all-deletion alleles, many alternative alignments),
    including D18S51 with its choose_pairs special case."""
    panel = [("D8S1179", "TCTA", 19, 7), ("D18S51", "AGAA", 22, 9), ("TH01", "AATG", 12, 5), ("FGA", "CTTT", 30, 16)]
    for k, (gene, unit, mx, mn) in enumerate(panel):
        loc = synth.make_str_like_locus(gene=gene, unit=unit, max_repeats=mx, min_repeats=mn, flank=170, seed=60 + k,
                                        var_id_base=100 * k)
        sample = ["%s*%d" % (gene, mn + 2), "%s*%d" % (gene, mx - 3)]
        al = synth.simulate_pairs(loc, sample, 0, read_len=100, frag_len=(250, 250), seed=70 + k, simulation_names=True,
                                  tile_interval=4)
        _compare_with_pyref(loc, synth.sam_text(loc, al), simulation=True)


def test_half_million_reads_properties():
    """Size-independent properties at (half of) the bench size, where no oracle finishes in seconds."""
    import numpy as np
    from hisatgenotype_amd import engine
    loc = synth.make_hla_like_locus(n_alleles=7000, n_vars=2500, seed=101)
    sample = synth.pick_sample(loc, 101)
    sam = synth.simulate_sam_fast(loc, sample, 250000, err_rate=0.002, seed=100)
    pl = hl.PackedLocus.from_synth(loc)
    batch = pl.parse_sam(sam)
    assert batch.n_pairs >= 249000 and batch.n_refs > batch.n_pieces
    db = engine.DeviceBatch(batch)
    bufs = engine.ScoreBuffers(pl, db)
    engine.score_pairs(pl, db, bufs)
    gcl = engine.Classes.dedup(bufs.gene_bits, db.n_pairs, pl.a_pad, hashes=bufs.gene_hash)
    bits, cnt, first = gcl.to_host()
    assert cnt.sum() == batch.n_pairs                      # every pair lands in exactly one class
    assert np.all(np.diff(first) > 0)                      # classes come out in first-seen order
    assert len({r.tobytes() for r in bits}) == len(bits)   # and are distinct
    # dedup is idempotent: the class matrix deduplicated again (weights = counts) is itself
    d_w = engine.DevArray.from_host(cnt)
    b_ptr, _, _ = gcl.device_ptrs()
    again = engine.Classes.dedup(engine._RawDev(b_ptr), gcl.n_classes, pl.a_pad, weights=d_w)
    b2, c2, _ = again.to_host()
    assert np.array_equal(b2, bits) and np.array_equal(c2, cnt)
    # Gene_counts = column sums of the weighted class matrix
    ac, _ = gcl.allele_counts()
    col = np.zeros(pl.a_pad, np.int64)
    for w in range(pl.w64):
        wb = bits[:, w]
        for b in range(64):
            col[64 * w + b] = int(cnt[(wb >> np.uint64(b)) & np.uint64(1) == 1].sum())
    assert np.array_equal(ac, col) and ac.max() <= batch.n_pairs
    # the whole path: abundances are a distribution and the two true alleles win
    res = hgx.type_locus(pl, sam)
    assert abs(sum(p for _, p in res.gene_prob) - 1.0) < 1e-9
    assert {a for a, _ in res.gene_prob[:2]} == set(sample)
    # re-scoring gives identical rows and hashes (deterministic kernels)
    rows1, h1 = bufs.gene_bits.to_host(), bufs.gene_hash.to_host()
    engine.score_pairs(pl, db, bufs)
    assert np.array_equal(rows1, bufs.gene_bits.to_host()) and np.array_equal(h1, bufs.gene_hash.to_host())


def test_genotyping_locus_from_index_files(tmp_path):
    """8f-2: the reference's 32-parameter genotyping_locus on a stand-alone index directory + SAM file."""
    fx = gu.load("hla_mid_real")
    loc = fx["_locus"]
    synth.write_index([loc], str(tmp_path / "ix"), "hla")
    sam = tmp_path / "sample1.sam"
    sam.write_text(fx["sam"])
    hgx.genotyping_locus("hla", ["A"], "", str(tmp_path / "ix"), [], True, [["hisat2", "graph"]], ["sample1.fq"], True,
                         str(sam), 1, 10, 150, 400, False, 2, 0.0, 0.0, [], False, "assembly_graph", True, True, False,
                         False, True, [], 0, False, str(tmp_path), True, {})
    rep = (tmp_path / "assembly_graph-hla.sample1.report").read_text()
    keep = lambda ls: [l for l in ls if "aligned" in l or "ranked" in l or "(count:" in l]
    assert keep(rep.split("\n")) == keep(fx["report"].split("\n"))


def test_genotyping_locus_genotype_genome_mode(tmp_path):
    """8f-2, genotype-genome indexes (typing_core.py:2326-2397, 438-441): two loci embedded in one chromosome, the sample's
    reads of BOTH loci in one coordinate-sorted BAM on chromosome coordinates.  Each locus must see exactly its own reads
    (samtools' overlap rule on `chr:left-right`) re-based to the locus, and report what the stand-alone run reports.
    As in the reference, typing() then sees base_fname = the genome's name ("genotype_genome"), not "hla": the report is
    named after it and the HLA-only exon-level stage is off (typing_core.py:291, 1732)."""
    from hisatgenotype_amd import bamio
    a = synth.make_hla_like_locus(gene="A", n_alleles=120, n_vars=320, seed=81, sibling_frac=0.3)
    b = synth.make_hla_like_locus(gene="B", n_alleles=90, n_vars=260, length=2600, seed=82, var_id_base=7000)
    ix_dir = str(tmp_path / "ix")
    spans = synth.write_genome_index([a, b], ix_dir, "genotype_genome", "hla", chrom="6", gap=300, seed=4)
    lines, expect = [], {}
    for k, loc in enumerate((a, b)):
        sample = synth.pick_sample(loc, 60 + k)
        al = synth.simulate_pairs(loc, sample, 400, err_rate=0.002, seed=90 + k)
        left = spans[loc.gene][0]
        for l in synth.sam_text(loc, al, base_locus=left).split("\n"):
            if l:
                f = l.split("\t")
                f[0] = "%s_%s" % (loc.gene, f[0])
                f[2] = "6"
                lines.append("\t".join(f))
        loc.base_fname = "genotype_genome"
        expect[loc.gene] = pyref.RefLocus(loc).run(synth.sam_text(loc, al))
    lines.sort(key=lambda l: int(l.split("\t")[3]))
    bam = tmp_path / "wgs.bam"
    bamio.write_bam(str(bam), "\n".join(lines) + "\n", [("6", spans["B"][1] + 301)])
    hgx.genotyping_locus("hla", ["A", "B"], "genotype_genome", ix_dir, [], True, [["hisat2", "graph"]], ["wgs.fq"], True,
                         str(bam), 1, 10, 150, 400, False, 2, 0.0, 0.0, [], False, "assembly_graph", True, True, False,
                         False, True, [], 0, False, str(tmp_path), True, {})
    rep = (tmp_path / "assembly_graph-genotype_genome.wgs.report").read_text().split("\n")
    got_counts = [l.strip() for l in rep if "(count:" in l]
    want_counts = ["%d %s (count: %d)" % (i + 1, n, c) for g in ("A", "B") for i, (n, c) in enumerate(expect[g]["counts_sorted"])]
    assert got_counts == want_counts
    aligned = [l.strip() for l in rep if "aligned" in l]
    assert aligned == ["%d reads and %d pairs are aligned" % (expect[g]["num_reads"], expect[g]["num_pairs"]) for g in ("A", "B")]
    got_ab = [l.strip() for l in rep if "abundance" in l]
    want_ab = []
    for g in ("A", "B"):
        for i, (n, p) in enumerate(expect[g]["gene_prob"]):
            if p < 0.01 or i >= 10:
                break
            want_ab.append("%d ranked %s (abundance: %.2f%%)" % (i + 1, n, p * 100.0))
    assert got_ab == want_ab


def test_run_panel_shards_tasks(tmp_path):
    """Config-4 shape at test size: samples x loci as independent tasks, split over two 'ranks' without communication."""
    from hisatgenotype_amd import indexio
    loci = [synth.make_hla_like_locus(gene=g, n_alleles=120, n_vars=300, seed=70 + i, var_id_base=1000 * i)
            for i, g in enumerate(("A", "B", "DRB1"))]
    synth.write_index(loci, str(tmp_path), "hla")
    ix = indexio.load_index(str(tmp_path), "hla")
    tasks, truth = [], {}
    for s in range(4):
        for loc in loci:
            sample = synth.pick_sample(loc, 10 * s + 1)
            sam = synth.simulate_sam_fast(loc, sample, 300, err_rate=0.002, seed=s)
            tasks.append((s, loc.gene, sam))
            truth[(s, loc.gene)] = set(sample)
    got = {}
    for rank in range(2):
        part = hgx.run_panel(tasks, ix, "hla", rank=rank, world=2)
        assert not set(part) & set(got)
        got.update(part)
    assert set(got) == set(truth)
    for key, res in got.items():
        assert {a for a, _ in res.gene_prob[:2]} == truth[key]
    # the same panel with three tasks in flight on the GPU: identical results
    conc = hgx.run_panel(tasks, ix, "hla", inflight=3)
    assert set(conc) == set(got)
    for key in got:
        assert conc[key].gene_prob == got[key].gene_prob and conc[key].counts_sorted == got[key].counts_sorted


def test_typing_reads_bam_without_samtools(tmp_path):
    """8f-3: the alignment may be a BAM; it is decoded in-process and name-grouped like `samtools view | sort -k1,1 -s`."""
    from hisatgenotype_amd import bamio
    fx = gu.load("hla_mid_real")
    loc = fx["_locus"]
    d = loc.reference_dicts()
    lines = [l for l in fx["sam"].split("\n") if l]
    lines.sort(key=lambda l: int(l.split("\t")[3]))                     # coordinate-sorted, as a real BAM would be
    bam = tmp_path / "sample.bam"
    bamio.write_bam(str(bam), "\n".join(lines) + "\n", [(loc.ref_allele, len(loc.backbone))])
    hgx.typing(False, str(tmp_path / "hla"), ["A"], "", True, set(), d["refGenes"], d["Genes"], d["Gene_names"], d["Gene_lengths"],
               d["refGene_loci"], d["Vars"], d["Var_list"], d["Links"], [["hisat2", "graph"]], 2, False, "assembly_graph", True, True,
               False, False, True, [], False, ["sample.fq"], str(bam), [], 150, 400, 1, False, 0, False, str(tmp_path), "NONE", True)
    rep = (tmp_path / "assembly_graph-hla.sample.report").read_text()
    keep = lambda ls: [l for l in ls if "aligned" in l or "ranked" in l or "(count:" in l]
    assert keep(rep.split("\n")) == keep(fx["report"].split("\n"))


def test_concurrent_samples_give_identical_results():
    """Two host threads typing two samples at the same time on one GPU (own streams, own class-row buffers, shared locus
    index and pool) get exactly what they get one after the other."""
    import threading
    from hisatgenotype_amd import synth, locus as hl, engine, capi
    ht = sys.modules["hisatgenotype_amd.typing"]        # the module (the package also exports the typing() function)
    loc = synth.make_hla_like_locus(n_alleles=900, n_vars=700, seed=12)
    pl = hl.PackedLocus.from_synth(loc)
    pl.index()
    batches = []
    for seed in (3, 4):
        sample = synth.pick_sample(loc, seed)
        sam = synth.simulate_sam_fast(loc, sample, 6000, err_rate=0.002, seed=seed)
        batches.append(pl.parse_sam(sam))

    def run(batch, out, own_stream):
        capi.set_device(capi.current_device())
        res = ht.LocusResult()
        res.num_reads, res.num_pairs = batch.n_reads, batch.n_pairs
        st = capi.get_stream(2) if own_stream else None
        r = ht._type_batch(pl, batch, res, True, stream=st, overlap=True)
        out.append((r.gene_prob, [e["n_iter"] for e in r.em], r.counts_sorted[:10]))

    seq = []
    for b in batches:
        run(b, seq, False)
    for _ in range(3):
        outs = [[], []]
        ths = [threading.Thread(target=run, args=(b, o, True)) for b, o in zip(batches, outs)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        assert outs[0][0] == seq[0] and outs[1][0] == seq[1]


def test_five_samples_in_flight_beyond_one_run_of_streams():
    """The library makes its streams in runs of 24 (three stream sets + three main streams for callers, placed on the hardware queues
    on purpose: hgx_type.hip make_streams).  Five host threads with a sample each need two runs of sets and more main streams than
    one run holds (the fourth and fifth come from the second run): every sample's result is the one it has alone, round after round."""
    import threading
    from hisatgenotype_amd import capi
    ht = sys.modules["hisatgenotype_amd.typing"]
    loc = synth.make_hla_like_locus(n_alleles=900, n_vars=700, seed=77)
    pl = hl.PackedLocus.from_synth(loc)
    pl.index()
    batches = []
    for seed in range(5):
        sam = synth.simulate_sam_fast(loc, synth.pick_sample(loc, 20 + seed), 5000, err_rate=0.002, seed=40 + seed)
        batches.append(pl.parse_sam(sam))

    def run(batch, out, slot):
        capi.set_device(capi.current_device())
        if slot is not None:
            capi.set_stream_slot(("five in flight", slot))
        res = ht.LocusResult()
        res.num_reads, res.num_pairs = batch.n_reads, batch.n_pairs
        r = ht._type_batch(pl, batch, res, True, stream=capi.get_stream(2) if slot is not None else None, overlap=True)
        out.append((r.gene_prob, [e["n_iter"] for e in r.em], r.counts_sorted[:10]))

    alone = []
    for b in batches:
        run(b, alone, None)
    mains = set()
    for _ in range(3):
        outs = [[] for _ in batches]
        ths = [threading.Thread(target=run, args=(b, o, k)) for k, (b, o) in enumerate(zip(batches, outs))]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        assert [o[0] for o in outs] == alone
    for k in range(5):
        capi.set_stream_slot(("five in flight", k))
        mains.add(capi.get_stream(2).value)
    assert len(mains) == 5                                   # a main stream per slot, none handed out twice


@pytest.mark.parametrize("name", ["hla_small_pair", "hla_mid_real", "hla_errors_filters", "hla_7000"])
def test_grouped_exon_path_equals_per_pair_path(name, monkeypatch):
    """type_locus through hgx_level_classes (pairs grouped by exon-level ref list, the default) and through the per-pair rows +
    dedup (hgx_type_opts.per_pair_exon) give the same classes, counts, EM results and report."""
    fx = gu.load(name)
    o = fx["options"]
    pl = hl.PackedLocus.from_synth(fx["_locus"])

    def run(per_pair):
        return hgx.type_locus(pl, fx["sam"], num_editdist=o["num_editdist"], error_correction=o["error_correction"],
                              allow_discordant=o["allow_discordant"], remove_low_abundance_alleles=o["remove_low"],
                              simulation=o["simulation"], keep_classes=True, per_pair_exon=per_pair)
    a = run(False)
    b = run(True)
    assert a.em == b.em and a.gene_prob == b.gene_prob
    for x, y in zip(a.exon_classes, b.exon_classes):
        assert np.array_equal(x, y)
    for x, y in zip(a.gene_classes, b.gene_classes):
        assert np.array_equal(x, y)
    assert hgx.report_lines(a, o["simulation"], o["sample"] if o["simulation"] else (), True) == \
        hgx.report_lines(b, o["simulation"], o["sample"] if o["simulation"] else (), True)


def test_grouped_exon_path_equals_per_pair_path_at_size(monkeypatch):
    """The same at 200 k pairs (overlapped streams, > 65 536 groups): identical EM results, bit for bit."""
    loc = synth.make_hla_like_locus(n_alleles=7000, n_vars=2500, seed=101)
    sample = synth.pick_sample(loc, 103)
    sam = synth.simulate_sam_fast(loc, sample, 200000, err_rate=0.002, seed=7)
    pl = hl.PackedLocus.from_synth(loc)
    a = hgx.type_locus(pl, sam)
    b = hgx.type_locus(pl, sam, per_pair_exon=True)
    assert a.em == b.em and a.gene_prob == b.gene_prob
    assert np.array_equal(a.counts, b.counts) and np.array_equal(a.counts_order, b.counts_order)


@pytest.mark.parametrize("seed0,k,scale", [(1000, 492, 1), (20000, 40, 2)])
def test_exact_ratio_pruning_ties_follow_the_reference(seed0, k, scale):
    """Two fuzz cases (tools/fuzz_parity.py) whose hand-off EM prunes on an EXACT 10 : 1 ratio (1/11 vs 10/11, 1/18 vs 5/9): the
    reference decides on its own rounding noise.  The hand-off EM sums in the reference's order, so it decides the same way."""
    import random
    rng = random.Random(seed0 + k)
    assert rng.random() >= 0.25                               # (both are HLA-like cases of the generator)
    loc = synth.make_hla_like_locus(n_alleles=rng.randint(30, 1200), n_vars=rng.randint(60, 900), seed=seed0 + k,
                                    insertion_frac=rng.choice([0.0, 0.03]), unlinked_vars=rng.randint(0, 4))
    sample = synth.pick_sample(loc, seed0 + k)
    al = synth.simulate_pairs(loc, sample, scale * rng.randint(60, 220), err_rate=rng.choice([0.0, 0.003, 0.01]), seed=k,
                              softclip_frac=rng.choice([0.0, 0.05]), novel_del_frac=rng.choice([0.0, 0.03]),
                              multi_hit_frac=rng.choice([0.0, 0.02]), dup_frac=rng.choice([0.0, 0.02]),
                              novel_ins_frac=rng.choice([0.0, 0.02]), single_end=rng.random() < 0.15)
    sam = synth.sam_text(loc, al)
    single = any(a.flag & 1 == 0 for a in al)
    pl = hl.PackedLocus.from_synth(loc)
    exp = pyref.RefLocus(loc, allow_discordant=single).run(sam)
    res = hgx.type_locus(pl, sam, allow_discordant=single)
    assert [g["n_iter"] for g in res.em] == [e["n_iter"] for e in exp["em"]]
    assert [a for a, _ in res.gene_prob] == [a for a, _ in exp["gene_prob"]]
    for (a, p), (_, q) in zip(res.gene_prob, exp["gene_prob"]):
        assert abs(p - q) <= 1e-9
    for (a, p), (_, q) in zip(res.em[1]["result"], exp["em"][1]["result"]):
        assert p == q                                         # bit-identical hand-off EM


@pytest.mark.parametrize("seed0,k,scale", [(200000, 2754, 1), (230000, 3313, 4), (200000, 1138, 1), (200000, 9927, 1), (101000, 53, 5),
                                           (102000, 38, 20)])
def test_long_em_fuzz_cases_are_bit_identical(seed0, k, scale):
    """Six cases a 24 600-case fuzz run and an earlier one turned up (tools/fuzz_parity.py make_case): LONG EMs on small problems, where
    rounding-level differences grow until an iteration count (15 vs 14, 59 vs 54), an abundance (1.4e-6) or the order inside a
    near-tie differed.  The exon-level EM of such problems (k_em_ref) and a hand-off with MORE than 64 alleles (the filtered class set
    inherits the name order and takes the same kernel) run in the reference's own order now: everything `==`."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                              "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    loc, sam, single = fz.make_case(seed0, k, scale)
    pl = hl.PackedLocus.from_synth(loc)
    exp = pyref.RefLocus(loc, allow_discordant=single).run(sam)
    res = hgx.type_locus(pl, sam, allow_discordant=single)
    assert [g["n_iter"] for g in res.em] == [e["n_iter"] for e in exp["em"]]
    assert res.gene_prob == [[a, p] for a, p in exp["gene_prob"]]
    for got, e in zip(res.em, exp["em"]):
        assert got["result"] == [[a, p] for a, p in e["result"]]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_locus_equals_unsharded(world):
    """8e, intra-locus read sharding: one sample's pairs of one locus split over `world` ranks (here: threads of one process
    on one GPU, the exchanges through dist.LocalComm) -- pileup all-reduce inside the front-end, class tables gathered and
    merged in rank order, EMs on the merged tables -- give exactly the unsharded result on every rank: counts, their order,
    every EM (classes, iterations, alleles, abundances) and the final abundances."""
    import threading
    from hisatgenotype_amd import capi, dist as hdist
    for loc, kw in ((synth.make_hla_like_locus(n_alleles=900, n_vars=700, seed=12, sibling_frac=0.4), dict(err_rate=0.004)),
                    (synth.make_str_like_locus(gene="TH01", unit="AATG", max_repeats=12, min_repeats=4),
                     dict(read_len=100, frag_len=(230, 270), err_rate=0.002))):
        pl = hl.PackedLocus.from_synth(loc)
        pl.index()
        sample = synth.pick_sample(loc, 3)
        sam = synth.simulate_sam_fast(loc, sample, 9000, seed=4, **kw)
        ref = hgx.type_locus(pl, sam)
        shards = hdist.split_name_grouped(sam, world)
        assert b"".join(shards) == sam.encode() and all(shards)
        # "device": every rank's shard through the DEVICE front end (k_fe_*; the shards are below its size gate), the pileup summed
        # over the ranks where k_fe_pileup left it, in HBM (comm.allreduce_u32_dev) -- no host parse on the sharded path (round 5);
        # "host": the host front end with the host form of the same exchange; "mixed": rank 0 on the host route, the others on the
        # device route -- the two forms are ONE collective
        for mode in ("device", "host", "mixed"):
            comms = hdist.LocalComm.make(world)
            out, errs, routes = [None] * world, [], [None] * world

            def run(r):
                try:
                    capi.set_device(capi.current_device())
                    front = "host" if (mode == "host" or (mode == "mixed" and r == 0)) else None
                    out[r] = hdist.type_locus_sharded(pl, shards[r], comms[r], stream=capi.get_stream(2), front=front)
                    routes[r] = engine.front_last()
                except BaseException as e:
                    errs.append(e)
                    comms[r].sh.barrier.abort()
            with engine.test_switches(front="device"):
                ths = [threading.Thread(target=run, args=(r,)) for r in range(world)]
                for t in ths:
                    t.start()
                for t in ths:
                    t.join()
            assert not errs, errs
            for r in range(world):
                if mode == "device" or (mode == "mixed" and r > 0):
                    assert routes[r] == (2, 0), (mode, r, routes)
            for res in out:
                assert (res.num_reads, res.num_pairs) == (ref.num_reads, ref.num_pairs)
                assert res.counts_sorted == ref.counts_sorted
                assert res.em == ref.em and res.gene_prob == ref.gene_prob


_BROADCAST_WORKER = r"""
import os, sys
sys.path.insert(0, %r)
import ctypes as C
import numpy as np
import torch
import torch.distributed as dist
torch.cuda.set_device(0)                       # torch's HIP runtime first: it cannot come up after libhgx's in one process
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29611")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import hisatgenotype_amd as hgx
from hisatgenotype_amd import capi, synth, locus as hl, dist as hdist
capi.set_device(0)
loc = synth.make_hla_like_locus(n_alleles=700, n_vars=600, seed=31)
pl = hl.PackedLocus.from_synth(loc)
sample = synth.pick_sample(loc, 2)
sam = synth.simulate_sam_fast(loc, sample, 3000, err_rate=0.002, seed=6)
ref = hgx.type_locus(pl, sam)
nbytes = hdist.broadcast_index(pl, src=0)
assert nbytes == (pl.n_words * pl.a_pad + 4 * pl.w64) * 4
block = hdist.index_block_tensor(pl.index())
t = pl.tables()
host = block.cpu().numpy()
nb = pl.n_words * pl.a_pad
assert np.array_equal(host[:nb].view(np.uint32).reshape(pl.n_words, pl.a_pad), t["link_bits"])
assert np.array_equal(host[nb:nb + 2 * pl.w64].view(np.uint64), t["exon_mask"])
assert np.array_equal(host[nb + 2 * pl.w64:].view(np.uint64), t["gene_mask"])
# the receiving side: an index with uninitialised tables, filled device-to-device through its aliasing tensor
h = C.c_void_p()
capi.check(capi.lib().hgx_index_create_device(C.byref(h), C.c_int32(pl.n_alleles), C.c_int32(pl.n_vars)))
hdist.index_block_tensor(h).copy_(block)
torch.cuda.synchronize()
old, pl._index = pl._index, h
got = hgx.type_locus(pl, sam)
pl._index = old
capi.lib().hgx_index_destroy(h)
assert got.gene_prob == ref.gene_prob and got.counts_sorted == ref.counts_sorted and got.em == ref.em
dist.destroy_process_group()
print("broadcast ok", nbytes)
"""


def test_index_broadcast_writes_into_index_memory(tmp_path):
    """8e: dist.broadcast_index over RCCL (backend nccl, here a world of one rank, in a fresh process: torch's HIP runtime
    has to come up before libhgx's) sends / receives the device block of the index itself -- a torch tensor aliasing
    [link bits | exon mask | gene mask] (hgx_index_device_block) -- and an index made by hgx_index_create_device + a
    device-to-device copy of that block types a sample exactly like the original."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "bcast.py"
    script.write_text(_BROADCAST_WORKER % root)
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "broadcast ok" in out.stdout


@pytest.mark.parametrize("front", FRONTS[:2], ids=FRONT_IDS[:2])
@pytest.mark.parametrize("name", gu.ALL)
def test_type_file_sam_bam_sorted_bam_with_regions(name, front, tmp_path):
    """hgx_type_file -- the entry bench.py's file -> result leg times -- on every fixture, from (a) the SAM text as the aligner
    writes it, (b) a BAM of the same records, (c) a coordinate-sorted BAM that also holds reads of ANOTHER reference sequence,
    with the locus' backbone as the region (what the reference's `samtools view F ref_allele | sort -k1,1 -s` sees): the
    reference's report lines in all three cases."""
    from hisatgenotype_amd import bamio
    fx = gu.load(name)
    if fx.get("error"):
        pytest.skip("the reference raises on this fixture")
    o = fx["options"]
    loc = fx["_locus"]
    pl = hl.PackedLocus.from_synth(loc)
    lines = [l for l in fx["sam"].split("\n") if l]
    sam = tmp_path / "reads.sam"
    sam.write_text("@HD\tVN:1.0\tSO:unsorted\n" + "\n".join(lines) + "\n")
    bam = tmp_path / "reads.bam"
    refs = [(loc.ref_allele, len(loc.backbone)), ("other*BACKBONE", 5000)]
    bamio.write_bam_native(str(bam), "\n".join(lines) + "\n", refs)
    # decoys on the other sequence (same read names as real reads, so a reader that ignored the region would pair them up)
    decoys = []
    for l in lines[:40]:
        c = l.split("\t")
        c[2] = "other*BACKBONE"
        decoys.append("\t".join(c))
    sbam = tmp_path / "sorted.bam"
    bamio.write_bam_native(str(sbam), "\n".join(lines + decoys) + "\n", refs, sort_by_coordinate=True)
    kw = dict(num_editdist=o["num_editdist"], error_correction=o["error_correction"], allow_discordant=o["allow_discordant"],
              remove_low_abundance_alleles=o["remove_low"], simulation=o["simulation"])
    keep = lambda ls: [l for l in ls if "aligned" in l or "ranked" in l or "(count:" in l]
    want = keep(fx["report"].split("\n"))
    ref = hgx.type_locus(pl, fx["sam"], **kw)
    for path in (sam, bam, sbam):
        # "device": the file through the record route -- for the BAMs also the device's BGZF inflate, record walk, region filter and
        # name sort (k_bgzf_inflate, k_bam_*) -- and the REFERENCE's report text out of it
        with _front(front, name) as f:
            res = hgx.type_file(pl, str(path), **kw)
            f.check()
        assert res.num_reads == ref.num_reads and res.num_pairs == ref.num_pairs, path.name
        assert res.gene_prob == ref.gene_prob and res.em == ref.em and res.counts_sorted == ref.counts_sorted, path.name
        got, _ = hgx.report_lines(res, o["simulation"], o["sample"] if o["simulation"] else (), True)
        assert keep(got) == want, path.name


def _type_sharded(pl, sam, world, fronts=None):
    """One sample's pairs of a locus split over `world` threads-as-ranks on one GPU (dist.LocalComm); returns rank 0's result.
    `fronts[r]` = "host": rank r takes the host front end whatever the switches say."""
    import threading
    from hisatgenotype_amd import capi, dist as hdist
    pl.index()
    shards = hdist.split_name_grouped(sam, world)
    comms = hdist.LocalComm.make(world)
    out, errs, routes = [None] * world, [], [None] * world

    def run(r):
        try:
            capi.set_device(capi.current_device())
            out[r] = hdist.type_locus_sharded(pl, shards[r], comms[r], stream=capi.get_stream(2), front=fronts[r] if fronts else None)
            routes[r] = engine.front_last()
        except BaseException as e:
            errs.append(e)
            comms[r].sh.barrier.abort()
    ths = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    for r in out[1:]:
        assert r.gene_prob == out[0].gene_prob and r.counts_sorted == out[0].counts_sorted
    out[0].front_routes = routes                   # (route, decline code) of every rank's parse
    return out[0]


def test_class1_at_stated_size_properties_and_sharding():
    """BASELINE configs[2] at its stated size -- HLA class I A / B / C with 7 000 / 8 000 / 7 000 alleles, 500 k pairs (1 M reads)
    each: size-independent properties per locus (both true alleles on top, pairs conserved, abundances sum to one, a second run
    identical) and, for the largest locus, the sharded form (two threads as ranks, pileup exchange + class-table merge) equal to
    the unsharded one."""
    from hisatgenotype_amd import dist as hdist
    cfg = [("A", 7000, 3569, 2500, 101), ("B", 8000, 4081, 2800, 102), ("C", 7000, 4305, 2600, 103)]
    n_pairs = 500000
    for i, (g, a, ln, v, sd) in enumerate(cfg):
        loc = synth.make_hla_like_locus(gene=g, n_alleles=a, length=ln, n_vars=v, seed=sd, var_id_base=100000 * i)
        pl = hl.PackedLocus.from_synth(loc)
        sample = synth.pick_sample(loc, 101 + i)
        sam = synth.simulate_sam_fast(loc, sample, n_pairs, err_rate=0.002, seed=100 + i)
        res = hgx.type_locus(pl, sam)
        assert 0.99 * n_pairs <= res.num_pairs <= n_pairs and res.num_reads > 1.9 * n_pairs
        assert sorted(x for x, _ in res.gene_prob[:2]) == sorted(sample)
        assert abs(sum(p for _, p in res.gene_prob) - 1.0) < 1e-9
        assert res.counts_sorted[0][1] <= res.num_pairs
        again = hgx.type_locus(pl, sam)
        assert again.gene_prob == res.gene_prob and again.em == res.em
        if g == "B":
            sharded = _type_sharded(pl, sam, 2)
            assert sharded.front_routes == [(2, 0), (2, 0)]      # 250 k pairs per rank: the device front end by itself, exchange in HBM
            assert sharded.num_reads == res.num_reads and sharded.num_pairs == res.num_pairs
            assert sharded.counts_sorted == res.counts_sorted
            assert [a for a, _ in sharded.gene_prob] == [a for a, _ in res.gene_prob]
            assert max(abs(p - q) for (_, p), (_, q) in zip(sharded.gene_prob, res.gene_prob)) <= 1e-9
        del sam


def test_sharded_d18s51_uses_the_whole_samples_pair_distance():
    """CODIS D18S51 sharded over threads-as-ranks: choose_pairs (typing_core.py:680-716) needs the MEDIAN inner distance of the
    whole sample (get_pair_interdist, typing_common.py:1187-1265) -- the ranks all-reduce a histogram of their distances
    (hgx_parse_opts.interdist_exchange) -- and the result equals the unsharded one (round 2 raised here)."""
    loc = synth.make_str_like_locus(gene="D18S51", unit="AGAA", max_repeats=22, min_repeats=9, flank=180, seed=71)
    loc.base_fname = "codis"
    pl = hl.PackedLocus.from_synth(loc)
    names = [a for a in loc.allele_names if "BACKBONE" not in a]
    sample = [names[3], names[-3]]
    sam = synth.simulate_sam_fast(loc, sample, 4000, read_len=100, frag_len=(200, 280), err_rate=0.002, seed=17)
    ref = hgx.type_locus(pl, sam)
    with engine.test_switches(front="host"):
        assert hgx.type_locus(pl, sam).gene_prob == ref.gene_prob          # (the unsharded sample: kernels == host stages)
    for world in (2, 3):
        got = _type_sharded(pl, sam, world)
        assert (got.num_reads, got.num_pairs) == (ref.num_reads, ref.num_pairs)
        assert got.counts_sorted == ref.counts_sorted and got.em == ref.em and got.gene_prob == ref.gene_prob
        # every shard through the kernels (k_fe_interdist_* count the distances, the histogram is exchanged after the pileup), and mixed
        # with a rank on the host stages: the same two exchanges in the same order
        with engine.test_switches(front="device"):
            for fronts in (None, ["host"] + [None] * (world - 1)):
                got = _type_sharded(pl, sam, world, fronts=fronts)
                assert got.counts_sorted == ref.counts_sorted and got.em == ref.em and got.gene_prob == ref.gene_prob
                assert got.front_routes[1:] == [(2, 0)] * (world - 1) and got.front_routes[0][0] == (0 if fronts else 2), got.front_routes


def test_a_failing_rank_fails_every_rank_of_a_sharded_locus():
    """A rank whose front-end raises (here: a record without NM, quirk Q8) must not leave its peers blocked in the pileup
    all-reduce: the failure travels with the exchange and every rank raises."""
    import threading
    from hisatgenotype_amd import capi, dist as hdist
    loc = synth.make_hla_like_locus(n_alleles=300, n_vars=400, seed=8)
    pl = hl.PackedLocus.from_synth(loc)
    pl.index()
    sam = synth.simulate_sam_fast(loc, synth.pick_sample(loc, 1), 600, seed=2)
    shards = hdist.split_name_grouped(sam, 2)
    shards[1] = shards[1].replace(b"NM:i:", b"XM:i:", 1)
    # by default (these shards are small: the host stages), with every rank forced onto the device route (the kernels decline the
    # record, the host stages word the error: the rank has exchanged on the device by then and must not exchange again), and mixed
    for switches, fronts in ((dict(), (None, None)), (dict(front="device"), (None, None)), (dict(front="device"), ("host", None)),
                             (dict(front="device"), (None, "host"))):
        comms = hdist.LocalComm.make(2)
        errs = [None, None]

        def run(r):
            try:
                capi.set_device(capi.current_device())
                hdist.type_locus_sharded(pl, shards[r], comms[r], stream=capi.get_stream(2), front=fronts[r])
            except BaseException as e:
                errs[r] = e
        with engine.test_switches(**switches):
            ths = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(2)]      # (daemon: a blocked rank must not hang the run)
            for t in ths:
                t.start()
            for t in ths:
                t.join(timeout=60)
        if any(t.is_alive() for t in ths):
            comms[0].sh.barrier.abort()
            pytest.fail("a rank is still blocked in an exchange")
        assert errs[0] is not None and errs[1] is not None, (switches, fronts)


_RCCL_ONE_WORKER = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np
import hisatgenotype_amd as hgx
from hisatgenotype_amd import capi, synth, locus as hl, dist as hdist
comm = hdist.RcclComm(0, 1, lambda x: x)           # (the communicator first: RCCL loads its kernels at this point)
loc = synth.make_hla_like_locus(n_alleles=700, n_vars=600, seed=31)
pl = hl.PackedLocus.from_synth(loc)
sam = synth.simulate_sam_fast(loc, synth.pick_sample(loc, 2), 3000, err_rate=0.002, seed=6)
ref = hgx.type_locus(pl, sam)
before = {k: v.copy() for k, v in pl.tables().items()}
comm.broadcast_index(pl, 0)
again = hgx.type_locus(pl, sam)
assert again.gene_prob == ref.gene_prob and all(np.array_equal(before[k], pl.tables()[k]) for k in before)
a = np.arange(1000, dtype=np.int64) * 3
assert np.array_equal(comm.allreduce_sum(a.copy()), a)
from hisatgenotype_amd import engine
for sw in (dict(), dict(front="device")):       # forced: the shard through k_fe_*, the pileup counters all-reduced in HBM by hgx_allreduce_sum_u32
    with engine.test_switches(**sw):
        got = hdist.type_locus_sharded(pl, sam.encode(), comm)
        assert not sw or engine.front_last() == (2, 0), engine.front_last()
    assert (got.num_reads, got.num_pairs) == (ref.num_reads, ref.num_pairs)
    assert got.counts_sorted == ref.counts_sorted and got.em == ref.em and got.gene_prob == ref.gene_prob
u = comm.allreduce_u32(np.array([7, 0xFFFFFFF0], np.uint32))
assert u.tolist() == [7, 0xFFFFFFF0]
comm.close()
print("rccl world-1 ok")
"""


def test_rccl_exports_with_a_world_of_one(tmp_path):
    """The C-ABI collectives (hgx_index_broadcast, hgx_allreduce_sum_i64, hgx_classes_allgather) on an RCCL communicator of one
    rank that the process creates itself (ncclCommInitRank through ctypes, in a fresh process, before the first GPU work):
    the index block is unchanged, the sum is the input, the gathered + merged class set is the rank's own, and a locus typed
    through type_locus_sharded with this communicator equals the unsharded result."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rccl1.py"
    script.write_text(_RCCL_ONE_WORKER % root)
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "rccl world-1 ok" in out.stdout


_TWO_GPU_WORKER = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np
import torch
import torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
dist.init_process_group("nccl", device_id=torch.device("cuda", int(os.environ["LOCAL_RANK"])))
import hisatgenotype_amd as hgx
from hisatgenotype_amd import capi, synth, locus as hl, dist as hdist
capi.set_device(int(os.environ["LOCAL_RANK"]))
loc = synth.make_hla_like_locus(n_alleles=900, n_vars=700, seed=12, sibling_frac=0.4)
pl = hl.PackedLocus.from_synth(loc)
own = {k: v.copy() for k, v in pl.tables().items()}
sample = synth.pick_sample(loc, 3)
sam = synth.simulate_sam_fast(loc, sample, 9000, err_rate=0.004, seed=4)
ref = hgx.type_locus(pl, sam)                       # every rank: the unsharded result on its own index
# 1. torch-side broadcast straight into the index block; the receiver compares the block with the tables it packed itself
hdist.broadcast_index(pl, src=0)
host = hdist.index_block_tensor(pl.index()).cpu().numpy()
nb = pl.n_words * pl.a_pad
assert np.array_equal(host[:nb].view(np.uint32).reshape(pl.n_words, pl.a_pad), own["link_bits"]), "received index differs"
# 2. the same through the C-ABI on this library's own RCCL communicator, then a sharded locus over both forms of exchange
comm = hdist.RcclComm.from_torch()
comm.broadcast_index(pl, 0)
shard = hdist.split_name_grouped(sam, world)[rank]
for c in (comm, hdist.TorchComm()):
    got = hdist.type_locus_sharded(pl, shard, c)
    assert (got.num_reads, got.num_pairs) == (ref.num_reads, ref.num_pairs)
    assert got.counts_sorted == ref.counts_sorted and got.em == ref.em and got.gene_prob == ref.gene_prob
comm.close()
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_two_gpus_broadcast_and_sharded_locus(tmp_path):
    """More than one rank on hardware: needs >= 2 GPUs, skipped otherwise.  torch.distributed.run as a fresh child
    process (started before anything here touches a second GPU): RCCL broadcast into an index' device block checked against
    the receiver's own tables; a locus sharded over two ranks through RcclComm (device-side exchanges through the C-ABI) and
    through TorchComm equals the unsharded result."""
    import subprocess
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "two.py"
    script.write_text(_TWO_GPU_WORKER % root)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29641", str(script)], env=env, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert out.stdout.count("ok") == 2
