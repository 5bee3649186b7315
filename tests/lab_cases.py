"""Lab build only (libhgx_lab.so, -DHGX_LAB): the opt-in EM back-ends of rounds 1-2 -- int8-MFMA mat-vec, persistent kernel,
resident-grid kernel (csrc/lab/*.inc) -- against the product's default path.  Not collected by the suite itself (the product
library is what the suite's process has loaded): tests/test_gpu_lab.py runs this file with pytest in a child process."""
import numpy as np
import pytest

import golden_util as gu
import tables
from hisatgenotype_amd import capi, engine, locus as hl

capi.use_lab()


def _setup(name):
    fx = gu.load(name)
    loc = fx["_locus"]
    return fx, tables.oracle_tables(loc), hl.PackedLocus.from_synth(loc)


@pytest.mark.parametrize("name", ["hla_7000", "hla_mid_real"])
def test_em_backends_agree(name):
    """The table-lookup mat-vec (256 subset sums per group of 8 columns, one LDS lookup per 8 matrix bits) against the EXEC-masked
    FP64 VALU mat-vec of round 1 (lab back-end 1):
    same iteration counts, abundances equal to rounding, on the reference's recorded EM inputs and on a big random one."""
    fx, t, pl = _setup(name)
    A = t["n_alleles"]
    w = (A + 63) // 64
    cases = []
    for em in fx["em"]:
        rows = np.zeros((len(em["cmpt"]), pl.w64), np.uint64)
        for k, (cid, n) in enumerate(em["cmpt"]):
            rows[k, :w] = gu.class_bits(fx, cid, A)
        cases.append((rows, np.array([n for _, n in em["cmpt"]], np.int64), em["remove_low"], em["use_length"]))
    rng = np.random.RandomState(9)
    big = np.zeros((3000, pl.w64), np.uint64)
    dens = rng.choice([0.002, 0.05, 0.6], size=3000)
    for k in range(3000):
        m = rng.rand(A) < dens[k]
        m[rng.randint(A)] = True
        big[k, :w] = np.packbits(np.pad(m, (0, 64 * w - A)), bitorder="little").view(np.uint64)
    cases.append((big, rng.randint(1, 500, 3000).astype(np.int64), True, False))
    cases.append((big, rng.randint(1, 500, 3000).astype(np.int64), False, True))
    try:
        for rows, counts, low, use_len in cases:
            cl = engine.Classes.from_host(rows, counts, pl.a_pad)
            ln = pl.allele_len if use_len else None
            engine.em_set_backend(1)
            p1, it1 = cl.em(A, low, ln)
            import os
            for backend in (3,):       # table lookup, one launch per pass (the int8-MFMA form and the persistent kernel lost twice: removed in round 6)
                engine.em_set_backend(backend)
                p2, it2 = cl.em(A, low, ln)
                assert it1 == it2, (backend, it1, it2)
                assert np.array_equal(p1 < 0, p2 < 0)
                assert np.max(np.abs(p1 - p2)) <= 1e-11, (backend, np.max(np.abs(p1 - p2)))
    finally:
        engine.em_set_backend(0)


def test_round2_mid_size_em_is_the_reference_bit_for_bit(orc):
    """k_em_ref (round 2's mid-size EM in the reference's order; lab code since round 4: k_emx took its place) and k_em_small on the
    problems the product's k_emx takes: the same doubles as the C oracle / within 1e-9 for the tolerance kernel."""
    rng = np.random.RandomState(2)
    ran = 0
    for A, n_used, C_, dens in [(700, 300, 400, 0.05), (2000, 900, 1500, 0.02), (1500, 1000, 60, 0.3)]:
        a_pad = capi.a_pad(A)
        w64 = a_pad // 64
        used = np.sort(rng.choice(A, n_used, replace=False))
        fam = rng.rand(6, n_used) < dens * rng.choice([0.5, 1.0, 3.0], size=6)[:, None]
        classes, rows = [], np.zeros((C_, w64), np.uint64)
        name_rank = rng.permutation(A).astype(np.int32)
        for c in range(C_):
            m = fam[rng.randint(6)] ^ (rng.rand(n_used) < 0.02)
            m[rng.randint(n_used)] = True
            mem = used[m]
            mem = mem[np.argsort(name_rank[mem])]
            classes.append([int(a) for a in mem])
            for a in mem:
                rows[c, a >> 6] |= np.uint64(1) << np.uint64(a & 63)
        counts = rng.randint(1, 300, C_).astype(np.int64)
        cl = engine.Classes.from_host(rows, counts, a_pad)
        cl.set_allele_rank(name_rank)
        for low in (True, False):
            try:
                oa, op, oit = orc.single_abundance(A, classes, counts, low, None)
            except KeyError:
                continue
            exp = np.full(A, -1.0)
            exp[oa] = op
            with engine.test_switches(em_skip="emx", em_mid_nnz=100000000):
                p1, it1 = cl.em(A, low, None)
                assert engine.em_last_exact() and it1 == oit and np.array_equal(p1, exp)
            if C_ <= 64:
                with engine.test_switches(em_skip="emx,wave", em_no_mid=1):
                    p2, it2 = cl.em(A, low, None)
                assert it2 == oit and np.max(np.abs(p2 - exp)) <= 1e-9
            ran += 1
    assert ran >= 4


# ---- round 2's fused gene-level form (hgx_pair_classes_dedup: csrc/lab/hgx_fused_*.inc) ---------------------------------------------
@pytest.mark.parametrize("name", ["hla_small_pair", "hla_mid_real", "hla_errors_filters", "hla_novel_sample", "hla_7000", "codis_like"])
def test_fused_pair_classes_dedup_equals_two_call_form(name):
    """hgx_pair_classes_dedup (rows claimed / verified in registers, only representatives stored) == hgx_pair_classes +
    hgx_dedup_classes: bit rows, counts, first pairs, order -- on both levels."""
    fx = gu.load(name)
    loc = fx["_locus"]
    pl = hl.PackedLocus.from_synth(loc)
    o = fx["options"]
    batch = pl.parse_sam(fx["sam"], num_editdist=o["num_editdist"], error_correction=o["error_correction"], allow_discordant=o["allow_discordant"],
                         simulation=o["simulation"])
    db = engine.DeviceBatch(batch)
    bufs = engine.ScoreBuffers(pl, db)
    engine.score_pairs(pl, db, bufs)
    hla = loc.base_fname == "hla"
    for level in ((0, 1) if hla else (1,)):
        rows, hashes = (bufs.exon_bits, bufs.exon_hash) if level == 0 else (bufs.gene_bits, bufs.gene_hash)
        want = engine.Classes.dedup(rows, batch.n_pairs, pl.a_pad, hashes=hashes).to_host()
        rows.zero()
        got = engine.Classes.of_pairs_fused(pl, db, bufs, level).to_host()
        for x, y in zip(got, want):
            assert np.array_equal(x, y)


def test_fused_pair_classes_dedup_at_size_and_with_zero_rows():
    """200 k pairs of the bench locus (hot classes: a fifth of the pairs share one class; > 65 536 rows), pairs without refs
    at the level (all-zero rows are dropped) and pairs with hundreds of refs (the slab-wise path) through the fused form."""
    from hisatgenotype_amd import synth
    loc = synth.make_hla_like_locus(n_alleles=7000, n_vars=2500, seed=101)
    sample = synth.pick_sample(loc, 103)
    sam = synth.simulate_sam_fast(loc, sample, 200000, err_rate=0.002, seed=7)
    pl = hl.PackedLocus.from_synth(loc)
    batch = pl.parse_sam(sam)
    db = engine.DeviceBatch(batch)
    bufs = engine.ScoreBuffers(pl, db)
    engine.score_pairs(pl, db, bufs)
    want = engine.Classes.dedup(bufs.gene_bits, batch.n_pairs, pl.a_pad, hashes=bufs.gene_hash).to_host()
    bufs.gene_bits.zero()
    for _ in range(3):                                   # repeated: stale representatives of an earlier run must not matter
        got = engine.Classes.of_pairs_fused(pl, db, bufs, 1).to_host()
        for x, y in zip(got, want):
            assert np.array_equal(x, y)
    assert want[1].sum() == batch.n_pairs and want[1].max() > batch.n_pairs // 10
    # small hand-made batch: pairs with no refs at the exon level, duplicates, and a 400-ref pair
    t = tables.oracle_tables(loc)
    rng = np.random.RandomState(9)
    names = [n for n in loc.allele_names[1:] if n in loc.allele_vars]
    pair_off, level, left, right, id_off, ids = [0], [], [], [], [0], []
    def piece(lv):
        l = int(rng.randint(0, len(loc.backbone) - 200)); r = l + int(rng.randint(1, 150))
        a = names[rng.randint(len(names))]
        vs = [v for v in loc.allele_vars[a] if l <= loc.var_pos[v] <= r]
        level.append(lv); left.append(l); right.append(r); ids.extend(vs); id_off.append(len(ids))
    for n_refs, lvls in ((2, (1,)), (0, ()), (3, (0, 1)), (400, (0, 1)), (1, (1,)), (2, (0, 1))):
        for _ in range(n_refs):
            for lv in lvls:
                piece(lv)
        pair_off.append(len(level))
    arrs = (np.array(pair_off, np.int32), np.array(level, np.uint8), np.array(left, np.int32), np.array(right, np.int32),
            np.array(id_off, np.int32), np.array(ids or [0], np.int32))
    b2 = pl.batch_from_haplotypes(*arrs)
    d2 = engine.DeviceBatch(b2)
    f2 = engine.ScoreBuffers(pl, d2)
    engine.score_pairs(pl, d2, f2)
    for lv, rows, hs in ((0, f2.exon_bits, f2.exon_hash), (1, f2.gene_bits, f2.gene_hash)):
        want = engine.Classes.dedup(rows, b2.n_pairs, pl.a_pad, hashes=hs).to_host()
        rows.zero()
        got = engine.Classes.of_pairs_fused(pl, d2, f2, lv).to_host()
        for x, y in zip(got, want):
            assert np.array_equal(x, y)
