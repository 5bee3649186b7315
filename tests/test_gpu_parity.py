"""GPU parity: the HIP path (through the C-ABI) against the C oracle and the golden vectors."""
import numpy as np
import pytest

import golden_util as gu
import tables
from hisatgenotype_amd import engine, locus as hl

pytestmark = pytest.mark.gpu


def _setup(orc, name):
    fx = gu.load(name)
    loc = fx["_locus"]
    t = tables.oracle_tables(loc)
    pl = hl.PackedLocus.from_synth(loc)
    arrs = tables.pieces_from_pairs(fx["pairs"], t["var_index"])
    batch = pl.batch_from_haplotypes(*arrs)
    L = orc.make_locus(t)
    eb, gb, gc, fp = orc.score_pairs(L, t["exon_keys"], t["gene_keys"], *arrs)
    return fx, loc, t, pl, batch, (eb, gb, gc, fp)


@pytest.mark.parametrize("name", gu.ALL)
def test_score_pairs_bit_exact(orc, name):
    fx, loc, t, pl, batch, (eb, gb, gc, fp) = _setup(orc, name)
    db = engine.DeviceBatch(batch)
    bufs = engine.ScoreBuffers(pl, db)
    engine.score_pairs(pl, db, bufs)
    w = (t["n_alleles"] + 63) // 64
    got_g = bufs.gene_bits.to_host()[:batch.n_pairs]
    got_e = bufs.exon_bits.to_host()[:batch.n_pairs]
    assert np.array_equal(got_g[:, :w], gb) and not got_g[:, w:].any()
    if loc.base_fname == "hla":
        assert np.array_equal(got_e[:, :w], eb) and not got_e[:, w:].any()
    # golden classes straight from the reference
    A = t["n_alleles"]
    for i, p in enumerate(fx["pairs"]):
        assert np.array_equal(got_g[i, :w], gu.class_bits(fx, p["gene_cls"], A))
    # identical rows <=> identical hashes on this data
    gh = bufs.gene_hash.to_host()[:batch.n_pairs]
    keys = {}
    for i in range(batch.n_pairs):
        k = got_g[i].tobytes()
        assert keys.setdefault(k, gh[i]) == gh[i]
    assert len(set(keys.values())) == len(keys)


@pytest.mark.parametrize("name", gu.ALL)
def test_dedup_counts_em(orc, name):
    fx, loc, t, pl, batch, (eb, gb, gc, fp) = _setup(orc, name)
    hla = loc.base_fname == "hla"
    A = t["n_alleles"]
    w = (A + 63) // 64
    db = engine.DeviceBatch(batch)
    bufs = engine.ScoreBuffers(pl, db)
    engine.score_pairs(pl, db, bufs)
    # gene level: dedup + Gene_counts + tie order
    gcl = engine.Classes.dedup(bufs.gene_bits, batch.n_pairs, pl.a_pad, hashes=bufs.gene_hash)
    ub, uc, fr = orc.dedup(gb)
    hb, hc, hf = gcl.to_host()
    assert np.array_equal(hb[:, :w], ub) and np.array_equal(hc, uc) and np.array_equal(hf, fr)
    cnt, first = gcl.allele_counts()
    assert np.array_equal(cnt[:A], gc)
    first_pair = np.where(first[:A] >= 0, hf[np.clip(first[:A], 0, None)], -1)
    assert np.array_equal(first_pair, fp)
    # EM calls recorded from the reference
    lengths = pl.allele_len
    level_rows = bufs.exon_bits if hla else bufs.gene_bits
    level_hash = bufs.exon_hash if hla else bufs.gene_hash
    cl0 = engine.Classes.dedup(level_rows, batch.n_pairs, pl.a_pad, hashes=level_hash)
    em0 = fx["em"][0]
    assert cl0.n_classes == len(em0["cmpt"])
    b0, c0, _ = cl0.to_host()
    for k, (cid, n) in enumerate(em0["cmpt"]):
        assert np.array_equal(b0[k, :w], gu.class_bits(fx, cid, A)) and c0[k] == n
    for em in fx["em"]:
        rows = np.zeros((len(em["cmpt"]), pl.w64), np.uint64)
        for k, (cid, n) in enumerate(em["cmpt"]):
            rows[k, :w] = gu.class_bits(fx, cid, A)
        cl = engine.Classes.from_host(rows, [n for _, n in em["cmpt"]], pl.a_pad)
        prob, it = cl.em(A, em["remove_low"], lengths if em["use_length"] else None)
        assert it == em["n_iter"]
        exp = {a: float(p) for a, p in em["result"]}
        for a in range(A):
            n = t["names"][a]
            if n in exp:
                assert abs(prob[a] - exp[n]) <= 1e-9, (n, prob[a], exp[n])   # north_star tolerance is 1e-5
            else:
                assert prob[a] == -1.0


def test_filtered_dedup_matches_oracle(orc):
    fx, loc, t, pl, batch, (eb, gb, gc, fp) = _setup(orc, "hla_mid_real")
    A = t["n_alleles"]
    w = (A + 63) // 64
    ub, uc, _ = orc.dedup(gb)
    rng = np.random.RandomState(5)
    mask = np.zeros(pl.w64, np.uint64)
    keep = rng.rand(A) < 0.3
    for a in np.nonzero(keep)[0]:
        mask[a >> 6] |= np.uint64(1) << np.uint64(a & 63)
    exp_b, exp_c, _ = orc.dedup(ub, weight=uc, and_mask=mask[:w])
    rows = np.zeros((len(ub), pl.w64), np.uint64)
    rows[:, :w] = ub
    d_rows = engine.DevArray.from_host(rows)
    d_w = engine.DevArray.from_host(uc)
    d_m = engine.DevArray.from_host(mask)
    cl = engine.Classes.dedup(d_rows, len(ub), pl.a_pad, weights=d_w, and_mask=d_m)
    hb, hc, _ = cl.to_host()
    assert np.array_equal(hb[:, :w], exp_b) and np.array_equal(hc, exp_c)


def test_random_pieces_large(orc):
    """Seeded random haplotypes (many of them impossible: zero-compatible pairs, quirk Q4) on a mid-size locus."""
    from hisatgenotype_amd import synth
    loc = synth.make_hla_like_locus(n_alleles=1500, n_vars=1800, seed=77, unlinked_vars=6)
    t = tables.oracle_tables(loc)
    pl = hl.PackedLocus.from_synth(loc)
    rng = np.random.RandomState(3)
    V = len(loc.var_ids)
    pair_off, level, left, right, id_off, ids = [0], [], [], [], [0], []
    names = [n for n in loc.allele_names[1:] if n in loc.allele_vars]
    for p in range(3000):
        for m in range(rng.randint(0, 4)):
            l = rng.randint(0, len(loc.backbone) - 160)
            r = l + rng.randint(1, 150)
            if rng.rand() < 0.05:                 # a wide piece: more than 8 variant words (k_piece_compat's direct path)
                r = min(len(loc.backbone) - 1, l + rng.randint(400, 2500))
            a = names[rng.randint(len(names))]
            vs = [v for v in loc.allele_vars[a] if l <= loc.var_pos[v] <= r]
            if rng.rand() < 0.2 and vs:
                vs = vs[:-1]                      # drop a variant: usually incompatible with a
            if rng.rand() < 0.1:
                vs = vs + [int(rng.randint(V))]   # random extra variant
            if rng.rand() < 0.1:
                vs = vs + [-1]                    # novel id
            for lv in ((0, 1) if rng.rand() < 0.7 else (1,)):
                level.append(lv); left.append(l); right.append(r)
                ids += vs
                id_off.append(len(ids))
        pair_off.append(len(level))
    arrs = (np.array(pair_off, np.int32), np.array(level, np.uint8), np.array(left, np.int32), np.array(right, np.int32),
            np.array(id_off, np.int32), np.array(ids, np.int32))
    L = orc.make_locus(t)
    eb, gb, gc, fp = orc.score_pairs(L, t["exon_keys"], t["gene_keys"], *arrs)
    batch = pl.batch_from_haplotypes(*arrs)
    db = engine.DeviceBatch(batch)
    bufs = engine.ScoreBuffers(pl, db)
    engine.score_pairs(pl, db, bufs)
    w = (t["n_alleles"] + 63) // 64
    assert np.array_equal(bufs.gene_bits.to_host()[:, :w], gb)
    assert np.array_equal(bufs.exon_bits.to_host()[:, :w], eb)
    cl = engine.Classes.dedup(bufs.exon_bits, batch.n_pairs, pl.a_pad, hashes=bufs.exon_hash)
    ub, uc, fr = orc.dedup(eb)
    hb, hc, hf = cl.to_host()
    assert np.array_equal(hb[:, :w], ub) and np.array_equal(hc, uc) and np.array_equal(hf, fr)
    # EM on these classes vs the C oracle (same iteration count, weights within 1e-9)
    classes = []
    for row in ub:
        idx = []
        for wd, word in enumerate(row):
            word = int(word)
            while word:
                b = (word & -word).bit_length() - 1
                idx.append(64 * wd + b)
                word &= word - 1
        classes.append(sorted(idx, key=lambda a: pl.name_rank[a]))
    for low, ln in ((False, None), (True, pl.allele_len)):
        oa, op, it = orc.single_abundance(t["n_alleles"], classes, uc, low, ln)
        prob, git = cl.em(t["n_alleles"], low, ln)
        assert git == it
        exp = dict(zip(oa.tolist(), op.tolist()))
        for a in range(t["n_alleles"]):
            if a in exp:
                assert abs(prob[a] - exp[a]) <= 1e-9
            else:
                assert prob[a] == -1.0


@pytest.mark.parametrize("name", ["hla_7000", "codis_like"])
def test_piece_compat_any_piece_order(orc, name):
    """Stage 1 is order independent: a shuffled piece table gives the same rows (permuted)."""
    import ctypes as C
    from hisatgenotype_amd import capi
    fx, loc, t, pl, batch, _ = _setup(orc, name)
    L = capi.lib()
    assert np.all(np.diff(batch.pieces["lo_word"].astype(np.int64)) >= 0)      # the front-end emits window-sorted tables
    d_masks = engine.DevArray.from_host(batch.masks)
    perm = np.random.RandomState(1).permutation(batch.n_pieces)
    out = []
    import os
    for pieces in (batch.pieces, batch.pieces[perm]):         # (any piece order is correct: the windows just get shorter)
        d_p = engine.DevArray.from_host(np.ascontiguousarray(pieces))
        a = engine.DevArray((batch.n_pieces, pl.w64), np.uint64)
        capi.check(L.hgx_piece_compat(pl.index(), capi.ptr(d_p), capi.ptr(d_masks), C.c_int32(batch.n_pieces), capi.ptr(a), None))
        out.append(a.to_host())
    assert np.array_equal(out[0][perm], out[1])


@pytest.mark.parametrize("name", ["hla_7000", "hla_mid_real", "codis_like"])
def test_em_ordered_and_first_classes_match_counts_pass(orc, name):
    """hgx_em_ordered's first-class output (the tie order of the result list) and hgx_first_classes agree with the full
    Gene_counts pass (hgx_allele_counts) on every path: single wavefront, single workgroup, compact multi-launch + tail."""
    import os
    fx, loc, t, pl, batch, _ = _setup(orc, name)
    db = engine.DeviceBatch(batch)
    bufs = engine.ScoreBuffers(pl, db)
    engine.score_pairs(pl, db, bufs)
    A = t["n_alleles"]
    rows, hashes = (bufs.exon_bits, bufs.exon_hash) if loc.base_fname == "hla" else (bufs.gene_bits, bufs.gene_hash)
    cl = engine.Classes.dedup(rows, batch.n_pairs, pl.a_pad, hashes=hashes)
    _, first_full = cl.allele_counts()
    some = np.random.RandomState(2).choice(A, min(A, 40), replace=False).astype(np.int32)
    assert np.array_equal(cl.first_classes(some), first_full[some])
    for env in ({}, {"em_skip": "wave"}):
        for k, v in env.items():
            engine.test_switch(k, v)
        try:
            prob, first, it = cl.em_ordered(A, True, None)
            prob2, it2 = cl.em(A, True, None)
        finally:
            for k in env:
                engine.test_switch(k, None)
        assert it == it2 and np.array_equal(prob, prob2)
        present = prob >= 0
        assert present.any()
        assert np.array_equal(first[present], first_full[:A][present])
        assert np.all(first[~present] == -1)


def test_more_than_8192_alleles(orc):
    """9 100 alleles (HLA-B today): three 64-word groups per class row (the KW = 4 pair kernel), EM vectors longer than
    the register-cached 8 192 (strided SQUAREM / advance kernels, no fused prologue, no single-workgroup EM).  End to end
    through the front-end against the C oracle: class rows bit-exact, dedup exact, EM with the same iteration count."""
    from hisatgenotype_amd import synth
    loc = synth.make_hla_like_locus(n_alleles=9100, n_vars=2300, seed=31)
    t = tables.oracle_tables(loc)
    pl = hl.PackedLocus.from_synth(loc)
    assert pl.a_pad > 8192
    sample = synth.pick_sample(loc, 8)
    sam = synth.simulate_sam_fast(loc, sample, 4000, err_rate=0.002, seed=13)
    import pyref
    rl = pyref.RefLocus(loc)
    rl.score = False
    fe = rl.run(sam)
    arrs = tables.pieces_from_pairs(fe["pairs"], t["var_index"])
    L = orc.make_locus(t)
    eb, gb, gc, fp = orc.score_pairs(L, t["exon_keys"], t["gene_keys"], *arrs)
    batch = pl.parse_sam(sam)
    assert batch.n_pairs == len(fe["pairs"])
    db = engine.DeviceBatch(batch)
    bufs = engine.ScoreBuffers(pl, db)
    engine.score_pairs(pl, db, bufs)
    A = t["n_alleles"]
    w = (A + 63) // 64
    assert np.array_equal(bufs.gene_bits.to_host()[:batch.n_pairs, :w], gb)
    assert np.array_equal(bufs.exon_bits.to_host()[:batch.n_pairs, :w], eb)
    cl = engine.Classes.dedup(bufs.exon_bits, batch.n_pairs, pl.a_pad, hashes=bufs.exon_hash)
    ub, uc, fr = orc.dedup(eb)
    hb, hc, hf = cl.to_host()
    assert np.array_equal(hb[:, :w], ub) and np.array_equal(hc, uc) and np.array_equal(hf, fr)
    cnt, _ = engine.Classes.dedup(bufs.gene_bits, batch.n_pairs, pl.a_pad, hashes=bufs.gene_hash).allele_counts()
    assert np.array_equal(cnt[:A], gc)
    classes = []
    for row in ub:
        bits = np.unpackbits(row.view(np.uint8), bitorder="little")
        classes.append(sorted(np.nonzero(bits)[0].tolist(), key=lambda a: pl.name_rank[a]))
    for low in (True, False):
        oa, op, it = orc.single_abundance(A, classes, uc, low, None)
        prob, git = cl.em(A, low, None)
        assert git == it
        exp = np.full(A, -1.0)
        exp[oa] = op
        assert np.array_equal(prob < 0, exp < 0)
        assert np.max(np.abs(prob - exp)) <= 1e-9


@pytest.mark.parametrize("n_rows", [3000, 70000])
def test_forged_hash_collision_is_detected(orc, n_rows):
    """The exact verify pass: two DIFFERENT rows given the same 64-bit key are never merged -- the hash-table form (one-round-trip
    and two-round-trip sizes) re-keys the colliding rows and returns exactly the classes honest keys give; the radix-sort form
    reports HGX_ECOLLISION; honest keys pass."""
    import os
    from hisatgenotype_amd import capi
    a_pad = 1024
    w64 = a_pad // 64
    rng = np.random.RandomState(11)
    base = rng.randint(0, 2 ** 63, size=(40, w64), dtype=np.int64).astype(np.uint64)
    pick = rng.randint(0, 40, n_rows)
    rows = base[pick].copy()
    keys = (pick.astype(np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
    d_rows = engine.DevArray.from_host(rows)
    for _ in (0,):
        try:
            cl = engine.Classes.dedup(d_rows, n_rows, a_pad, hashes=engine.DevArray.from_host(keys))
            assert cl.n_classes == len(set(pick.tolist()))
            cl.close()
            victim = n_rows - 7                               # one row changes a bit but keeps its key
            forged = rows.copy()
            forged[victim, 3] ^= np.uint64(1) << np.uint64(17)
            d_forged = engine.DevArray.from_host(forged)
            got = engine.Classes.dedup(d_forged, n_rows, a_pad, hashes=engine.DevArray.from_host(keys))
            want = engine.Classes.dedup(d_forged, n_rows, a_pad)           # keys computed from the rows
            assert got.n_classes == want.n_classes == len(set(pick.tolist())) + 1
            for x, y in zip(got.to_host(), want.to_host()):
                assert np.array_equal(x, y)
            # many collisions at once, weighted: every third row flips a bit of its own and keeps its (now shared) key
            many = rows.copy()
            flip = np.arange(0, n_rows, 3)
            many[flip, 5] ^= (np.uint64(1) << (flip % 61).astype(np.uint64))
            wts = engine.DevArray.from_host(rng.randint(1, 6, n_rows).astype(np.int64))
            d_many = engine.DevArray.from_host(many)
            got = engine.Classes.dedup(d_many, n_rows, a_pad, hashes=engine.DevArray.from_host(keys), weights=wts)
            want = engine.Classes.dedup(d_many, n_rows, a_pad, weights=wts)
            assert got.n_classes == want.n_classes
            for x, y in zip(got.to_host(), want.to_host()):
                assert np.array_equal(x, y)
        finally:
            pass


@pytest.mark.parametrize("name", ["hla_7000", "hla_mid_real", "hla_errors_filters"])
def test_em_masked_equals_dedup_then_em(orc, name):
    """hgx_em_masked (filter gene classes to a set of alleles, merge, EM -- the hand-off, core:1752-1782) against the explicit
    route dedup(and_mask, weights) + em_ordered: with a small allele set (one-launch kernel), with the kernel switched off, and
    with > 64 alleles in the mask (general path)."""
    import os
    fx, loc, t, pl, batch, _ = _setup(orc, name)
    db = engine.DeviceBatch(batch)
    bufs = engine.ScoreBuffers(pl, db)
    engine.score_pairs(pl, db, bufs)
    A = t["n_alleles"]
    gcl = engine.Classes.dedup(bufs.gene_bits, batch.n_pairs, pl.a_pad, hashes=bufs.gene_hash)
    cnt, _ = gcl.allele_counts()
    top = np.argsort(-cnt[:A], kind="stable")
    gb, gc, _ = gcl.device_ptrs()
    for n_keep in (3, 20, 64, 65, 300):
        keep = np.sort(top[:min(n_keep, A)])
        mask = np.zeros(pl.w64, np.uint64)
        np.bitwise_or.at(mask, keep >> 6, np.uint64(1) << (keep & 63).astype(np.uint64))
        d_mask = engine.DevArray.from_host(mask)
        sub = engine.Classes.dedup(engine._RawDev(gb), gcl.n_classes, pl.a_pad, weights=engine._RawDev(gc), and_mask=d_mask)
        p_ref, f_ref, it_ref = sub.em_ordered(A, True, pl.allele_len)
        for env in ({}, {"em_skip": "masked"}):
            for k, v in env.items():
                engine.test_switch(k, v)
            try:
                p, f, it, nc = gcl.em_masked(mask, A, True, pl.allele_len)
            finally:
                for k in env:
                    engine.test_switch(k, None)
            assert it == it_ref and nc == sub.n_classes, (n_keep, env, it, it_ref, nc, sub.n_classes)
            assert np.array_equal(p < 0, p_ref < 0)
            assert np.max(np.abs(p - p_ref)) <= 1e-12
            assert np.array_equal(f, f_ref)
        sub.close()


def test_empty_and_degenerate_inputs(orc):
    """Zero pairs, pairs without pieces, and a locus whose exon level has no representatives."""
    from hisatgenotype_amd import synth
    loc = synth.make_hla_like_locus(n_alleles=70, n_vars=90, seed=4)
    pl = hl.PackedLocus.from_synth(loc)
    empty = pl.batch_from_haplotypes(np.zeros(1, np.int32), [], [], [], np.zeros(1, np.int32), [])
    assert empty.n_pairs == 0
    cl = engine.Classes.dedup(engine.DevArray((1, pl.w64), np.uint64), 0, pl.a_pad)
    assert cl.n_classes == 0
    prob, it = cl.em(pl.n_alleles)
    assert it == 1 and np.all(prob == -1.0)      # (single_abundance({}) makes one pass of its loop and returns [])
    # three pairs with no piece at all: every level's class is its whole allele mask (quirk Q4)
    b = pl.batch_from_haplotypes(np.zeros(4, np.int32), [], [], [], np.zeros(1, np.int32), [])
    db = engine.DeviceBatch(b)
    bufs = engine.ScoreBuffers(pl, db)
    engine.score_pairs(pl, db, bufs)
    t = pl.tables()
    assert np.array_equal(bufs.gene_bits.to_host(), np.tile(t["gene_mask"], (3, 1)))
    assert np.array_equal(bufs.exon_bits.to_host(), np.tile(t["exon_mask"], (3, 1)))
    gcl = engine.Classes.dedup(bufs.gene_bits, 3, pl.a_pad, hashes=bufs.gene_hash)
    assert gcl.n_classes == 1 and gcl.to_host()[1][0] == 3


def test_em_single_workgroup_path_equals_multi_launch_path(orc):
    """Problems with <= 64 classes run in one workgroup (k_em_small); both paths must agree to rounding."""
    import os
    fx, loc, t, pl, batch, (eb, gb, gc, fp) = _setup(orc, "hla_mid_real")
    A = t["n_alleles"]
    w = (A + 63) // 64
    ub, uc, _ = orc.dedup(eb)
    rows = np.zeros((40, pl.w64), np.uint64)
    rows[:, :w] = ub[:40]
    cl = engine.Classes.from_host(rows, uc[:40], pl.a_pad)
    for low, ln in ((False, None), (True, pl.allele_len), (True, None)):
        p_small, it_small = cl.em(A, low, ln)
        engine.test_switch("em_skip", "wave")
        try:
            p_big, it_big = cl.em(A, low, ln)
        finally:
            engine.test_switch("em_skip", None)
        assert it_small == it_big
        assert np.array_equal(p_small < 0, p_big < 0)
        assert np.max(np.abs(p_small - p_big)) <= 1e-12


def test_em_single_wavefront_path_matches_oracle_and_other_paths(orc):
    """<= 64 classes over <= 64 distinct alleles (hand-off EM, STR loci) run in one wavefront (k_em_wave).  Random
    problems scattered over a wide allele range: same iteration count and abundances as the C oracle and as the
    single-workgroup path; 65 distinct alleles must fall through to the other paths and still agree."""
    import os
    rng = np.random.RandomState(31)
    for trial in range(24):
        A = int(rng.choice([40, 700, 7000]))
        a_pad = engine.capi.a_pad(A)
        w64 = a_pad // 64
        n_used = 65 if trial == 23 else int(rng.randint(1, 65))
        n_used = min(n_used, A)
        used = np.sort(rng.choice(A, n_used, replace=False))
        C_ = int(rng.randint(1, 65))
        classes, rows = [], np.zeros((C_, w64), np.uint64)
        for c in range(C_):
            k = int(rng.randint(1, max(2, min(n_used, 12))))
            mem = np.sort(rng.choice(used, min(k, n_used), replace=False))
            if trial == 23 and c == 0:
                mem = used
            classes.append([int(a) for a in mem])
            for a in mem:
                rows[c, a >> 6] |= np.uint64(1) << np.uint64(a & 63)
        counts = rng.randint(1, 300, C_).astype(np.int64)
        lengths = rng.randint(200, 3500, A).astype(np.int32)
        cl = engine.Classes.from_host(rows, counts, a_pad)
        for low, ln in ((True, None), (False, lengths), (True, lengths)):
            try:
                oa, op, oit = orc.single_abundance(A, classes, counts, low, ln)
            except KeyError:
                continue
            p_w, it_w = cl.em(A, low, ln)
            engine.test_switch("em_skip", "wave")
            try:
                p_s, it_s = cl.em(A, low, ln)
            finally:
                engine.test_switch("em_skip", None)
            assert it_w == oit == it_s, (trial, it_w, oit, it_s)
            exp = np.full(A, -1.0)
            exp[oa] = op
            assert np.array_equal(p_w < 0, exp < 0)
            assert np.max(np.abs(p_w - exp)) <= 1e-9
            assert np.max(np.abs(p_w - p_s)) <= 1e-9


def test_em_mid_size_in_reference_order_is_bit_identical(orc):
    """65 ... 2048 classes over hundreds to thousands of distinct alleles in the reference's own order of operations (k_emx,
    hgx_emx.hip, since round 3: up to 4096 classes x 8192 alleles; k_em_ref, round 2's one-workgroup kernel for <= 1024 alleles,
    is still reachable with the test switch em_skip=emx): abundances `==` the C oracle's (which the golden vectors pin to the real
    reference), same iteration counts, with pruning, with allele lengths, with alleles scattered over a wide index range and an
    arbitrary name order.  The table-lookup path on the same problems (both switched off) agrees to 1e-9."""
    import os
    rng = np.random.RandomState(2024)
    cases = [(300, 90, 120, 0.10), (700, 400, 600, 0.03), (7000, 1024, 2048, 0.01), (1200, 1000, 300, 0.2), (5000, 200, 1500, 0.3),
             (2000, 1100, 500, 0.05)]
    ran_exact = 0
    for A, n_used, C_, dens in cases:
        a_pad = engine.capi.a_pad(A)
        w64 = a_pad // 64
        used = np.sort(rng.choice(A, n_used, replace=False))
        fam = rng.rand(6, n_used) < dens * rng.choice([0.5, 1.0, 3.0], size=6)[:, None]
        classes, rows = [], np.zeros((C_, w64), np.uint64)
        name_rank = rng.permutation(A).astype(np.int32)           # name order != index order
        for c in range(C_):
            m = fam[rng.randint(6)] ^ (rng.rand(n_used) < 0.02)
            m[rng.randint(n_used)] = True
            mem = used[m]
            mem = mem[np.argsort(name_rank[mem])]                # class key = alleles in name order
            classes.append([int(a) for a in mem])
            for a in mem:
                rows[c, a >> 6] |= np.uint64(1) << np.uint64(a & 63)
        counts = rng.randint(1, 300, C_).astype(np.int64)
        lengths = rng.randint(200, 3500, A).astype(np.int32)
        cl = engine.Classes.from_host(rows, counts, a_pad)
        cl.set_allele_rank(name_rank)
        for low, ln in ((True, None), (False, lengths), (True, lengths)):
            try:
                oa, op, oit = orc.single_abundance(A, classes, counts, low, ln)
            except KeyError:
                continue
            p, it = cl.em(A, low, ln)
            exp = np.full(A, -1.0)
            exp[oa] = op
            assert it == oit, (A, n_used, C_, low, it, oit)
            assert np.array_equal(p < 0, exp < 0)
            assert engine.em_last_exact()
            assert np.array_equal(p, exp), (A, n_used, C_, low, float(np.max(np.abs(p - exp))))
            ran_exact += 1
            engine.test_switch("em_skip", "emx")                  # the table-lookup path: close, not identical
            try:
                p2, it2 = cl.em(A, low, ln)
                assert not engine.em_last_exact()
            finally:
                engine.test_switch("em_skip", None)
            assert it2 == it and np.max(np.abs(p2 - p)) <= 1e-9
    assert ran_exact >= 15


def test_em_compact_tail_equals_full_iterations(orc):
    """After pruning leaves <= 64 alleles the EM finishes on one wavefront over merged 64-bit class masks (k_em_tail).
    Same iteration count and abundances as iterating over the whole matrix, and as the C oracle."""
    import os
    rng = np.random.RandomState(77)
    for A, C_ in ((900, 400), (7000, 3000), (3000, 9000)):
        a_pad = engine.capi.a_pad(A)
        w64 = a_pad // 64
        truth = rng.choice(A, 3, replace=False)
        classes, rows = [], np.zeros((C_, w64), np.uint64)
        for c in range(C_):
            k = int(rng.randint(1, 40))
            mem = set(int(a) for a in rng.choice(A, k, replace=False))
            if rng.rand() < 0.8:
                mem.add(int(truth[rng.randint(3)]))
            mem = sorted(mem)
            classes.append(mem)
            for a in mem:
                rows[c, a >> 6] |= np.uint64(1) << np.uint64(a & 63)
        counts = rng.randint(1, 50, C_).astype(np.int64)
        lengths = rng.randint(2000, 3500, A).astype(np.int32)
        cl = engine.Classes.from_host(rows, counts, a_pad)
        for low, ln in ((True, None), (True, lengths)):
            oa, op, oit = orc.single_abundance(A, classes, counts, low, ln)
            p_t, it_t = cl.em(A, low, ln)
            engine.test_switch("em_skip", "tail")
            try:
                p_f, it_f = cl.em(A, low, ln)
            finally:
                engine.test_switch("em_skip", None)
            assert it_t == it_f == oit, (A, C_, it_t, it_f, oit)
            assert oit > 11                                   # the tail really took over
            exp = np.full(A, -1.0)
            exp[oa] = op
            assert np.array_equal(p_t < 0, exp < 0) and np.array_equal(p_f < 0, exp < 0)
            assert np.max(np.abs(p_t - exp)) <= 1e-9 and np.max(np.abs(p_t - p_f)) <= 1e-9


def _level_equals_per_pair(pl, batch, collided=False):
    db = engine.DeviceBatch(batch)
    bufs = engine.ScoreBuffers(pl, db)
    engine.score_pairs(pl, db, bufs)
    two = bufs.exon_bits is not None
    ref = {}
    if two:
        ref[0] = engine.Classes.dedup(bufs.exon_bits, batch.n_pairs, pl.a_pad, hashes=bufs.exon_hash)
    ref[1] = engine.Classes.dedup(bufs.gene_bits, batch.n_pairs, pl.a_pad, hashes=bufs.gene_hash)
    want = {lv: c.to_host() for lv, c in ref.items()}
    for lv in want:
        for scratch in (True, False):
            b2 = bufs if scratch else type("B", (), {"compat": bufs.compat, "exon_bits": None, "exon_hash": None,
                                                     "gene_bits": None, "gene_hash": None})()
            groups = None if scratch else engine.Groups(db, lv)         # two-step form (grouping queued separately)
            got = engine.Classes.of_level(pl, db, b2, lv, groups=groups)
            gh = got.to_host()
            if groups is not None:
                assert (groups.n_groups == 0) if collided else (0 < groups.n_groups <= batch.n_pairs)
                groups.close()
            assert got.n_classes == ref[lv].n_classes
            for x, y in zip(gh, want[lv]):
                assert np.array_equal(x, y)
            got.close()
    for c in ref.values():
        c.close()
    return {lv: len(w[1]) for lv, w in want.items()}


@pytest.mark.parametrize("name", gu.ALL)
def test_level_classes_equals_per_pair_rows_then_dedup(orc, name):
    """hgx_level_classes (pairs grouped by ref list, one row per distinct list, weighted row dedup) gives exactly the classes,
    counts, order and first pairs of the per-pair form on every golden fixture, with and without caller scratch."""
    fx, loc, t, pl, batch, _ = _setup(orc, name)
    _level_equals_per_pair(pl, batch)


def test_level_classes_many_ref_lists_and_repeats():
    """> 65536 distinct ref lists (the dedup's large path behind the grouping), heavy repeats, pairs without refs at a level,
    the same pieces in another order (a separate group that must merge again at the row level)."""
    from hisatgenotype_amd import synth
    loc = synth.make_hla_like_locus(n_alleles=900, n_vars=1500, seed=12)
    pl = hl.PackedLocus.from_synth(loc)
    rng = np.random.RandomState(8)
    names = [n for n in loc.allele_names[1:] if n in loc.allele_vars]
    protos = []
    for _ in range(700):                      # distinct pieces to draw from
        l = rng.randint(0, len(loc.backbone) - 200)
        r = l + rng.randint(20, 180)
        a = names[rng.randint(len(names))]
        vs = [v for v in loc.allele_vars[a] if l <= loc.var_pos[v] <= r]
        if rng.rand() < 0.15 and vs:
            vs = vs[1:]
        protos.append((l, r, vs))
    pair_off, level, left, right, id_off, ids = [0], [], [], [], [0], []
    n_pairs = 250000
    pick = rng.randint(0, len(protos), size=(n_pairs, 3))
    n_in = rng.randint(0, 4, size=n_pairs)
    lvl_mode = rng.randint(0, 4, size=n_pairs)
    for p in range(n_pairs):
        for m in range(n_in[p]):
            l, r, vs = protos[pick[p, m]]
            for lv in ((0, 1), (1,), (0,), (1, 0))[lvl_mode[p]]:
                level.append(lv); left.append(l); right.append(r)
                ids += vs
                id_off.append(len(ids))
        pair_off.append(len(level))
    arrs = (np.array(pair_off, np.int32), np.array(level, np.uint8), np.array(left, np.int32), np.array(right, np.int32),
            np.array(id_off, np.int32), np.array(ids, np.int32))
    batch = pl.batch_from_haplotypes(*arrs)
    off, ref = np.asarray(batch.pair_off), np.asarray(batch.pair_ref)
    lists = {tuple(ref[off[p]:off[p + 1]][(ref[off[p]:off[p + 1]] >> 31) == 1].tolist()) for p in range(n_pairs)}
    assert len(lists) > 65536
    n_cls = _level_equals_per_pair(pl, batch)
    assert n_cls[1] < len(lists)              # different lists, same class: merged by the row dedup


def test_allele_counts_direct_equals_matvec_form(orc, monkeypatch):
    """Gene_counts straight from the row-major class matrix (integer column sums + first classes) against the two-pass bit
    mat-vec over the transposed matrix, on real class sets and on random ones; counts >= 2^24 take the mat-vec form."""
    fx, loc, t, pl, batch, _ = _setup(orc, "hla_7000")
    db = engine.DeviceBatch(batch)
    bufs = engine.ScoreBuffers(pl, db)
    engine.score_pairs(pl, db, bufs)
    sets = [engine.Classes.dedup(bufs.gene_bits, batch.n_pairs, pl.a_pad, hashes=bufs.gene_hash)]
    rng = np.random.RandomState(11)
    # (the last one: class counts beyond 32 bits -- hand-made sets only -- take the plain 64-bit kernel, k_allele_counts_wide)
    for n_cls, density, hi in ((1, 0.5, 100), (37, 0.02, 5), (3000, 0.3, 1 << 20), (700, 0.1, 1 << 30), (90, 0.2, 1 << 44)):
        bits = np.zeros((n_cls, pl.w64), np.uint64)
        for w in range(pl.w64):
            m = rng.rand(n_cls, 64) < density
            bits[:, w] = (m * (np.uint64(1) << np.arange(64, dtype=np.uint64))).sum(axis=1).astype(np.uint64)
        cnt = rng.randint(1, hi, n_cls).astype(np.int64)
        sets.append(engine.Classes.from_host(bits, cnt, pl.a_pad))
    for cl in sets:
        got_c, got_f = cl.allele_counts()
        b, c, _ = cl.to_host()
        member = np.unpackbits(np.ascontiguousarray(b).view(np.uint8), axis=1, bitorder="little").astype(bool)     # [class][allele]
        exp_c = (member * c[:, None].astype(object)).sum(axis=0) if c.max() >= (1 << 31) else (member.astype(np.int64) * c[:, None]).sum(axis=0)
        exp_f = np.where(member.any(axis=0), member.argmax(axis=0), -1)
        assert np.array_equal(got_c, np.asarray(exp_c, dtype=np.int64)) and np.array_equal(got_f, exp_f)
        a = int(np.flatnonzero(got_c)[0]) if got_c.any() else 0
        col = (b[:, a >> 6] >> np.uint64(a & 63)) & np.uint64(1)
        assert got_c[a] == int(c[col == 1].sum()) and got_f[a] == (int(np.flatnonzero(col)[0]) if col.any() else -1)
        cl.close()


def test_level_classes_falls_back_to_per_pair_rows_on_a_list_key_collision(orc, monkeypatch):
    """Two different ref lists with one 64-bit list key (never seen; forced here) make hgx_level_classes take the per-pair form:
    same classes, counts, order and first pairs."""
    fx, loc, t, pl, batch, _ = _setup(orc, "hla_errors_filters")
    engine.test_switch("test_group_collision", "1")
    _level_equals_per_pair(pl, batch, collided=True)


def test_pairs_with_hundreds_of_refs_and_very_wide_pieces(orc):
    """No 255 ceilings (the reference has none, typing_core.py:1250-1270): pairs with 16 / 255 / 256 / 700 /
    1500 add_count calls per level (the slab-wise 16-plane counters beyond 255, on their own and beside a short pair in the
    two-pairs-per-wavefront kernel, in the two-level kernel and in the single-level one) and pieces spanning more than 255
    variant words, bit-exact against the C oracle -- class rows of both levels and, through them, the row hashes' dedup."""
    from hisatgenotype_amd import synth
    loc = synth.make_hla_like_locus(n_alleles=2500, n_vars=9000, length=30000, seed=78, unlinked_vars=6)
    t = tables.oracle_tables(loc)
    pl = hl.PackedLocus.from_synth(loc)
    rng = np.random.RandomState(5)
    names = [n for n in loc.allele_names[1:] if n in loc.allele_vars]
    pair_off, level, left, right, id_off, ids = [0], [], [], [], [0], []

    def piece(lv, wide=False):
        l = rng.randint(0, len(loc.backbone) - 200)
        r = l + rng.randint(1, 150)
        if wide:
            l, r = rng.randint(0, 300), len(loc.backbone) - 1 - rng.randint(0, 300)      # nearly the whole locus: > 255 words
        a = names[rng.randint(len(names))]
        vs = [v for v in loc.allele_vars[a] if l <= loc.var_pos[v] <= r]
        if rng.rand() < 0.3 and vs:
            vs = vs[:-1]
        level.append(lv); left.append(l); right.append(r)
        ids.extend(vs)
        id_off.append(len(ids))

    for n_refs in (2, 700, 1, 16, 255, 256, 3, 1500, 300, 2):
        for k in range(n_refs):
            for lv in (0, 1):
                piece(lv, wide=(k % 97 == 5))
        pair_off.append(len(level))
    arrs = (np.array(pair_off, np.int32), np.array(level, np.uint8), np.array(left, np.int32), np.array(right, np.int32),
            np.array(id_off, np.int32), np.array(ids, np.int32))
    L = orc.make_locus(t)
    eb, gb, gc, fp = orc.score_pairs(L, t["exon_keys"], t["gene_keys"], *arrs)
    batch = pl.batch_from_haplotypes(*arrs)
    assert int(batch.pieces["n_words"].max()) > 255
    db = engine.DeviceBatch(batch)
    bufs = engine.ScoreBuffers(pl, db)
    w = (t["n_alleles"] + 63) // 64
    engine.score_pairs(pl, db, bufs)                                   # both levels in one launch (one pair per wavefront)
    assert np.array_equal(bufs.gene_bits.to_host()[:, :w], gb) and np.array_equal(bufs.exon_bits.to_host()[:, :w], eb)
    h2 = bufs.gene_hash.to_host()
    bufs.gene_bits.zero()
    engine.pair_classes(pl, db, bufs, exon=False)                      # one level: two pairs per wavefront
    assert np.array_equal(bufs.gene_bits.to_host()[:, :w], gb) and np.array_equal(bufs.gene_hash.to_host(), h2)
    bufs.exon_bits.zero()
    engine.pair_classes(pl, db, bufs, gene=False)
    assert np.array_equal(bufs.exon_bits.to_host()[:, :w], eb)
    cl = engine.Classes.of_level(pl, db, bufs, 1)                      # grouped by ref list
    ub, uc, fr = orc.dedup(gb)
    hb, hc, _ = cl.to_host()
    assert np.array_equal(hb[:, :w], ub) and np.array_equal(hc, uc)


COMPAT_FORMS = [{}]           # (tests/lab_cases.py adds the lab build's comparison kernels)


@pytest.mark.parametrize("n_alleles,n_vars,dense", [(700, 300, False), (5000, 700, False), (3000, 260, True), (9000, 500, True)])
def test_piece_compat_kernels_agree_with_the_definition(n_alleles, n_vars, dense):
    """hgx_piece_compat straight through the C-ABI on a random index: the pattern form (word tests once per DISTINCT value of a
    variant word; the LDS-tiled and L2-served kernels of rounds 1-3 are lab code: tests/lab_cases.py runs this test's body on
    them) against compat(a) <=> for every word i: (bits[lo + i][a] & MP_i) == P_i in numpy.  `dense`: independent random bits -- every allele
    has its own value in every word (more than HGX_PAT_D = 512 of them), so the pattern form takes its straight-from-the-index
    path; otherwise alleles copy a few hundred founders, as real loci do (a few hundred values per word at most)."""
    import ctypes as C
    from hisatgenotype_amd import capi
    rng = np.random.RandomState(n_alleles + n_vars)
    a_pad = capi.a_pad(n_alleles)
    n_words = (n_vars + 31) // 32
    bits = np.zeros((n_words, a_pad), np.uint32)
    if dense:
        bits[:, :n_alleles] = rng.randint(0, 1 << 32, size=(n_words, n_alleles), dtype=np.uint64).astype(np.uint32) & \
            rng.randint(0, 1 << 32, size=(n_words, n_alleles), dtype=np.uint64).astype(np.uint32)
    else:
        founders = rng.randint(0, 1 << 32, size=(n_words, 150), dtype=np.uint64).astype(np.uint32) & \
            rng.randint(0, 1 << 32, size=(n_words, 150), dtype=np.uint64).astype(np.uint32)
        bits[:, :n_alleles] = founders[:, rng.randint(0, 150, n_alleles)]
        flip = rng.rand(n_words, n_alleles) < 0.02                       # private variants
        bits[:, :n_alleles] ^= (flip * (1 << rng.randint(0, 32, size=(n_words, n_alleles)))).astype(np.uint32)
    if n_vars % 32:
        bits[-1] &= np.uint32((1 << (n_vars % 32)) - 1)
    em = np.zeros(a_pad // 64, np.uint64)
    ix = C.c_void_p()
    L = capi.lib()
    capi.check(L.hgx_index_create(C.byref(ix), C.c_int32(n_alleles), C.c_int32(n_vars), capi.ptr(bits), capi.ptr(em), capi.ptr(em)))
    try:
        n_pieces = 3000
        pieces = np.zeros(n_pieces, capi.PIECE_DTYPE)
        masks = []
        for p in range(n_pieces):
            nw = int(rng.choice([1, 2, 3, 4, 5, 8, 9, 12])) if rng.rand() < 0.3 else int(rng.randint(1, 6))
            nw = min(nw, n_words)
            lo = int(rng.randint(0, n_words - nw + 1))
            pieces[p] = (len(masks), lo, nw)
            a = int(rng.randint(n_alleles))                             # masks cut from a real allele, so that some alleles pass
            for i in range(nw):
                mp = int(rng.randint(0, 1 << 32, dtype=np.uint64)) & int(rng.randint(0, 1 << 32, dtype=np.uint64))
                pw = int(bits[lo + i, a]) & mp
                if rng.rand() < 0.03:
                    pw ^= mp & (1 << int(rng.randint(32)))
                masks += [mp, pw]
        order = np.lexsort((pieces["n_words"], pieces["lo_word"]))      # the front end's order: by first word, then width
        pieces = pieces[order]
        masks = np.array(masks, np.uint32)
        want = np.zeros((n_pieces, a_pad // 64), np.uint64)
        for p in range(n_pieces):
            mo, lo, nw = int(pieces[p]["mask_off"]), int(pieces[p]["lo_word"]), int(pieces[p]["n_words"])
            ok = np.ones(a_pad, bool)
            for i in range(nw):
                ok &= (bits[lo + i] & masks[mo + 2 * i]) == masks[mo + 2 * i + 1]
            want[p] = np.packbits(ok, bitorder="little").view(np.uint64)
        d_pieces, d_masks = capi.DevArray.from_host(pieces), capi.DevArray.from_host(masks)
        for sw in COMPAT_FORMS:
            out = capi.DevArray((n_pieces, a_pad // 64), np.uint64)
            out.zero()
            with engine.test_switches(**sw):
                capi.check(L.hgx_piece_compat(ix, capi.ptr(d_pieces), capi.ptr(d_masks), C.c_int32(n_pieces), capi.ptr(out), None))
            capi.sync()
            got = out.to_host()
            assert np.array_equal(got, want), (sw, int((got != want).any(axis=1).sum()))
    finally:
        L.hgx_index_destroy(ix)
