"""Lab build only (libhgx_lab.so, -DHGX_LAB): the per-key / per-pair functions of the DEVICE front end (csrc/hgx_front_core.hpp:
CIGAR x MD x Zs walk, error correction, ambiguity sets, haplotypes, exon pieces, piece masks, pair protocol -- the code the kernels
of hgx_front.hip run one lane per key) executed as loops on the CPU (hgx_front_host.cpp, hgx_front_emulate) against the pinned host
front end (hgx_sam.cpp).  Batches must be identical byte for byte.  Not collected by the suite itself: tests/test_front_emulation.py
runs this file with pytest in a child process (the suite's own process has the product library loaded)."""
import ctypes as C
import random

import numpy as np
import pytest

import golden_util as gu
from d18_cases import d18s51_cases
from hisatgenotype_amd import capi, locus as hl, synth

capi.use_lab()


def emulated(pl, sam, num_editdist=2, error_correction=True, allow_discordant=False, simulation=False, n_threads=0, keep_trace=False):
    data = sam if isinstance(sam, (bytes, bytearray)) else sam.encode()
    o = capi.ParseOpts(num_editdist, int(error_correction), int(allow_discordant), int(simulation), 0, int(keep_trace),
                       int(pl.base_fname == "codis" and pl.gene == "D18S51"), int(n_threads))
    h, dec = C.c_void_p(), C.c_int32(0)
    capi.check(capi.lib().hgx_lab_parse_sam_emulated(C.byref(h), pl.h, data, C.c_size_t(len(data)), C.byref(o), C.byref(dec)))
    return hl.Batch(h), dec.value


def emulated_records(pl, sam=None, path=None, regions=None, num_editdist=2, error_correction=True, allow_discordant=False, simulation=False,
                     keep_trace=False):
    """The record route: fields, filters and key grouping emulated too (hgx_lab_parse_records_emulated); text or a SAM / BAM file."""
    data = None if sam is None else (sam if isinstance(sam, (bytes, bytearray)) else sam.encode())
    o = capi.ParseOpts(num_editdist, int(error_correction), int(allow_discordant), int(simulation), 0, int(keep_trace),
                       int(pl.base_fname == "codis" and pl.gene == "D18S51"), 0)
    h, dec = C.c_void_p(), (C.c_int32 * 2)(0, 0)
    capi.check(capi.lib().hgx_lab_parse_records_emulated(C.byref(h), pl.h, data, C.c_size_t(len(data) if data else 0),
                                                         path.encode() if path else None, regions.encode() if regions else None,
                                                         C.byref(o), dec))
    return hl.Batch(h), (dec[0], dec[1])


def same_batch(a, b, length):
    assert (a.n_reads, a.n_pairs, a.n_pieces, a.n_refs, a.n_mask_u32) == (b.n_reads, b.n_pairs, b.n_pieces, b.n_refs, b.n_mask_u32)
    assert a.pieces.tobytes() == b.pieces.tobytes()
    assert a.masks.tobytes() == b.masks.tobytes()
    assert a.pair_off.tobytes() == b.pair_off.tobytes()
    assert a.pair_ref.tobytes() == b.pair_ref.tobytes()
    na, ca = a.pileup(length)
    nb, cb = b.pileup(length)
    assert np.array_equal(na, nb) and np.array_equal(ca, cb)


@pytest.mark.parametrize("name", gu.ALL + gu.LEAN)
def test_emulated_device_stages_equal_the_host_front_end_on_every_fixture(name):
    fx = gu.load(name)
    o = fx["options"]
    pl = hl.PackedLocus.from_synth(fx["_locus"])
    kw = dict(num_editdist=o["num_editdist"], error_correction=o["error_correction"], allow_discordant=o["allow_discordant"],
              simulation=o["simulation"])
    host = pl.parse_sam(fx["sam"], **kw)
    emu, declined = emulated(pl, fx["sam"], **kw)
    assert declined == 0, declined               # (codis_d18s51 too: choose_pairs, typing_core.py:1547-1552, on the stream's last pair)
    same_batch(host, emu, len(fx["_locus"].backbone))


@pytest.mark.parametrize("name", gu.ALL)
def test_emulated_kernels_against_the_reference_per_record(name, tmp_path):
    """DIRECTLY against the reference's recorded intermediates, not through the host front end: the pileup tables the emulated
    k_fe_pileup / k_fe_nt_set made == get_mpileup's (G5), and -- keep_trace on the device route -- cmp_list2, cmp_left / cmp_right and
    both alternative sets of every kept record as fe_key computed them == what the reference's loop held (G3:
    identify_ambigious_diffs typing_common.py:1663-1955 after error_correct typing_core.py:119-243), by both routes and from a BAM."""
    from hisatgenotype_amd import bamio
    from trace_util import check_pileup, check_trace
    fx = gu.load(name)
    o = fx["options"]
    loc = fx["_locus"]
    pl = hl.PackedLocus.from_synth(loc)
    kw = dict(num_editdist=o["num_editdist"], error_correction=o["error_correction"], allow_discordant=o["allow_discordant"],
              simulation=o["simulation"], keep_trace=True)
    p_bam = str(tmp_path / "s.bam")
    bamio.write_bam_native(p_bam, fx["sam"].encode(), [(loc.ref_allele, len(loc.backbone))], sort_by_coordinate=True)
    emu, dec = emulated(pl, fx["sam"], **kw)
    emu_r, dec_r = emulated_records(pl, sam=fx["sam"], **kw)
    emu_b, dec_b = emulated_records(pl, path=p_bam, regions=loc.ref_allele, **kw)
    assert dec == 0 and dec_r == (0, 0) and dec_b == (0, 0)
    for b in (emu, emu_r, emu_b):
        check_pileup(fx, b)
        check_trace(fx, b)


def test_two_or_more_unparseable_records_decline_without_touching_their_bytes(tmp_path):
    """records the kernels cannot take apart are made inert (FE_R_FAILED: zero offsets and lengths) and the call declines
    after the record stage -- the filters, the key table and the byte-for-byte key compare never follow a stale offset.  Two and
    more bad records in one input (blank inside a line, CR, a mapped BAM record without CIGAR / SEQ), between good ones; UNMAPPED
    BAM records without CIGAR and SEQ are routine and do not decline: the filters drop them on their flag."""
    from hisatgenotype_amd import bamio
    fx = gu.load("hla_small_pair")
    loc = fx["_locus"]
    pl = hl.PackedLocus.from_synth(loc)
    lines = [l for l in fx["sam"].split("\n") if l]
    bad = list(lines)
    for k in (3, 4, 9, 40):
        f = bad[k].split("\t")
        f[0] = f[0] + " x" if k != 9 else f[0] + "\r"
        bad[k] = "\t".join(f)
    sam_bad = "\n".join(bad) + "\n"
    try:
        host = pl.parse_sam(sam_bad, simulation=True)
    except capi.HgxError:
        host = None
    if host is None:
        with pytest.raises(capi.HgxError):
            emulated_records(pl, sam=sam_bad, simulation=True)
    else:
        emu, dec = emulated_records(pl, sam=sam_bad, simulation=True)
        assert dec[0] != 0                                           # the record route declined ...
        same_batch(host, emu, len(loc.backbone))                     # ... and the host stages finished the job
    # unmapped mates (CIGAR '*', SEQ '*', no tags) among the records, as SAM text and as BAM: taken, and they change nothing; the
    # same records with the unmapped flag cleared: whatever the host front end does with them (it follows the reference)
    host = pl.parse_sam(fx["sam"], simulation=True)
    for flag_or in (0x4, 0):
        out = list(lines)
        for k in (2, 3, 20):
            f = lines[k].split("\t")
            out.insert(k, "\t".join([f[0], str((int(f[1]) & ~0x4) | flag_or), f[2], f[3], "0", "*", "=", f[7], "0", "*", "*"]))
        sam2 = "\n".join(out) + "\n"
        p2 = str(tmp_path / ("m%d.bam" % flag_or))
        bamio.write_bam(p2, sam2, [(loc.ref_allele, len(loc.backbone))])
        for kind in ("sam", "bam"):
            run_host = (lambda: pl.parse_sam(sam2, simulation=True)) if kind == "sam" else \
                (lambda: pl.parse_alignment_file(p2, regions=[loc.ref_allele], simulation=True))
            run_emu = (lambda: emulated_records(pl, sam=sam2, simulation=True)) if kind == "sam" else \
                (lambda: emulated_records(pl, path=p2, regions=loc.ref_allele, simulation=True))
            try:
                want = run_host()
            except capi.HgxError:
                assert flag_or == 0
                with pytest.raises(capi.HgxError):
                    run_emu()
                continue
            emu, dec = run_emu()
            same_batch(want, emu, len(loc.backbone))
            if flag_or:
                assert dec == (0, 0), (kind, dec)                    # unmapped records do not decline the record route ...
                same_batch(host, emu, len(loc.backbone))             # ... and change nothing


@pytest.mark.parametrize("name", gu.ALL + gu.LEAN)
def test_emulated_record_route_equals_the_host_front_end_on_every_fixture(name, tmp_path):
    """Fields, record filters and key grouping emulated too -- from SAM text, a SAM file, a name-grouped BAM and a coordinate-sorted
    BAM with regions (the BAM records are read in binary: CIGAR words, packed SEQ)."""
    from hisatgenotype_amd import bamio
    fx = gu.load(name)
    o = fx["options"]
    loc = fx["_locus"]
    pl = hl.PackedLocus.from_synth(loc)
    kw = dict(num_editdist=o["num_editdist"], error_correction=o["error_correction"], allow_discordant=o["allow_discordant"],
              simulation=o["simulation"])
    host = pl.parse_sam(fx["sam"], **kw)
    want = (0, 0)                                  # (codis_d18s51 too: get_pair_interdist and choose_pairs are device stages since round 6)
    emu, dec = emulated_records(pl, sam=fx["sam"], **kw)
    assert dec == want, dec
    same_batch(host, emu, len(loc.backbone))
    p_sam, p_bam, p_sorted = str(tmp_path / "r.sam"), str(tmp_path / "r.bam"), str(tmp_path / "s.bam")
    open(p_sam, "w").write(fx["sam"])
    bamio.write_bam_native(p_bam, fx["sam"].encode(), [(loc.ref_allele, len(loc.backbone))])
    bamio.write_bam_native(p_sorted, fx["sam"].encode(), [(loc.ref_allele, len(loc.backbone))], sort_by_coordinate=True)
    for path in (p_sam, p_bam, p_sorted):
        emu, dec = emulated_records(pl, path=path, regions=loc.ref_allele, **kw)
        assert dec == want, (path, dec)
        same_batch(host, emu, len(loc.backbone))


def test_emulated_choose_pairs_on_the_last_pair_of_a_d18s51_stream(tmp_path):
    """get_pair_interdist (typing_common.py:1187-1265: the median inner distance of the sample's unique concordant pairs, as
    k_fe_interdist_* count it) and choose_pairs on the stream's last pair (fe_choose_pairs in k_fe_pair_choose / k_fe_pair_emit) == the
    host front end, by the key route, the record route and from a BAM; in a good share of the cases the choice CHANGES the batch."""
    from hisatgenotype_amd import bamio
    n_all = n_changed = 0
    pls = {}
    for d18, plain, text in d18s51_cases():
        if id(d18) not in pls:
            pls[id(d18)] = (hl.PackedLocus.from_synth(d18), hl.PackedLocus.from_synth(plain))
        pl, pl0 = pls[id(d18)]
        host = pl.parse_sam(text)
        n_all += 1
        n_changed += host.pair_ref.tobytes() != pl0.parse_sam(text).pair_ref.tobytes()
        emu, dec = emulated(pl, text)
        assert dec == 0
        same_batch(host, emu, len(d18.backbone))
        emu, dec = emulated_records(pl, sam=text)
        assert dec == (0, 0), dec
        same_batch(host, emu, len(d18.backbone))
        if n_all % 8 == 0:
            # (a BAM's records come back in name order -- stable, so the moved pair is no longer last: another last pair, the same rule)
            p_bam = str(tmp_path / "d.bam")
            bamio.write_bam_native(p_bam, text.encode(), [(d18.ref_allele, len(d18.backbone))], sort_by_coordinate=True)
            host_b = pl.parse_alignment_file(p_bam, d18.ref_allele)
            emu, dec = emulated_records(pl, path=p_bam, regions=d18.ref_allele)
            assert dec == (0, 0), dec
            same_batch(host_b, emu, len(d18.backbone))
    assert n_all >= 150 and n_changed >= n_all // 5, (n_all, n_changed)
    # records that do not count: a YT tag that is not CP, NH > 1, unaligned, a later (short) YT tag that overrides an earlier one
    d18, plain, text = next(iter(d18s51_cases(1)))
    pl = pls[id(d18)][0] if id(d18) in pls else hl.PackedLocus.from_synth(d18)
    lines = text.split("\n")[:-1]
    for k in range(0, len(lines), 5):
        lines[k] = lines[k].replace("YT:Z:CP", "YT:Z:DP")
    for k in range(1, len(lines), 11):
        lines[k] = lines[k] + "\tYT"
    for k in range(2, len(lines), 13):
        lines[k] = lines[k] + "\tYT:Z:CP"
    text2 = "\n".join(lines) + "\n"
    host = pl.parse_sam(text2)
    for run in (lambda: emulated(pl, text2), lambda: emulated_records(pl, sam=text2)):
        emu, dec = run()
        assert dec in (0, (0, 0)), dec
        same_batch(host, emu, len(d18.backbone))


def test_key_find_on_variant_ids_equals_the_text_search():
    """identify_ambigious_diffs asks `key.find(cur_join) != -1` (typing_common.py:1744, 1868) of every candidate alternative; the kernels
    ask it of the variant IDS (fe_key_contains: equal ids, the needle's LAST id a decimal prefix of the key's -- "hv4" is found in
    "hv40", the reference's quirk).  Both forms on made-up keys: ids whose names are prefixes of one another, needles that end in
    such an id, start in the middle of the key, hold a novel id, are longer than the key."""
    rng = random.Random(5)
    names = ["hv%d" % k for k in (1, 10, 100, 11, 12, 2, 20, 21, 3, 30, 4, 40, 400, 41, 5, 7, 77, 770, 8, 9)]
    text = "\n".join(names).encode()
    n_true = n_prefix_only = 0
    out = (C.c_int32 * 2)()
    for _ in range(20000):
        nk = rng.randint(0, 7)
        key = [rng.randrange(len(names)) for _ in range(nk)]
        n = rng.randint(1, 4)
        if nk >= n and rng.random() < 0.6:                      # mostly: a window of the key, its last id cut to a prefix now and then
            p = rng.randint(0, nk - n)
            cur = key[p:p + n]
            if rng.random() < 0.4:
                shorter = [i for i, nm in enumerate(names) if names[cur[-1]].startswith(nm)]
                cur[-1] = rng.choice(shorter)
            if rng.random() < 0.1:
                cur[rng.randrange(n)] = rng.choice([-1, len(names) + 3])            # a novel id
        else:
            cur = [rng.randrange(len(names)) for _ in range(n)]
        ka, ca = (C.c_int32 * max(nk, 1))(*key), (C.c_int32 * n)(*cur)
        capi.check(capi.lib().hgx_lab_key_contains(text, ka, C.c_int32(nk), ca, C.c_int32(n), out))
        assert out[0] == out[1], (key, cur, out[0], out[1])
        n_true += out[0]
        want_py = "-".join(names[c] if 0 <= c < len(names) else "nv" for c in cur) in "529-" + "".join(names[k] + "-" for k in key) + "606"
        assert bool(out[1]) == want_py
        if out[0] and nk >= n and not any(key[p:p + n] == cur for p in range(nk - n + 1)):
            n_prefix_only += 1
    assert n_true > 5000 and n_prefix_only > 300, (n_true, n_prefix_only)


def test_emulated_device_stages_on_fuzz_cases():
    """The cases of tools/fuzz_parity.py (HLA-like loci with deletions / insertions / unlinked variants, STR loci, sequencing
    errors, soft clips, novel indels, multi-hit and duplicate records, single-end samples) plus deeper samples of the fast generator."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_parity
    n_dev = n_all = n_rec = 0
    why, why_rec = {}, {}
    for k in range(int(os.environ.get("HGX_FRONT_FUZZ", "120"))):
        loc, sam, single = fuzz_parity.make_case(770000, k, 1 + k % 3)
        pl = hl.PackedLocus.from_synth(loc)
        for ec in (True, False):
            try:
                host = pl.parse_sam(sam, error_correction=ec, allow_discordant=single)
            except capi.HgxError:
                with pytest.raises(capi.HgxError):
                    emulated(pl, sam, error_correction=ec, allow_discordant=single)
                continue
            emu, declined = emulated(pl, sam, error_correction=ec, allow_discordant=single)
            n_all += 1
            n_dev += declined == 0
            why[declined] = why.get(declined, 0) + 1
            same_batch(host, emu, len(loc.backbone))
            emu, dec = emulated_records(pl, sam=sam, error_correction=ec, allow_discordant=single)
            n_rec += dec[0] == 0
            why_rec[dec[0]] = why_rec.get(dec[0], 0) + 1
            same_batch(host, emu, len(loc.backbone))
        pl.close()
    rng = random.Random(20261003)
    for _ in range(6):
        loc = synth.make_hla_like_locus(n_alleles=rng.choice([40, 400]), n_vars=rng.choice([150, 900]), seed=rng.randrange(1 << 30),
                                        deletion_frac=rng.choice([0.07, 0.3]))
        sam = synth.simulate_sam_fast(loc, synth.pick_sample(loc, rng.randrange(1 << 30)), rng.choice([1200, 15000]),
                                      err_rate=rng.choice([0.002, 0.01]), seed=rng.randrange(1 << 30))
        pl = hl.PackedLocus.from_synth(loc)
        host = pl.parse_sam(sam)
        emu, declined = emulated(pl, sam)
        n_all += 1
        n_dev += declined == 0
        why[declined] = why.get(declined, 0) + 1
        same_batch(host, emu, len(loc.backbone))
    print("device stages took %d of %d inputs; decline codes %s; record route %d, decline codes %s" % (n_dev, n_all, why, n_rec, why_rec))
    assert n_dev >= 0.8 * n_all, (n_dev, n_all, why)


# ---- MANY tasks of one locus in one pass (hgx_many_create_files / _sams): emulated stages against hgx_batch_merge of the host's batches ----
def many_emulated(pl, sams=None, paths=None, regions=None, **kw):
    n = len(sams if sams is not None else paths)
    o = capi.ParseOpts(kw.get("num_editdist", 2), int(kw.get("error_correction", True)), int(kw.get("allow_discordant", False)),
                       int(kw.get("simulation", False)), 0, 0, int(pl.base_fname == "codis" and pl.gene == "D18S51"), 0)
    h, dec = C.c_void_p(), C.c_int32(0)
    base, reads, pieces, refs = (C.c_int32 * (n + 1))(), (C.c_int32 * n)(), (C.c_int32 * n)(), (C.c_int64 * n)()
    p_arr = r_arr = s_arr = n_arr = None
    if paths is not None:
        p_arr = (C.c_char_p * n)(*[p.encode() for p in paths])
        if regions is not None:
            r_arr = (C.c_char_p * n)(*[r.encode() if r else None for r in regions])
    else:
        keep = [s if isinstance(s, bytes) else s.encode() for s in sams]
        s_arr = (C.c_char_p * n)(*keep)
        n_arr = (C.c_size_t * n)(*[len(s) for s in keep])
    capi.check(capi.lib().hgx_lab_many_emulated(C.byref(h), pl.h, p_arr, r_arr, s_arr, n_arr, C.c_int32(n), C.byref(o), base, reads, pieces,
                                                refs, C.byref(dec)))
    if dec.value:
        return None, dec.value, None
    return hl.Batch(h), 0, (list(base), list(reads), list(pieces), list(refs))


def host_merged(pl, sams, **kw):
    bs = [pl.parse_sam(s, **kw) for s in sams]
    n = len(bs)
    arr = (C.c_void_p * n)(*[b.h for b in bs])
    base = (C.c_int32 * (n + 1))()
    h = C.c_void_p()
    capi.check(capi.lib().hgx_lab_batch_merge(C.byref(h), arr, C.c_int32(n), base))
    return hl.Batch(h), (list(base), [b.n_reads for b in bs], [b.n_pieces for b in bs], [b.n_refs for b in bs]), bs


def same_merged(a, b):
    assert (a.n_reads, a.n_pairs, a.n_pieces, a.n_refs, a.n_mask_u32) == (b.n_reads, b.n_pairs, b.n_pieces, b.n_refs, b.n_mask_u32)
    assert a.pieces.tobytes() == b.pieces.tobytes()
    assert a.masks.tobytes() == b.masks.tobytes()
    assert a.pair_off.tobytes() == b.pair_off.tobytes()
    assert a.pair_ref.tobytes() == b.pair_ref.tobytes()


def _samples(loc, n, pairs, seed, **kw):
    rng = random.Random(seed)
    return [synth.simulate_sam_fast(loc, synth.pick_sample(loc, rng.randrange(1 << 30)), pairs + 37 * t, err_rate=kw.get("err_rate", 0.004),
                                    seed=rng.randrange(1 << 30)) for t in range(n)]


def test_many_tasks_in_one_emulated_pass_equal_the_merge_of_the_hosts_batches(tmp_path):
    """Samples of one locus with DIFFERENT alleles (so the same read text decodes differently under each sample's own pileup), an
    empty task among them, the same sample twice (equal records in two tasks stay two keys); texts in memory, SAM files and BAMs."""
    from hisatgenotype_amd import bamio
    loc = synth.make_hla_like_locus(n_alleles=120, n_vars=400, seed=5, deletion_frac=0.15)
    pl = hl.PackedLocus.from_synth(loc)
    sams = _samples(loc, 5, 700, 11, err_rate=0.01)
    sams = sams[:2] + [""] + sams[2:] + [sams[0]]
    want, wt, _ = host_merged(pl, sams)
    got, dec, gt = many_emulated(pl, sams=sams)
    assert dec == 0, dec
    same_merged(want, got)
    assert gt == wt, (gt, wt)
    p_sam, p_bam = [], []
    for t, s in enumerate(sams):
        p_sam.append(str(tmp_path / ("t%d.sam" % t)))
        open(p_sam[-1], "w").write(s)
        p_bam.append(str(tmp_path / ("t%d.bam" % t)))
        bamio.write_bam_native(p_bam[-1], s.encode(), [(loc.ref_allele, len(loc.backbone))], sort_by_coordinate=(t % 2 == 1))
    for paths in (p_sam, p_bam):
        got, dec, gt = many_emulated(pl, paths=paths, regions=[loc.ref_allele] * len(paths))
        assert dec == 0, dec
        same_merged(want, got)
        assert gt == wt
    # SAM text and BAM records in one batch: declined (the host goes task by task)
    got, dec, _ = many_emulated(pl, paths=[p_sam[0], p_bam[1]])
    assert got is None and dec == 1


def test_many_task_emulation_on_fixtures_and_fuzz_cases():
    """Every golden fixture three times over as the tasks of one batch, and fuzz loci with several samples each."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_parity
    for name in gu.ALL + gu.LEAN:
        if name == "codis_d18s51":
            continue
        fx = gu.load(name)
        o = fx["options"]
        pl = hl.PackedLocus.from_synth(fx["_locus"])
        kw = dict(num_editdist=o["num_editdist"], error_correction=o["error_correction"], allow_discordant=o["allow_discordant"],
                  simulation=o["simulation"])
        lines = fx["sam"].splitlines(keepends=True)
        half = "".join(lines[: len(lines) // 2])
        # (a read's records stay together: cut at a read-id boundary)
        cut = len(lines) // 2
        while 0 < cut < len(lines) and lines[cut].split("\t")[0] == lines[cut - 1].split("\t")[0]:
            cut += 1
        half = "".join(lines[:cut])
        sams = [fx["sam"], half, fx["sam"]]
        want, wt, _ = host_merged(pl, sams, **kw)
        got, dec, gt = many_emulated(pl, sams=sams, **kw)
        assert dec == 0, (name, dec)
        same_merged(want, got)
        assert gt == wt, name
    n_dev = n_all = 0
    for k in range(int(os.environ.get("HGX_FRONT_FUZZ", "120")) // 3):
        rng = random.Random(880000 + k)
        loc, sam0, single = fuzz_parity.make_case(880000, k, 1 + k % 3)
        sams = [sam0]
        for j in range(1 + k % 3):
            _, s, _ = fuzz_parity.make_case(880000, k, 1 + (k + j + 1) % 3)
            sams.append(s)
        pl = hl.PackedLocus.from_synth(loc)
        try:
            want, wt, _ = host_merged(pl, sams, allow_discordant=single)
        except capi.HgxError:
            continue
        got, dec, gt = many_emulated(pl, sams=sams, allow_discordant=single)
        n_all += 1
        if dec:
            continue
        n_dev += 1
        same_merged(want, got)
        assert gt == wt
        pl.close()
    print("many-task emulation took %d of %d fuzz batches" % (n_dev, n_all))
    assert n_dev >= 0.7 * n_all, (n_dev, n_all)
