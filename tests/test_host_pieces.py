"""Host logic (no GPU): locus packing and the haplotype -> piece-mask reduction of libhgx against
the C oracle's add_count restatement, evaluated with numpy on the packed bit matrix."""
import numpy as np
import pytest

import golden_util as gu
import tables
from hisatgenotype_amd import locus as hl


def _numpy_classes(pl, batch):
    """Evaluate the mask rule + arg-max on the CPU with numpy (checker for the host-built masks only)."""
    t = pl.tables()
    bits = t["link_bits"]
    A, w64 = pl.n_alleles, pl.w64
    compat = np.zeros((batch.n_pieces, pl.a_pad), dtype=bool)
    for i, pc in enumerate(batch.pieces):
        lo, nw, off = int(pc["lo_word"]), int(pc["n_words"]), int(pc["mask_off"])
        m = batch.masks[off:off + 2 * nw].reshape(nw, 2)
        ok = np.ones(pl.a_pad, dtype=bool)
        for k in range(nw):
            ok &= (bits[lo + k] & m[k, 0]) == m[k, 1]
        compat[i] = ok
    def unpack(mask):
        return np.unpackbits(mask.view(np.uint8), bitorder="little").astype(bool)
    lvl_mask = [unpack(t["exon_mask"]), unpack(t["gene_mask"])]
    out = [np.zeros((batch.n_pairs, w64), np.uint64), np.zeros((batch.n_pairs, w64), np.uint64)]
    for p in range(batch.n_pairs):
        refs = batch.pair_ref[batch.pair_off[p]:batch.pair_off[p + 1]]
        for lv in (0, 1):
            cnt = np.zeros(pl.a_pad, np.int32)
            for r in refs:
                if (int(r) >> 31) == lv:
                    cnt += compat[int(r) & 0x7fffffff]
            m = lvl_mask[lv]
            if not m.any():
                continue
            mx = cnt[m].max()
            cls = m & (cnt == mx)
            out[lv][p] = np.packbits(cls, bitorder="little").view(np.uint64)
    return out


@pytest.mark.parametrize("name", gu.SMALL)
def test_piece_masks_match_oracle(orc, name):
    fx = gu.load(name)
    loc = fx["_locus"]
    t = tables.oracle_tables(loc)
    pl = hl.PackedLocus.from_synth(loc)
    assert pl.names == t["names"]
    # representatives (get_rep_alleles) agree with the Python oracle
    rep = pl.tables()["rep_of"]
    for a, n in enumerate(pl.names):
        exp = t["reps"].get(n)
        assert (exp is None and rep[a] == -1) or (exp is not None and pl.names[rep[a]] == exp)
    arrs = tables.pieces_from_pairs(fx["pairs"], t["var_index"])
    batch = pl.batch_from_haplotypes(*arrs)
    assert batch.n_pairs == len(fx["pairs"])
    L = orc.make_locus(t)
    eb, gb, gc, fp = orc.score_pairs(L, t["exon_keys"], t["gene_keys"], *arrs)
    got_e, got_g = _numpy_classes(pl, batch)
    w = (t["n_alleles"] + 63) // 64
    assert np.array_equal(got_g[:, :w], gb)
    assert not got_g[:, w:].any()
    if loc.base_fname == "hla":
        assert np.array_equal(got_e[:, :w], eb)


def test_piece_dedup_is_effective():
    fx = gu.load("hla_mid_real")
    t = tables.oracle_tables(fx["_locus"])
    pl = hl.PackedLocus.from_synth(fx["_locus"])
    batch = pl.batch_from_haplotypes(*tables.pieces_from_pairs(fx["pairs"], t["var_index"]))
    assert batch.n_refs > batch.n_pieces   # identical add_count arguments are stored once
    assert batch.pair_off[-1] == batch.n_refs


def test_packed_index_cache_round_trip(tmp_path):
    """PackedLocus.save_cache / load_cache (SURVEY 8f-1 "packed binary cache"): identical derived tables, alternatives and
    front-end output, for an HLA-like locus with insertions and for a CODIS-like one."""
    from hisatgenotype_amd import synth, locus as hl
    for k, loc in enumerate((synth.make_hla_like_locus(n_alleles=200, n_vars=180, seed=8, insertion_frac=0.05),
                             synth.make_str_like_locus(seed=3))):
        pl = hl.PackedLocus.from_synth(loc)
        path = str(tmp_path / ("locus%d.npz" % k))
        pl.save_cache(path)
        p2 = hl.PackedLocus.load_cache(path)
        assert (p2.gene, p2.base_fname, p2.names, p2.var_ids, p2.ref_seq) == (pl.gene, pl.base_fname, pl.names, pl.var_ids, pl.ref_seq)
        t1, t2 = pl.tables(), p2.tables()
        assert all(np.array_equal(t1[x], t2[x]) for x in t1)
        assert np.array_equal(pl.allele_len, p2.allele_len) and np.array_equal(pl.name_rank, p2.name_rank)
        assert pl.alternatives_text() == p2.alternatives_text()
        sample = synth.pick_sample(loc, 5)
        if k == 0:
            sam = synth.simulate_sam_fast(loc, sample, 300, err_rate=0.002, seed=9)
        else:
            sam = synth.sam_text(loc, synth.simulate_pairs(loc, sample, 150, read_len=100, frag_len=(250, 250), seed=9))
        b1, b2 = pl.parse_sam(sam), p2.parse_sam(sam)
        assert (b1.n_reads, b1.n_pairs, b1.n_pieces) == (b2.n_reads, b2.n_pairs, b2.n_pieces)
        assert np.array_equal(b1.pieces, b2.pieces) and np.array_equal(b1.masks, b2.masks)
        assert np.array_equal(b1.pair_off, b2.pair_off) and np.array_equal(b1.pair_ref, b2.pair_ref)


def test_front_end_batch_is_independent_of_the_worker_count(tmp_path):
    """The piece batch of a read set -- distinct pieces in their canonical order, masks, per-pair refs -- is the same whatever
    the number of front-end workers (chunks are merged through hash partitions; the table is ordered by content), from SAM
    text, from a SAM file and from a BAM file; and a name-grouped file is recognised as such (no sort) without changing it."""
    import numpy as np
    from hisatgenotype_amd import bamio, locus as hl, synth
    loc = synth.make_hla_like_locus(n_alleles=300, n_vars=500, seed=77)
    sample = synth.pick_sample(loc, 5)
    sam = synth.simulate_sam_fast(loc, sample, 15000, err_rate=0.003, seed=8)           # 30 000 records: the threaded paths
    pl = hl.PackedLocus.from_synth(loc)
    ref = pl.parse_sam(sam, n_threads=1)

    def same(b):
        assert (b.n_reads, b.n_pairs, b.n_pieces, b.n_refs) == (ref.n_reads, ref.n_pairs, ref.n_pieces, ref.n_refs)
        assert np.array_equal(b.pieces, ref.pieces) and np.array_equal(b.masks, ref.masks)
        assert np.array_equal(b.pair_off, ref.pair_off) and np.array_equal(b.pair_ref, ref.pair_ref)

    for nt in (2, 3, 8):
        same(pl.parse_sam(sam, n_threads=nt))
    path = str(tmp_path / "s.sam")
    open(path, "w").write("@SQ\tSN:%s\tLN:%d\n" % (loc.ref_allele, len(loc.backbone)) + sam)
    for nt in (1, 8):
        same(pl.parse_alignment_file(path, [loc.ref_allele], n_threads=nt))
        same(pl.parse_alignment_file(path, None, n_threads=nt))
    lines = [l for l in sam.split("\n") if l]
    lines.sort(key=lambda l: int(l.split("\t")[3]))                                    # coordinate order: the reader must sort
    bam = str(tmp_path / "s.bam")
    bamio.write_bam(bam, "\n".join(lines[:6000]) + "\n", [(loc.ref_allele, len(loc.backbone))])
    a = pl.parse_alignment_file(bam, [loc.ref_allele], n_threads=1)
    b = pl.parse_alignment_file(bam, [loc.ref_allele], n_threads=8)
    assert np.array_equal(a.pieces, b.pieces) and np.array_equal(a.pair_ref, b.pair_ref) and a.n_reads == b.n_reads > 0
