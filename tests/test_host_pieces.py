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
