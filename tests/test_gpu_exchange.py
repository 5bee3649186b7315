"""The class-table exchange of a sharded locus (8e) with the data of a world of N ranks on ONE GPU: every shard's class tables are
made as dist.type_shard makes them, packed by hgx_classes_pack_rows (the send half of hgx_classes_allgather), laid side by side in
rank order by the test -- exactly what ncclAllGather leaves in the receive buffer -- and unpacked + merged by
hgx_classes_merge_gathered (the receive half).  The merged set must be the class dict of the unsharded sample: same rows, same
counts, same first-seen order (typing_core.py:1229-1234), at both levels; shards without pairs and tables of very different sizes
included.  hgx_classes_allgather itself is those two halves around one collective (hgx_rccl.hip)."""
import ctypes as C

import numpy as np
import pytest

from hisatgenotype_amd import capi, dist as hdist, engine, locus as hl, synth

pytestmark = pytest.mark.gpu


def _pileup_sum(pl, shards):
    """The pileup counts of the whole sample as the shards' exchange leaves them (get_mpileup covers the whole alignment,
    typing_common.py:1059-1184; error correction needs all of it): every shard's own counts, summed."""
    parts = []
    for s in shards:
        if s:
            pl.parse_sam(s, pileup_exchange=lambda c: parts.append(c.copy())).close()
    return np.sum(np.stack(parts), axis=0, dtype=np.uint32) if parts else None


def _level_classes(pl, sam_bytes, pileup=None):
    """(exon classes or None, gene classes or None) of one shard -- dist.type_shard's first half; `pileup`: the summed counts the
    shard's front end receives in place of its own (dist.parse_shard's exchange)."""
    if not sam_bytes:
        return None, None

    def give(c):
        c[...] = pileup
    batch = pl.parse_sam(sam_bytes, pileup_exchange=give if pileup is not None else None)
    if batch.n_pairs == 0:
        return None, None
    db = engine.DeviceBatch(batch)
    hla = pl.base_fname == "hla"
    bufs = engine.ScoreBuffers(pl, db, exon=hla)
    engine.piece_compat(pl, db, bufs)
    engine.pair_classes(pl, db, bufs, exon=False)
    ex = engine.Classes.of_level(pl, db, bufs, 0) if hla else None
    ge = engine.Classes.dedup(bufs.gene_bits, db.n_pairs, pl.a_pad, hashes=bufs.gene_hash)
    capi.sync()
    db.close()
    return ex, ge


def _exchange_on_one_gpu(tables, a_pad):
    """tables: per rank a Classes or None.  Pack every rank's rows, assemble the receive buffer in rank order, merge."""
    L = capi.lib()
    w64 = a_pad // 64
    world = len(tables)
    sizes = np.array([0 if t is None else t.n_classes for t in tables], np.int32)
    cap = max(int(sizes.max()), 1)
    recv_host = np.zeros((world, cap, w64 + 1), np.uint64)
    for r, t in enumerate(tables):
        send = capi.DevArray((cap, w64 + 1), np.uint64)
        capi.check(L.hgx_classes_pack_rows(t.h if t is not None else None, C.c_int32(a_pad), C.c_int32(cap), capi.ptr(send), None))
        recv_host[r] = send.to_host()
        assert not recv_host[r, sizes[r]:].any()                          # the padding rows travel as zeros
    recv = capi.DevArray.from_host(np.ascontiguousarray(recv_host))
    h = C.c_void_p()
    capi.check(L.hgx_classes_merge_gathered(C.byref(h), capi.ptr(recv), capi.ptr(sizes), C.c_int32(world), C.c_int32(cap), C.c_int32(a_pad), None))
    return engine.Classes(h), recv_host, sizes


CASES = [
    ("hla", dict(n_alleles=1500, n_vars=1200, seed=41), 6000, 2),
    ("hla", dict(n_alleles=1500, n_vars=1200, seed=41), 6000, 3),
    ("hla", dict(n_alleles=700, n_vars=600, seed=31, sibling_frac=0.4), 2500, 5),
    ("str", None, 900, 4),
]


@pytest.mark.parametrize("kind,kw,n_pairs,world", CASES)
@pytest.mark.parametrize("split", ["even", "first_rank_empty", "one_rank_has_everything", "ragged"])
def test_gathered_class_tables_merge_to_the_unsharded_set(kind, kw, n_pairs, world, split):
    capi.set_device(0)
    if kind == "hla":
        loc = synth.make_hla_like_locus(**kw)
        sam = synth.simulate_sam_fast(loc, synth.pick_sample(loc, 9), n_pairs, err_rate=0.004, seed=17)
    else:
        loc = synth.make_str_like_locus(gene="TH01", unit="AATG", max_repeats=12, min_repeats=4, seed=5)
        names = [a for a in loc.allele_names if "BACKBONE" not in a]
        sam = synth.simulate_sam_fast(loc, [names[2], names[-3]], n_pairs, read_len=100, frag_len=(200, 300), err_rate=0.002, seed=8)
    pl = hl.PackedLocus.from_synth(loc)
    if split == "even":
        shards = hdist.split_name_grouped(sam, world)
    elif split == "first_rank_empty":
        shards = [b""] + hdist.split_name_grouped(sam, world - 1)
    elif split == "one_rank_has_everything":
        shards = [b""] * (world - 1) + [sam.encode()]
    else:                                   # very different table sizes: the first rank a sliver, the rest in growing pieces
        parts = hdist.split_name_grouped(sam, 16)
        cuts = sorted(set([1] + [1 + (15 * (k + 1)) // (world - 1) for k in range(world - 1)]))
        cuts = [0] + cuts
        shards = [b"".join(parts[cuts[i]:cuts[i + 1]]) for i in range(len(cuts) - 1)]
        shards += [b""] * (world - len(shards))
    assert len(shards) == world and b"".join(shards) == sam.encode()
    whole = _level_classes(pl, sam.encode())
    total = _pileup_sum(pl, shards)
    per_rank = [_level_classes(pl, s, total) for s in shards]
    for level, name in ((0, "exon"), (1, "gene")):
        if whole[level] is None:
            assert all(p[level] is None for p in per_rank)
            continue
        merged, recv_host, sizes = _exchange_on_one_gpu([p[level] for p in per_rank], pl.a_pad)
        bits, cnt, _ = merged.to_host()
        wb, wc, _ = whole[level].to_host()
        assert bits.shape == wb.shape, (name, bits.shape, wb.shape)
        assert np.array_equal(bits, wb), name                    # same rows in the same (first-seen) order
        assert np.array_equal(cnt, wc), name
        assert int(cnt.sum()) == sum(int(recv_host[r, :sizes[r], -1].sum()) for r in range(world))
        # the host form of the same merge (dist.TorchComm.all_gather_tables + merge_class_tables) agrees
        tabs = [(np.ascontiguousarray(recv_host[r, :sizes[r], :-1]), recv_host[r, :sizes[r], -1].astype(np.int64)) for r in range(world)]
        hm = hdist.merge_class_tables(tabs, pl.a_pad)
        hb, hc, _ = hm.to_host()
        assert np.array_equal(hb, bits) and np.array_equal(hc, cnt)
        hm.close()
        merged.close()
    pl.close()


def test_merge_gathered_rejects_sizes_beyond_cap_and_counts_nothing_for_an_all_empty_world():
    capi.set_device(0)
    L = capi.lib()
    a_pad = 512
    recv = capi.DevArray.from_host(np.zeros((3, 4, a_pad // 64 + 1), np.uint64))
    h = C.c_void_p()
    sizes = np.array([1, 5, 0], np.int32)
    assert L.hgx_classes_merge_gathered(C.byref(h), capi.ptr(recv), capi.ptr(sizes), C.c_int32(3), C.c_int32(4), C.c_int32(a_pad), None) == -1
    sizes = np.zeros(3, np.int32)
    capi.check(L.hgx_classes_merge_gathered(C.byref(h), capi.ptr(recv), capi.ptr(sizes), C.c_int32(3), C.c_int32(4), C.c_int32(a_pad), None))
    cl = engine.Classes(h)
    assert cl.n_classes == 0
    cl.close()
    n, s, r = C.c_uint64(), C.c_uint64(), C.c_uint64()
    capi.check(L.hgx_rccl_stats(C.byref(n), C.byref(s), C.byref(r), C.c_int32(1)))
    assert (n.value, s.value, r.value) == (0, 0, 0)             # no collective was issued by the halves
