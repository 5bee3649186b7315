"""Host logic of round 6 that needs no GPU: the in-process locus cache of typing() (identity key, content-digest fallback, in-place edits,
eviction), the per-process memo of the index files, and the native parser of single_abundance's class keys (hgx_keyset_*) against the
Python statement of the same walk (typing_common.py:1282-1305: names in order of first appearance, one membership row per key)."""
import copy
import os

import numpy as np
import pytest

import golden_util as gu
import hisatgenotype_amd  # noqa: F401
import sys
from hisatgenotype_amd import capi, indexio, locus as hl, synth

T = sys.modules["hisatgenotype_amd.typing"]


def _dicts(loc):
    d = loc.reference_dicts()
    return (loc.gene, loc.base_fname, d["refGenes"], d["Genes"], d["Gene_names"], d["Gene_lengths"], d["refGene_loci"], d["Vars"], d["Var_list"], d["Links"])


def test_locus_cache_identity_content_and_in_place_edits():
    cache = hl.LocusCache(limit=2)
    loc = gu.load("hla_mid_real")["_locus"]
    args = _dicts(loc)
    a = cache.get(*args)
    assert (cache.hits_identity, cache.hits_content, cache.misses) == (0, 0, 1) and getattr(a, "cached", False)
    assert cache.get(*args) is a and cache.hits_identity == 1                       # the same dict objects: found by identity
    twin = copy.deepcopy(args)
    assert cache.get(*twin) is a and cache.hits_content == 1                        # equal content in other objects: found by digest ...
    assert cache.get(*twin) is a and cache.hits_identity == 2                       # ... and re-keyed to them
    # an index edited IN PLACE between calls (same objects, same container sizes, one more carrier of one variant) is another locus
    links = twin[9]
    vid = next(iter(links))
    newcomer = [n for n in twin[4][loc.gene] if "BACKBONE" not in n and n not in links[vid]][0]
    links[vid].append(newcomer)
    b = cache.get(*twin)
    assert b is not a and cache.misses == 2
    ta, tb = a.tables(), b.tables()
    assert not np.array_equal(ta["link_bits"], tb["link_bits"])                      # the packed link matrix follows the edit
    fresh = hl.PackedLocus.from_reference_dicts(*twin)
    assert np.array_equal(fresh.tables()["link_bits"], tb["link_bits"])
    # a changed variant (same sizes everywhere) changes the digest too
    third = copy.deepcopy(args)
    v0 = third[8][loc.gene][0][1]
    t_, p_, d_ = third[7][loc.gene][v0]
    if t_ == "single":
        third[7][loc.gene][v0] = [t_, p_, "A" if d_ != "A" else "C"]
    else:
        third[7][loc.gene][v0] = [t_, p_, str(int(d_) + 1) if t_ == "deletion" else d_ + "A"]
    c = cache.get(*third)
    assert c is not a and c is not b and cache.misses == 3
    # limit 2: the least recently used entry left the cache (and is no longer marked cached), the others are still there
    assert len(cache._by_content) == 2 and not getattr(a, "cached", True)
    assert cache.get(*third) is c
    cache.clear()
    assert len(cache._by_content) == 0 and len(cache._by_id) == 0
    for x in (fresh,):
        x.close()


def test_index_memo_returns_the_same_dicts_until_a_file_changes(tmp_path):
    loc = synth.make_hla_like_locus(gene="A", n_alleles=60, n_vars=150, seed=3)
    ix_dir = str(tmp_path / "ix")
    synth.write_index([loc], ix_dir, "hla")
    a = indexio.load_index_memo(ix_dir, "hla")
    assert indexio.load_index_memo(ix_dir, "hla") is a                              # the same objects: typing()'s cache finds its loci by identity
    pl = indexio.packed_locus(ix_dir, "hla", "A", a)                                # (writes hla.A.hgx.npz next to the index: not an index file)
    pl.close()
    assert indexio.load_index_memo(ix_dir, "hla") is a
    p = os.path.join(ix_dir, "hla.link")
    text = open(p).read()
    with open(p, "w") as f:
        f.write(text + "")
    st = os.stat(p)
    os.utime(p, ns=(st.st_atime_ns, st.st_mtime_ns + 5_000_000))
    b = indexio.load_index_memo(ix_dir, "hla")
    assert b is not a and b["Links"] == a["Links"]


def _py_keys(keys):
    names, idx, split = [], {}, []
    for key in keys:
        al = key.split("-")
        for a in al:
            if a not in idx:
                idx[a] = len(names)
                names.append(a)
        split.append(al)
    A = len(names)
    ap = capi.a_pad(max(A, 1))
    bits = np.zeros((len(split), ap // 64), np.uint64)
    for c, al in enumerate(split):
        for a in al:
            bits[c, idx[a] >> 6] |= np.uint64(1) << np.uint64(idx[a] & 63)
    rank = None
    if all(al == sorted(al) for al in split):
        rank = np.zeros(A, np.int32)
        rank[np.array(sorted(range(A), key=lambda i: names[i]), np.int64)] = np.arange(A, dtype=np.int32)
    return names, bits, ap, rank


@pytest.mark.parametrize("keys", [
    ["A*01:01-A*02:01", "A*02:01", "A*01:01-A*03:01:01:02-A*11:01"],
    ["b-a", "a"],                                   # a key that is NOT sorted: no name rank (the EM keeps its own order)
    [""], ["x"], ["a-b", "b-c", "", "c-c"],         # the empty key is the allele ""; repeated names in one key
    ["é-ü", "a-é"],                                 # UTF-8: bytewise order = code-point order = Python's str order
    ["D8S1179*%d" % k for k in range(7, 20)],
])
def test_class_keys_parsed_natively_equal_the_python_walk(keys):
    names, bits, ap, rank = T._keys_to_classes(keys)
    n2, b2, ap2, r2 = _py_keys(keys)
    assert names == n2 and ap == ap2 and np.array_equal(bits, b2)
    assert (rank is None) == (r2 is None) and (rank is None or np.array_equal(rank, r2))


def test_class_keys_of_the_reference_s_recorded_em_input():
    fx = gu.load("hla_7000")
    for em in fx["em"]:
        keys = [gu.class_key(fx, cid) for cid, _ in em["cmpt"]]
        names, bits, ap, rank = T._keys_to_classes(keys)
        n2, b2, ap2, r2 = _py_keys(keys)
        assert names == n2 and ap == ap2 and np.array_equal(bits, b2) and np.array_equal(rank, r2)
    with pytest.raises(ValueError):
        T._keys_to_classes(["a\nb", "c"])


def _bgzf(payload, rng, level_choices=(0, 1, 6)):
    """`payload` as a BGZF file: blocks of 1-64 KB, each at a compression level of its own (0 = stored: the payload's bytes appear in the file as they are)."""
    import struct
    import zlib
    out = bytearray()
    at = 0
    while at < len(payload):
        n = rng.choice([900, 4000, 20000, 65280])
        chunk = payload[at:at + n]
        at += len(chunk)
        c = zlib.compressobj(rng.choice(level_choices), zlib.DEFLATED, -15)
        body = c.compress(chunk) + c.flush()
        bsize = 18 + len(body) + 8 - 1
        assert bsize < 65536
        out += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + body + struct.pack("<II", zlib.crc32(chunk), len(chunk))
    out += bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")       # the EOF block
    return bytes(out)


def test_bgzf_block_chain_in_ranges_equals_the_walk_of_one_thread():
    """The reader walks a BAM's BGZF container in front of the device inflate; since round 6 in ranges on several threads (every range
    finds a block start and the ranges must link up: hgx_bgzf_scan_par).  Same descriptors as the one-thread walk on files whose
    PAYLOAD is full of gzip magics and whole fake block headers (stored blocks: the bytes are in the file verbatim), on truncated and
    corrupted files (both refuse), and below the size where ranges are used."""
    import ctypes as C
    import random
    rng = random.Random(11)
    L = capi.lib()
    fake = b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00"
    def check(data, want_blocks=None):
        for nt in (1, 2, 5, 8, 16):
            nb, same = C.c_int64(), C.c_int32()
            capi.check(L.hgx_bgzf_scan_compare(data, C.c_size_t(len(data)), C.c_int32(nt), C.byref(nb), C.byref(same)))
            assert same.value == 1, (len(data), nt, nb.value)
            if want_blocks is not None:
                assert (nb.value >= 0) == want_blocks, nb.value
        return nb.value
    for size in (300_000, 3_000_000, 9_000_000):
        # payload: random bytes salted with magics, with fake headers whose BSIZE points at other fake headers, and runs of 0x1f
        p = bytearray(rng.randbytes(size))
        for _ in range(size // 3000):
            k = rng.randrange(size - 64)
            kind = rng.random()
            if kind < 0.4:
                p[k:k + 4] = b"\x1f\x8b\x08\x04"
            elif kind < 0.8:
                hop = rng.choice([27, 200, 5000])
                p[k:k + 18] = fake + (hop - 1).to_bytes(2, "little")
                if k + hop + 18 < size:
                    p[k + hop:k + hop + 18] = fake + (hop - 1).to_bytes(2, "little")
            else:
                p[k:k + 40] = b"\x1f" * 40
        data = _bgzf(bytes(p), rng)
        n_blocks = check(data, True)
        assert n_blocks > size // 66000
        check(data[:-5], False)                                            # truncated inside the last block
        broken = bytearray(data)
        broken[len(data) // 2] ^= 0xFF                                      # somewhere in the middle: mostly payload -- the chain still links
        check(bytes(broken))
        cut = bytearray(data)
        # a header in the middle of the file loses its magic: both walks refuse
        at = 0
        for _ in range(n_blocks // 2):
            at += int.from_bytes(data[at + 16:at + 18], "little") + 1
        cut[at] = 0
        check(bytes(cut), False)
    check(b"not a bgzf file at all" * 100000, False)
