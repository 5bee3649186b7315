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
