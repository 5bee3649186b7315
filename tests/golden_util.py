"""Helpers to read the golden fixtures recorded from the real reference (tests/golden/make_golden.py)."""
import functools
import gzip
import json
import os

import numpy as np

from hisatgenotype_amd import synth

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ALL = ["hla_small_pair", "hla_small_single", "hla_errors_filters", "hla_mid_real", "hla_keep_low",
       "hla_single_end", "hla_novel_sample", "hla_insertions", "hla_7000", "codis_like", "codis_d18s51"]
SMALL = [n for n in ALL if n != "hla_7000"]
LEAN = ["hla_7000_10k", "codis_10k"]      # BASELINE configs[0] at full size: SAM, EM calls (with their class dicts) and report only


@functools.lru_cache(maxsize=None)
def load(name):
    with gzip.open(os.path.join(GOLDEN_DIR, name + ".json.gz"), "rb") as f:
        fx = json.loads(f.read().decode())
    if fx["locus"] is not None:
        loc = synth.Locus.from_json(fx["locus"])
    else:
        loc = synth.make_hla_like_locus(**fx["locus_params"])
        assert loc.allele_names == fx["allele_names"], "generator drifted from the recorded fixture"
    fx["_locus"] = loc
    return fx


def class_key(fx, cid):
    """Decode a recorded class id back to the reference's '-'.join(sorted(names)) key."""
    bits = int(fx["classes"][cid], 16)
    names = fx["allele_names"]
    out = []
    i = 0
    while bits:
        if bits & 1:
            out.append(names[i])
        bits >>= 1
        i += 1
    return "-".join(sorted(out))


def class_bits(fx, cid, n_alleles):
    """Recorded class as a bitset over scored alleles (Gene_names minus the backbone at index 0)."""
    bits = int(fx["classes"][cid], 16) >> 1
    w64 = (n_alleles + 63) // 64
    out = np.zeros(w64, dtype=np.uint64)
    for w in range(w64):
        out[w] = (bits >> (64 * w)) & 0xFFFFFFFFFFFFFFFF
    return out


def parse_ht(ht, var_index):
    """'left-id-..-right' -> (left, right, [var indices; -1 for nv ids])."""
    f = ht.split("-")
    ids = [var_index.get(v, -1) if v.startswith("hv") else -1 for v in f[1:-1]]
    return int(f[0]), int(f[-1]), ids
