"""The shard-by-shard form of the oracle chain (tests/oracle_util.py: oracle_type_sharded, what the full-size configs[1] GPU test
runs on the host cores) must BE the oracle chain: every field `==` the one-process run -- pileup injection, the per-key memo of
pyref, class dicts merged in stream order, Gene_counts and first pairs, both EMs."""
import numpy as np
import pytest

import golden_util as gu
import oracle_util as ou
from hisatgenotype_amd import synth


class _Serial:
    def map(self, fn, *cols):
        return [fn(*a) for a in zip(*cols)]


def _same(a, b):
    assert (a["num_reads"], a["num_pairs"]) == (b["num_reads"], b["num_pairs"])
    assert np.array_equal(a["gene_counts"], b["gene_counts"]) and np.array_equal(a["first_pair"], b["first_pair"])
    for key in ("gene_classes", "exon_classes"):
        if a.get(key) is None or a[key][0] is None:
            assert b.get(key) is None or b[key][0] is None
            continue
        assert np.array_equal(a[key][0], b[key][0]) and np.array_equal(a[key][1], b[key][1]), key
    assert a["em"] == b["em"] and a["gene_prob"] == b["gene_prob"]


@pytest.mark.parametrize("n_shards", [2, 5])
def test_sharded_oracle_is_the_oracle_hla(n_shards):
    loc = synth.make_hla_like_locus(n_alleles=400, n_vars=900, seed=77, unlinked_vars=3)
    sample = synth.pick_sample(loc, 5)
    al = synth.simulate_pairs(loc, sample, 900, err_rate=0.006, seed=3, softclip_frac=0.03, novel_del_frac=0.03, multi_hit_frac=0.01,
                              dup_frac=0.2)
    sam = synth.sam_text(loc, al)
    one = ou.oracle_type(loc.to_json(), sam)
    many = ou.oracle_type_sharded(loc, sam, n_shards=n_shards, executor=_Serial())
    _same(one, many)


def test_sharded_oracle_is_the_oracle_on_a_recorded_fixture():
    """... and on a sample of the real reference (fixture hla_mid_real): the merged exon-level class dict is the reference's own."""
    fx = gu.load("hla_mid_real")
    o = fx["options"]
    loc = fx["_locus"]
    many = ou.oracle_type_sharded(loc, fx["sam"], n_shards=3, simulation=o["simulation"], allow_discordant=o["allow_discordant"],
                                  remove_low=o["remove_low"], executor=_Serial())
    A = len([n for n in loc.allele_names if "BACKBONE" not in n])
    exp = fx["em"][0]
    want = np.stack([gu.class_bits(fx, cid, A) for cid, _ in exp["cmpt"]])
    bits, cnt = many["exon_classes"] if loc.base_fname == "hla" else many["gene_classes"]
    assert np.array_equal(bits[:, :want.shape[1]], want) and cnt.tolist() == [n for _, n in exp["cmpt"]]
    assert [it for _, it, _ in many["em"]] == [e["n_iter"] for e in fx["em"]]
    for (_, _, got), e in zip(many["em"], fx["em"]):
        assert [[a, repr(p)] for a, p in got] == e["result"]


def test_split_never_cuts_a_read_name():
    fx = gu.load("hla_small_pair")
    parts = ou.split_name_grouped(fx["sam"], 7)
    assert "".join(parts) == "\n".join(l for l in fx["sam"].split("\n") if l) + "\n"
    for a, b in zip(parts, parts[1:]):
        assert a.rstrip("\n").rsplit("\n", 1)[-1].split("\t", 1)[0] != b.split("\t", 1)[0]
