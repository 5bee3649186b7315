"""Pin the Python oracle (oracle/pyref.py) against vectors recorded from the real reference."""
import os
import sys

import pytest

import golden_util as gu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import pyref  # noqa: E402


def _run(fx, memo=False):
    o = fx["options"]
    rl = pyref.RefLocus(fx["_locus"], num_editdist=o["num_editdist"], error_correction=o["error_correction"],
                        allow_discordant=o["allow_discordant"], remove_low=o["remove_low"],
                        simulation=o["simulation"])
    rl.trace = []
    if memo:
        rl.memo = {}               # the per-key cache the full-size oracle runs use: must change nothing
    res = rl.run(fx["sam"])
    return rl, res


@pytest.mark.parametrize("memo", [False, True], ids=["plain", "memo"])
@pytest.mark.parametrize("name", gu.SMALL)
def test_pyref_matches_reference(name, memo):
    fx = gu.load(name)
    rl, res = _run(fx, memo)
    # G4 alternatives
    assert {k: sorted(v) for k, v in rl.alts.left.items()} == fx["alts"]["left"]
    assert {k: sorted(v) for k, v in rl.alts.right.items()} == fx["alts"]["right"]
    # G5 pileup
    assert ["".join(sorted(s)) for s in rl.nt_sets] == fx["mpileup"]["nt_set"]
    assert rl.pileup_counts == fx["mpileup"]["counts"]
    # G3 per-record cmp_list2 + ambiguity result
    assert len(rl.trace) == len(fx["records"])
    for got, exp in zip(rl.trace, fx["records"]):
        assert got["cmp"] == exp["cmp"]
        assert got["iad"] == exp["iad"]
    # G6 per-pair pieces and classes
    assert len(res["pairs"]) == len(fx["pairs"])
    for got, exp in zip(res["pairs"], fx["pairs"]):
        assert got["exon"] == exp["exon"]
        assert got["gene"] == exp["gene"]
        assert got["exon_cls"] == gu.class_key(fx, exp["exon_cls"])
        assert got["gene_cls"] == gu.class_key(fx, exp["gene_cls"])
    # G7 EM calls: same ordered inputs, identical floats and iteration counts
    assert len(res["em"]) == len(fx["em"])
    for got, exp in zip(res["em"], fx["em"]):
        assert [[k, v] for k, v in got["cmpt"]] == [[gu.class_key(fx, c), v] for c, v in exp["cmpt"]]
        assert got["remove_low"] == exp["remove_low"] and got["use_length"] == exp["use_length"]
        assert got["n_iter"] == exp["n_iter"]
        assert [[a, repr(p)] for a, p in got["result"]] == exp["result"]
    # G8 report body
    o = fx["options"]
    lines = pyref.report_lines(res, o["simulation"], o["sample"] if o["simulation"] else (), True)
    exp_lines = [l for l in fx["report"].split("\n") if "aligned" in l or "ranked" in l or "(count:" in l]
    got_lines = [l for l in lines if "aligned" in l or "ranked" in l or "(count:" in l]
    assert got_lines == exp_lines


def test_pyref_hla_7000_classes_and_em():
    fx = gu.load("hla_7000")
    rl, res = _run(fx)
    assert res["num_reads"] == len(fx["records"])
    for got, exp in zip(res["pairs"], fx["pairs"]):
        assert got["exon"] == exp["exon"] and got["gene"] == exp["gene"]
        assert got["gene_cls"] == gu.class_key(fx, exp["gene_cls"])
    for got, exp in zip(res["em"], fx["em"]):
        assert got["n_iter"] == exp["n_iter"]
        assert [[a, repr(p)] for a, p in got["result"]] == exp["result"]
