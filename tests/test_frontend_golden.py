"""Host front-end of libhgx (SAM -> pieces, C++) against vectors recorded from the real reference.
CPU only: device entry points are not called; the class check evaluates the masks with numpy."""
import numpy as np
import pytest

import golden_util as gu
import tables
from hisatgenotype_amd import locus as hl
from test_host_pieces import _numpy_classes


def _parse(fx, keep_trace=True):
    o = fx["options"]
    pl = hl.PackedLocus.from_synth(fx["_locus"])
    batch = pl.parse_sam(fx["sam"], num_editdist=o["num_editdist"], error_correction=o["error_correction"],
                         allow_discordant=o["allow_discordant"], simulation=o["simulation"], keep_trace=keep_trace)
    return pl, batch


from trace_util import check_pileup, check_trace  # noqa: E402


@pytest.mark.parametrize("name", gu.ALL)
def test_alternatives_and_pileup(name):
    fx = gu.load(name)
    pl, batch = _parse(fx, keep_trace=False)
    got = {"L": {}, "R": {}}
    for line in pl.alternatives_text().splitlines():
        d, k, a = line.split("\t")
        got[d].setdefault(k, set()).add(a)
    assert {k: sorted(v) for k, v in got["L"].items()} == fx["alts"]["left"]
    assert {k: sorted(v) for k, v in got["R"].items()} == fx["alts"]["right"]
    check_pileup(fx, batch)


@pytest.mark.parametrize("name", gu.ALL)
def test_records_pieces_classes(name):
    fx = gu.load(name)
    pl, batch = _parse(fx)
    assert batch.n_reads == len(fx["records"])
    assert batch.n_pairs == len(fx["pairs"])
    check_trace(fx, batch)
    # number of add_count calls per pair and level (core:1250-1270)
    for p, exp in enumerate(fx["pairs"]):
        refs = batch.pair_ref[batch.pair_off[p]:batch.pair_off[p + 1]]
        assert int((refs >> 31).sum()) == len(exp["gene"])
        if fx["_locus"].base_fname == "hla":
            assert int((refs >> 31 == 0).sum()) == len(exp["exon"])
    # classes implied by the pieces == classes recorded from the reference
    t = tables.oracle_tables(fx["_locus"])
    A = t["n_alleles"]
    w = (A + 63) // 64
    got_e, got_g = _numpy_classes(pl, batch)
    for p, exp in enumerate(fx["pairs"]):
        assert np.array_equal(got_g[p, :w], gu.class_bits(fx, exp["gene_cls"], A)), p
        if fx["_locus"].base_fname == "hla":
            assert np.array_equal(got_e[p, :w], gu.class_bits(fx, exp["exon_cls"], A)), p


def test_missing_nm_tag_is_an_error():
    """Quirk Q8: the reference raises on a record without NM; the front-end reports it instead of guessing."""
    from hisatgenotype_amd import capi
    fx = gu.load("hla_small_pair")
    pl = hl.PackedLocus.from_synth(fx["_locus"])
    bad = "\n".join(l.replace("NM:i:", "XM:i:") for l in fx["sam"].split("\n")[:4]) + "\n"
    with pytest.raises(capi.HgxError):
        pl.parse_sam(bad, simulation=True)


def test_empty_input():
    fx = gu.load("hla_small_pair")
    pl = hl.PackedLocus.from_synth(fx["_locus"])
    b = pl.parse_sam("")
    assert b.n_pairs == 0 and b.n_reads == 0 and b.n_pieces == 0


def test_malformed_records_fail_as_the_reference_does():
    """tests/golden/malformed_records.json (made by driving the REAL reference, make_malformed_golden.py): a record the reference's
    loop cannot take apart -- too few fields, FLAG / POS / NM that int() rejects, a SEQ shorter than its CIGAR -- kills the
    reference with ValueError / IndexError / TypeError; the front-end refuses the input with the same exception named in its
    message instead of dropping the record.  One documented difference: a BLANK line is no record here (the
    reader drops blank and header lines, as `samtools view` never prints one) where the reference, fed one, dies of it."""
    import json
    import os
    from hisatgenotype_amd import capi
    fx = gu.load("hla_small_pair")
    pl = hl.PackedLocus.from_synth(fx["_locus"])
    cases = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "malformed_records.json")))["cases"]
    assert len(cases) >= 12
    for c in cases:
        want = c["reference_exception"]
        if c["name"] == "blank_line":
            assert want == "ValueError"
            pl.parse_sam(c["sam"], simulation=True)
            continue
        if not want:
            assert pl.parse_sam(c["sam"], simulation=True).n_reads > 0
            continue
        with pytest.raises(capi.HgxError) as e:
            pl.parse_sam(c["sam"], simulation=True)
        msg = str(e.value)
        assert "the reference would fail on this input" in msg and want in msg, (c["name"], want, msg)
        if want == "ValueError":                       # the message the reference printed, word for word
            assert c["reference_message"].split(": ", 1)[1] in msg, (c["name"], msg)
