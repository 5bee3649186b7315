"""8f-2: the simulation self-test loop of genotyping_locus against the REAL reference's loop
(tests/golden/make_selftest_golden.py drove typing_core.genotyping_locus; HISAT2 replaced on both sides by the alignment
each simulated read spells in its name).

CPU part: allele sampling + simulate_reads -- every FASTA file byte-identical (sha256), the per-test allele lines of the
transcript identical.  GPU part (marked): the whole loop -- transcript with "Passed so far" / totals, every report body."""
import contextlib
import gzip
import hashlib
import io
import json
import os
import re

import pytest

import hisatgenotype_amd as hgx
from hisatgenotype_amd import driver

HERE = os.path.dirname(os.path.abspath(__file__))
with gzip.open(os.path.join(HERE, "golden", "selftest_loop.json.gz"), "rb") as f:
    CASES = json.loads(f.read().decode())


def _clean(text):
    out, lines, k = [], text.split("\n"), 0
    while k < len(lines):
        l = lines[k]
        if l.startswith("# COMMAND"):
            k += 2
            continue
        if not l.startswith("#"):
            out.append(re.sub(r"^(Test \d+) .*$", r"\1", l))
        k += 1
    return "\n".join(out)


def _run(case, tmp_path, monkeypatch):
    spec = CASES[case]
    ix_dir, out_dir = tmp_path / "ix", tmp_path / "out"
    ix_dir.mkdir()
    out_dir.mkdir()
    for name, text in spec["index_files"].items():
        (ix_dir / name).write_text(text)
    p = spec["params"]
    monkeypatch.chdir(tmp_path)
    err = io.StringIO()
    with contextlib.redirect_stderr(err):
        # (the reference iterates a set of gene names; this driver walks locus_list in order: ask for the recorded order)
        passed = hgx.genotyping_locus("hla", list(spec["gene_order"]), "", str(ix_dir), [], True, [["hisat2", "graph"]], [], False, "",
                                      1, p["simulate_interval"], p["read_len"], p["fragment_len"], False, 2, p["perbase_errorrate"],
                                      0.0, [], False, "assembly_graph", True, False, False, False, True, [], 0, False, str(out_dir),
                                      False, dict(p["debug"]))
    return spec, out_dir, err.getvalue(), passed


def _check_fasta(spec, out_dir, tmp_path):
    n = 0
    for rel, rec in list(spec["out_dir"].items()) + list(spec["cwd_fasta"].items()):
        if not rel.endswith(".fa"):
            continue
        path = (out_dir / rel) if rel in spec["out_dir"] else (tmp_path / rel)
        data = path.read_bytes()
        assert data.decode().split("\n")[:2] == rec["head"], rel
        assert len(data) == rec["bytes"] and hashlib.sha256(data).hexdigest() == rec["sha256"], rel
        n += 1
    assert n >= 4


@pytest.mark.parametrize("case", sorted(CASES))
def test_sampling_and_simulated_reads_match_reference(case, tmp_path, monkeypatch):
    """No GPU: typing() is replaced by a stub, so only the loop's own work is compared -- which alleles each test draws
    (random.seed / random.sample as the reference), the reads simulate_reads writes for them (errors from the same random
    stream), and the per-allele lines it prints."""
    monkeypatch.setattr(driver, "typing", lambda *a, **k: {})
    spec, out_dir, err, passed = _run(case, tmp_path, monkeypatch)
    _check_fasta(spec, out_dir, tmp_path)
    pick = lambda t: [l for l in _clean(t).split("\n") if l.startswith("Test ") or " bp (" in l]
    assert pick(err) == pick(spec["stderr"])
    assert passed == {} and "Test Failed!" in err


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(CASES))
def test_selftest_loop_matches_reference(case, tmp_path, monkeypatch):
    spec, out_dir, err, passed = _run(case, tmp_path, monkeypatch)
    _check_fasta(spec, out_dir, tmp_path)
    keep = lambda t: [l for l in _clean(t).split("\n") if l.strip() and "hot path: hgx" not in l]
    assert keep(err) == keep(spec["stderr"])
    n = 0
    for rel, rec in spec["out_dir"].items():
        if rel.endswith(".report"):
            assert keep((out_dir / rel).read_text()) == keep(rec["body"]), rel
            n += 1
    assert n >= 2
    total = [l for l in spec["stderr"].split("\n") if "passed (" in l][-1]
    assert passed == {"hisat2 graph": int(total.split("\t")[1].split("/")[0])}
    assert not os.path.exists(tmp_path / "hla_output.bam")               # keep_alignment=False removes the alignment
