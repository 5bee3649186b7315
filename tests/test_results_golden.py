"""8f-4: the report summariser against the REAL reference's `call_nuance_results` / `flatten` / `result_process`
(vectors recorded by tests/golden/make_results_golden.py): parsed trees, flattened lists at every trim level, the printed
text and the CSV -- identical, including the cases where the reference raises."""
import contextlib
import gzip
import io
import json
import os
import types

import pytest

import hisatgenotype_amd as hgx
from hisatgenotype_amd import results

HERE = os.path.dirname(os.path.abspath(__file__))
with gzip.open(os.path.join(HERE, "golden", "results_summary.json.gz"), "rb") as f:
    CASES = json.loads(f.read().decode())


def _listify(x):            # JSON turned the reference's tuples into lists
    if isinstance(x, (list, tuple)):
        return [_listify(v) for v in x]
    if isinstance(x, dict):
        return {k: _listify(v) for k, v in x.items()}
    return x


@pytest.mark.parametrize("name", sorted(CASES))
def test_summariser_matches_reference(name, tmp_path):
    case = CASES[name]
    path = tmp_path / ("assembly_graph-hla.%s.report" % name)
    path.write_text(case["report"])
    if "error" in case:
        etype, msg = case["error"].split(": ", 1)
        with pytest.raises(Exception) as ei:
            results.call_nuance_results(str(path))
        assert type(ei.value).__name__ == etype and str(ei.value) == msg
        return
    tree = results.call_nuance_results(str(path))
    assert tree == case["datatree"]
    assert list(tree["EM"]) == list(case["datatree"]["EM"])                      # gene order = report order
    for trim, exp in case["process"].items():
        flat = {g: results.flatten(t["children"], g, trim=int(trim)) for g, t in tree["Allele splitting"].items()}
        assert _listify(flat) == exp["flatten"]
        args = types.SimpleNamespace(read_dir=str(tmp_path), trim_level=int(trim), csv=True, ofile=str(tmp_path / ("o%s.csv" % trim)))
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            results.result_process(args)
        assert buf.getvalue().replace(str(tmp_path), "<DIR>") == exp["stdout"]
        assert open(args.ofile).read().replace(str(tmp_path), "<DIR>") == exp["csv"]


def test_summariser_is_exported_and_has_a_command_line(tmp_path, capsys):
    assert hgx.call_nuance_results is results.call_nuance_results and hgx.build_tree is results.build_tree
    (tmp_path / "x.report").write_text(CASES["split_resolution"]["report"])
    results.main(["--in-dir", str(tmp_path), "-t", "3"])
    out = capsys.readouterr().out
    assert "A*01:01:01 - Trimmed (score: 0.5000)" in out and "Analysis - Allele splitting" in out
