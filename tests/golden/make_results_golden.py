#!/usr/bin/env python3
"""Golden vectors for the report summariser (8f-4) from the REAL reference: `typing_common.call_nuance_results` and the
`hisatgenotype_parse_results` tool's `flatten` / `result_process` (imported from /root/reference in the build container)
run on report files -- the report text the reference itself wrote for the committed typing fixtures, plus hand-written
reports for the corners (split resolution, '***' lines of the simulation mode, an Assembly section, several genes).
Stores inputs and expected outputs only: tests/golden/results_summary.json.gz.
Run: python tests/golden/make_results_golden.py"""
import contextlib
import gzip
import importlib.util
import io
import json
import os
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, os.path.join(REF, "hisatgenotype_modules"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

import hisatgenotype_typing_common as common  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_parse_results", os.path.join(REF, "hisatgenotype_tools", "hisatgenotype_parse_results.py"))
tool = importlib.util.module_from_spec(spec)
spec.loader.exec_module(tool)

import golden_util as gu  # noqa: E402

HAND = {
    "split_resolution": "\n".join([
        "\t\t\t\t1 ranked A*01:01:01:01 (abundance: 25.00%)",
        "\t\t\t\t2 ranked A*01:01:01:02 (abundance: 25.00%)",
        "\t\t\t\t3 ranked A*02:01:01:01 (abundance: 50.00%)", ""]),
    "simulation_stars_two_genes": "\n".join([
        "\t\t\t200 reads and 100 pairs are aligned",
        "\t\t\t*** 1 ranked A*03:01:01:01 (count: 100)",
        "\t\t\t*** 1 ranked A*03:01:01:01 (abundance: 51.20%)",
        "\t\t\t\t2 ranked A*11:01:01:01 (abundance: 30.05%)",
        "\t\t\t\t3 ranked A*11:01:02 (abundance: 18.75%)",
        "\t\t\t*** 1 ranked DRB1*15:01:01:02 (abundance: 60.00%)",
        "\t\t\t\t2 ranked DRB1*15:01:01:01 (abundance: 21.00%)",
        "\t\t\t\t3 ranked DRB1*04:03 (abundance: 19.00%)", ""]),
    "with_assembly": "\n".join([
        "\t\t\t\t1 ranked B*07:02:01:01 (abundance: 70.00%)",
        "\t\t\t\t2 ranked B*08:01:01 (abundance: 30.00%)",
        "Assembly graph results",
        "B: B*07:02:01:01 and B*08:01:01",
        "note: second line", ""]),
    "repeated_allele_and_low": "\n".join([
        "\t\t\t\t1 ranked C*04:01:01:01 (abundance: 45.00%)",
        "\t\t\t\t2 ranked C*04:01:01:01 (abundance: 35.00%)",
        "\t\t\t\t3 ranked C*07:02 (abundance: 15.00%)",
        "\t\t\t\t4 ranked C*07 (abundance: 5.00%)", ""]),
    "name_extends_a_leaf": "\n".join([
        "\t\t\t\t1 ranked C*07 (abundance: 60.00%)",
        "\t\t\t\t2 ranked C*07:02 (abundance: 40.00%)", ""]),
    "single_model_line": "\n".join([
        "\t\t\t\t1 ranked A*01:01 (abundance: 50.00%)",
        "SingleModel A*01:01 (abundance: 50.00%)", ""]),
}


def main():
    cases = dict(HAND)
    for name in ("hla_mid_real", "hla_small_pair", "hla_7000", "codis_like"):
        cases["fixture_" + name] = gu.load(name)["report"]
    # the 13 reports the reference itself holds as expected outputs of its integration runs (devel/hg_test1..5: data, not source)
    import glob
    for path in sorted(glob.glob(os.path.join(REF, "devel", "hg_test*", "*.report"))):
        cases["devel_%s_%s" % (os.path.basename(os.path.dirname(path)), os.path.basename(path).split(".")[1])] = open(path).read()
    out = {}
    for name, text in cases.items():
        d = tempfile.mkdtemp()
        path = os.path.join(d, "assembly_graph-hla.%s.report" % name)
        with open(path, "w") as f:
            f.write(text)
        entry = {"report": text}
        try:
            entry["datatree"] = common.call_nuance_results(path)
        except Exception as e:                 # part of the observable behaviour (e.g. a malformed abundance line)
            entry["error"] = "%s: %s" % (type(e).__name__, e)
            out[name] = entry
            continue
        entry["process"] = {}
        for trim in (4, 3, 2, 1):
            flat = {g: tool.flatten(t["children"], g, trim=trim) for g, t in entry["datatree"]["Allele splitting"].items()}
            args = types.SimpleNamespace(read_dir=d, trim_level=trim, csv=True, ofile=os.path.join(d, "out_%d.csv" % trim))
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                tool.result_process(args)
            entry["process"][str(trim)] = {"flatten": flat, "stdout": buf.getvalue().replace(d, "<DIR>"),
                                           "csv": open(args.ofile).read().replace(d, "<DIR>")}
        out[name] = entry
    dst = os.path.join(HERE, "results_summary.json.gz")
    with gzip.GzipFile(dst, "wb", mtime=0) as f:
        f.write(json.dumps(out, separators=(",", ":")).encode())
    print("%d cases -> %s (%.1f KB)" % (len(out), dst, os.path.getsize(dst) / 1024.0))
    for k, v in out.items():
        print("  %-32s %s" % (k, v.get("error") or "genes=%s" % list(v["datatree"]["EM"].keys())))


if __name__ == "__main__":
    main()
