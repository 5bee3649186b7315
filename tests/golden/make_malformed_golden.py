#!/usr/bin/env python3
"""Malformed alignment records through the REAL reference (typing_core.py:800-898): which exception does its record loop raise?

Takes the first records of the hla_small_pair fixture, damages one record per case (short line, non-numeric POS / FLAG / NM,
SEQ shorter than the CIGAR, '*' SEQ, a blank line) and runs hisatgenotype_typing_core.typing() on the result with the harness of
make_golden.py.  Records, per case: the SAM text and the exception type the reference died with ("" if it ran through).
Output: tests/golden/malformed_records.json (data only).  Run in the build container: python3 tests/golden/make_malformed_golden.py
"""
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import make_golden as mg  # noqa: E402
import golden_util as gu  # noqa: E402


def cases(lines):
    def with_line(k, new):
        out = list(lines)
        out[k] = new
        return "\n".join(out) + "\n"
    c2 = lines[2].split("\t")
    def edit(i, v):
        d = list(c2)
        d[i] = v
        return "\t".join(d)
    nm = next(i for i, t in enumerate(c2) if t.startswith("NM:i:"))
    return {
        "five_fields": with_line(2, "\t".join(c2[:5])),
        "eight_fields": with_line(2, "\t".join(c2[:8])),
        "ten_fields_no_qual": with_line(2, "\t".join(c2[:10])),
        "eleven_fields_no_tags": with_line(2, "\t".join(c2[:11])),
        "pos_not_a_number": with_line(2, edit(3, "abc")),
        "flag_not_a_number": with_line(2, edit(1, "x99")),
        "pos_empty_like": with_line(2, edit(3, "12x")),
        "nm_not_a_number": with_line(2, edit(nm, "NM:i:one")),
        "seq_shorter_than_cigar": with_line(2, edit(9, c2[9][:40])),
        "seq_star": with_line(2, edit(9, "*")),
        "blank_line": with_line(2, ""),
        "bad_flag_on_a_record_left_of_the_locus": with_line(2, "\t".join([c2[0], "zz", c2[2], "0"] + c2[4:])),
        "intact": "\n".join(lines) + "\n",
    }


def main():
    fx = gu.load("hla_small_pair")
    loc = fx["_locus"]
    lines = [l for l in fx["sam"].split("\n") if l][:8]
    tmp = mg.setup_reference()
    import hisatgenotype_typing_common as common
    import hisatgenotype_typing_core as core
    out = {"source": "first 8 records of tests/golden/hla_small_pair.json.gz, one record damaged per case", "cases": []}
    try:
        for name, sam in cases(lines).items():
            o = fx["options"]
            cap, report, err = mg.run_reference(core, common, loc, sam, simulation=o["simulation"], sample=o["sample"],
                                                workdir=os.path.join(tmp, "mal_" + name), profile_closures=False)
            kind = (err or "").split(":", 1)[0]
            out["cases"].append({"name": name, "sam": sam, "reference_exception": kind, "reference_message": err or ""})
            print("%-42s %s" % (name, err))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    with open(os.path.join(HERE, "malformed_records.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
