#!/usr/bin/env python3
"""Golden vectors for the simulation self-test loop of genotyping_locus (8f-2) from the REAL reference.

The reference's `typing_core.genotyping_locus` (typing_core.py:2278-2648) is run in simulation mode (no reads, no alignment)
on an index written by `synth.write_index`: it samples the test alleles (`random.seed(set_seed)`; `random.sample`), writes
reads with ITS `typing_common.simulate_reads`, calls ITS `typing(simulation=True)` per test and prints "Passed so far".
Harness accommodations (nothing of the reference is stored): the network steps (clone / download / extract / build index)
are no-ops; `samtools` is the stub of make_golden.py (now honouring the RNAME argument of `view`); HISAT2 is absent, so
`typing_common.align_reads` is replaced by `hisatgenotype_amd.simulate.truth_align` -- the alignment each simulated read
spells in its name -- on BOTH sides of the comparison.
Recorded: index files, parameters, every FASTA file the loop wrote (sha256 + size + first record), the stderr transcript
(time stamps and the version / command header dropped), every report body and the pass counts.
Run: PYTHONHASHSEED=0 python tests/golden/make_selftest_golden.py"""
import contextlib
import gzip
import hashlib
import io
import json
import os
import re
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (re-execs with PYTHONHASHSEED=0, puts the package on sys.path)
from hisatgenotype_amd import indexio, simulate, synth  # noqa: E402

CASES = {
    # name: (loci builder, genotyping_locus keyword values, debug_instr)
    "pairs_two_genes": dict(
        loci=lambda: [synth.make_hla_like_locus(gene="A", n_alleles=40, n_vars=160, length=1800, seed=41, sibling_frac=0.3),
                      synth.make_hla_like_locus(gene="B", n_alleles=30, n_vars=140, length=1600, seed=42, var_id_base=5000)],
        locus_list=["A", "B"], simulate_interval=13, read_len=100, fragment_len=350, perbase_errorrate=0.0,
        debug={"pair": True, "test_size": "3", "set_seed": 7}),
    "basic_with_errors": dict(
        loci=lambda: [synth.make_hla_like_locus(gene="A", n_alleles=60, n_vars=200, length=2000, seed=43, sibling_frac=0.3,
                                                deletion_frac=0.1)],
        locus_list=["A"], simulate_interval=9, read_len=100, fragment_len=300, perbase_errorrate=0.3,
        debug={"test_size": "4", "set_seed": 11}),
    "single_test_id_and_list": dict(
        loci=lambda: [synth.make_hla_like_locus(gene="A", n_alleles=40, n_vars=160, length=1800, seed=41, sibling_frac=0.3)],
        locus_list=["A"], simulate_interval=17, read_len=100, fragment_len=350, perbase_errorrate=0.0,
        debug={"pair": True, "test_size": "5", "set_seed": 3, "test_id": "2-4"}),
}


def _clean(text):
    out = []
    lines = text.split("\n")
    k = 0
    while k < len(lines):
        l = lines[k]
        if l.startswith("# COMMAND"):
            k += 2                      # the command line follows on its own line
            continue
        if l.startswith("#"):
            k += 1
            continue
        out.append(re.sub(r"^(Test \d+) .*$", r"\1", l))
        k += 1
    return "\n".join(out)


def _tree(root):
    files = {}
    for d, _, fs in os.walk(root):
        for f in fs:
            p = os.path.join(d, f)
            data = open(p, "rb").read()
            rec = {"sha256": hashlib.sha256(data).hexdigest(), "bytes": len(data)}
            if f.endswith(".fa"):
                rec["head"] = data.decode().split("\n")[:2]
            if f.endswith(".report"):
                rec["body"] = _clean(data.decode())
                del rec["sha256"], rec["bytes"]
            files[os.path.relpath(p, root)] = rec
    return files


def main():
    tmp = mg.setup_reference()
    # the stub samtools of make_golden.py, honouring `view FILE RNAME` (the reference always passes the backbone, core:443)
    stub = os.path.join(tmp, "bin", "samtools")
    with open(stub, "w") as f:
        f.write("#!/bin/sh\ncase \"$1\" in\n"
                "  view) if [ -n \"$3\" ]; then grep -v '^@' \"$2\" | awk -F'\\t' -v r=\"$3\" '$3==r'; else grep -v '^@' \"$2\"; fi ;;\n"
                "  index) touch \"$2.bai\" ;;\n  *) exit 1 ;;\nesac\n")
    import hisatgenotype_typing_common as common
    import hisatgenotype_typing_core as core
    for name in ("clone_hisatgenotype_database", "download_genome_and_index", "extract_database_if_not_exists",
                 "build_index_if_not_exists"):
        setattr(common, name, lambda *a, **k: None)
    out = {}
    try:
        for case, spec in CASES.items():
            work = os.path.join(tmp, "work_" + case)
            ix_dir, out_dir = os.path.join(work, "ix"), os.path.join(work, "out")
            os.makedirs(out_dir)
            loci = spec["loci"]()
            synth.write_index(loci, ix_dir, "hla")
            ix = indexio.load_index(ix_dir, "hla")

            def align(aligner, simulation, index_name, index_type, base_fname, read_fname, fastq, threads, out_fname, verbose):
                simulate.truth_align(read_fname, out_fname, ix["Genes"], ix["Vars"], ix["refGenes"])
            common.align_reads = align
            cwd = os.getcwd()
            os.chdir(work)
            err = io.StringIO()
            try:
                with contextlib.redirect_stderr(err):
                    core.genotyping_locus("hla", list(spec["locus_list"]), "", ix_dir, [], True, [["hisat2", "graph"]], [], False,
                                          "", 1, spec["simulate_interval"], spec["read_len"], spec["fragment_len"], False, 2,
                                          spec["perbase_errorrate"], 0.0, [], False, "assembly_graph", True, False, False, False,
                                          True, [], 0, False, out_dir, False, dict(spec["debug"]))
            finally:
                os.chdir(cwd)
            files = {f: open(os.path.join(ix_dir, f)).read() for f in sorted(os.listdir(ix_dir)) if not f.endswith(".npz")}
            # the reference walks the genes in SET order (typing_core.py:2508), i.e. in an order that depends on the
            # interpreter's hash seed; record the order this run used so that the comparison can ask for the same one
            gene_order = list(set(spec["locus_list"]) & set(ix["Gene_names"].keys()))
            out[case] = {"index_files": files, "params": {k: v for k, v in spec.items() if k != "loci"}, "gene_order": gene_order,
                         "stderr": _clean(err.getvalue()), "out_dir": _tree(out_dir),
                         "cwd_fasta": {f: _tree(work)[f] for f in ("hla_input_1.fa", "hla_input_2.fa")}}
            print("%-26s tests run: %d, transcript tail: %s" % (
                case, out[case]["stderr"].count("Test "), out[case]["stderr"].strip().split("\n")[-1]))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    dst = os.path.join(HERE, "selftest_loop.json.gz")
    with gzip.GzipFile(dst, "wb", mtime=0) as f:
        f.write(json.dumps(out, separators=(",", ":")).encode())
    print("-> %s (%.1f KB)" % (dst, os.path.getsize(dst) / 1024.0))


if __name__ == "__main__":
    main()
