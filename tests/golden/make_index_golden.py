#!/usr/bin/env python3
"""Golden vectors for the index-file readers (8f-1): files written by ``synth.write_index`` are parsed with the REAL
reference's readers (typing_common.read_locus / read_variants / read_links / read_allele_seq and
typing_core.read_Gene_alleles_from_vars) and the parsed structures are recorded.  Data only; run in the build container:
    PYTHONHASHSEED=0 python tests/golden/make_index_golden.py
"""
import gzip
import json
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (re-execs with PYTHONHASHSEED=0, sets up sys.path for the package)
from hisatgenotype_amd import synth  # noqa: E402


def main():
    tmp = mg.setup_reference()
    import hisatgenotype_typing_common as common
    import hisatgenotype_typing_core as core
    try:
        a = synth.make_hla_like_locus(gene="A", n_alleles=60, n_vars=150, seed=1, unlinked_vars=2, insertion_frac=0.05)
        b = synth.make_hla_like_locus(gene="B", n_alleles=40, n_vars=120, seed=2, var_id_base=1000, length=2000)
        c = synth.make_str_like_locus(gene="TH01", unit="AATG", max_repeats=12, min_repeats=5, flank=120, seed=3, var_id_base=2000)
        d = os.path.join(tmp, "ix")
        synth.write_index([a, b, c], d, "hla")
        full = os.path.join(d, "hla")
        refGenes, refGene_loci = common.read_locus(full + ".locus", False, "hla", {}, {})
        Vars, Var_list = common.read_variants(full + ".snp", True)
        Links = common.read_links(full + ".link")
        Genes = common.read_allele_seq(full + "_backbone.fa", {}, True)
        core.read_Gene_alleles_from_vars(Vars, Var_list, Links, Genes)
        files = {f: open(os.path.join(d, f)).read() for f in sorted(os.listdir(d))}
        fx = {"files": files,
              "refGenes": refGenes, "refGene_loci": refGene_loci, "Vars": Vars, "Var_list": Var_list, "Links": Links,
              "Gene_names": {g: list(v.keys()) for g, v in Genes.items()},
              "Gene_lengths": {g: {n: len(s) for n, s in v.items()} for g, v in Genes.items()},
              "backbones": {g: v[refGenes[g]] for g, v in Genes.items()}}
        out = os.path.join(HERE, "index_files.json.gz")
        with gzip.GzipFile(out, "wb", mtime=0) as f:
            f.write(json.dumps(fx, separators=(",", ":")).encode())
        print("index golden: %d files, genes %s, %.1f KB" % (len(files), sorted(refGenes), os.path.getsize(out) / 1024.0))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
