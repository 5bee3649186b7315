"""Index-file readers (8f-1) against structures parsed by the real reference's readers (tests/golden/make_index_golden.py)."""
import gzip
import json
import os

from hisatgenotype_amd import indexio

HERE = os.path.dirname(os.path.abspath(__file__))


def test_readers_match_reference(tmp_path):
    fx = json.loads(gzip.open(os.path.join(HERE, "golden", "index_files.json.gz")).read().decode())
    for name, text in fx["files"].items():
        (tmp_path / name).write_text(text)
    ix = indexio.load_index(str(tmp_path), "hla")
    assert ix["refGenes"] == fx["refGenes"]
    assert ix["refGene_loci"] == fx["refGene_loci"]
    assert ix["Vars"] == fx["Vars"] and ix["Var_list"] == fx["Var_list"] and ix["Links"] == fx["Links"]
    for g in fx["Gene_names"]:
        ref_names = fx["Gene_names"][g]
        # the reference appends backbone-equal alleles later (from .allele, in set order: quirk Q7); everything it
        # derives from the variant files must agree in content AND order
        assert ix["Gene_names"][g][:len(ref_names)] == ref_names
        assert {n: ix["Gene_lengths"][g][n] for n in ref_names} == fx["Gene_lengths"][g]
        assert ix["Genes"][g][fx["refGenes"][g]] == fx["backbones"][g]


def test_packed_locus_from_index_files(tmp_path):
    """A locus loaded from files packs to the same tables as the locus it was written from."""
    import numpy as np
    from hisatgenotype_amd import synth, locus as hl
    a = synth.make_hla_like_locus(gene="A", n_alleles=90, n_vars=200, seed=8)
    synth.write_index([a], str(tmp_path), "hla")
    ix = indexio.load_index(str(tmp_path), "hla")
    p1 = hl.PackedLocus.from_synth(a)
    p2 = hl.PackedLocus.from_reference_dicts("A", "hla", ix["refGenes"], ix["Genes"], ix["Gene_names"], ix["Gene_lengths"],
                                             ix["refGene_loci"], ix["Vars"], ix["Var_list"], ix["Links"])
    assert p1.names == p2.names
    t1, t2 = p1.tables(), p2.tables()
    for k in t1:
        assert np.array_equal(t1[k], t2[k]), k
    assert np.array_equal(p1.allele_len, p2.allele_len)


def test_packed_locus_cache_next_to_index(tmp_path):
    """indexio.packed_locus writes `<base>.<gene>.hgx.npz` on first use and serves the identical locus from it afterwards;
    touching an index file invalidates it."""
    import os, time
    import numpy as np
    from hisatgenotype_amd import synth, indexio
    loc = synth.make_hla_like_locus(n_alleles=120, n_vars=100, seed=21)
    synth.write_index([loc], str(tmp_path), "hla")
    gene = loc.gene
    a = indexio.packed_locus(str(tmp_path), "hla", gene)
    cache = os.path.join(str(tmp_path), "hla.%s.hgx.npz" % gene)
    assert os.path.exists(cache)
    b = indexio.packed_locus(str(tmp_path), "hla", gene)
    assert a.names == b.names and a.var_ids == b.var_ids
    ta, tb = a.tables(), b.tables()
    assert all(np.array_equal(ta[k], tb[k]) for k in ta)
    m0 = os.path.getmtime(cache)
    os.utime(os.path.join(str(tmp_path), "hla.snp"), (time.time() + 5, time.time() + 5))   # index changed after the cache was written
    indexio.packed_locus(str(tmp_path), "hla", gene)
    assert os.path.getmtime(cache) >= m0          # rebuilt from the text files and rewritten


def test_genome_index_readers_match_reference(tmp_path):
    """Genotype-genome index (typing_core.py:2326-2397): load_genome_index against the structures the real reference's
    readers produced from the same files (tests/golden/make_genome_index_golden.py) -- loci in chromosome coordinates,
    variants re-based to their locus, backbones cut out of the genome FASTA through the .fai index, allele order and lengths;
    allele_sequence spells out alleles the way read_Gene_alleles_from_vars does."""
    fx = json.loads(gzip.open(os.path.join(HERE, "golden", "genome_index.json.gz")).read().decode())
    for name, text in fx["files"].items():
        (tmp_path / name).write_text(text)
    ix = indexio.load_genome_index(str(tmp_path), "genotype_genome", "hla")
    assert ix["refGenes"] == fx["refGenes"]
    assert ix["refGene_loci"] == fx["refGene_loci"]
    assert ix["Vars"] == fx["Vars"] and ix["Var_list"] == fx["Var_list"] and ix["Links"] == fx["Links"]
    assert sorted(ix["alleles"]) == fx["alleles"] and ix["partial_alleles"] == set()
    for g in fx["Gene_names"]:
        ref_names = fx["Gene_names"][g]
        assert ix["Gene_names"][g][:len(ref_names)] == ref_names
        assert {n: ix["Gene_lengths"][g][n] for n in ref_names} == fx["Gene_lengths"][g]
        assert ix["Genes"][g][fx["refGenes"][g]] == fx["backbones"][g]
        left, right = fx["spans"][g]
        assert ix["refGene_loci"][g][2:4] == [left, right]
        for n, seq in fx["sample_sequences"][g].items():
            assert indexio.allele_sequence(ix, g, n) == seq
    assert indexio.load_genome_index(str(tmp_path), "genotype_genome", "codis")["refGenes"] == {}      # another family: nothing
