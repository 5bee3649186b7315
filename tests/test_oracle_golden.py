"""Pin the C oracle (oracle/hgx_oracle.c) against vectors recorded from the real reference."""
import numpy as np
import pytest

import golden_util as gu
import tables


def _bits_to_indices(row):
    out = []
    for w, word in enumerate(row):
        word = int(word)
        while word:
            b = (word & -word).bit_length() - 1
            out.append(64 * w + b)
            word &= word - 1
    return out


@pytest.mark.parametrize("name", gu.ALL)
def test_add_count_add_stat(orc, name):
    """add_count + add_stat (typing_core.py:626-677, 1171-1236) on the recorded haplotypes."""
    fx = gu.load(name)
    t = tables.oracle_tables(fx["_locus"])
    L = orc.make_locus(t)
    arrs = tables.pieces_from_pairs(fx["pairs"], t["var_index"])
    hla = fx["_locus"].base_fname == "hla"
    eb, gb, gc, fp = orc.score_pairs(L, t["exon_keys"], t["gene_keys"], *arrs)
    A = t["n_alleles"]
    for i, p in enumerate(fx["pairs"]):
        assert np.array_equal(gb[i], gu.class_bits(fx, p["gene_cls"], A)), (name, i)
        if hla:
            assert np.array_equal(eb[i], gu.class_bits(fx, p["exon_cls"], A)), (name, i)
    # Gene_counts + tie order (typing_core.py:1650-1672) against the report's count lines
    order = sorted([a for a in range(A) if gc[a] > 0], key=lambda a: (fp[a], a))
    order = sorted(order, key=lambda a: -gc[a])
    got = [(t["names"][a], int(gc[a])) for a in order]
    exp = []
    for l in fx["report"].split("\n"):
        if "(count:" in l:
            f = l.strip().replace("*** ", "").replace("ranked ", "").split()
            exp.append((f[1], int(f[3].rstrip(")"))))
    if fx["options"]["simulation"]:
        # simulation prints only the true alleles and the top five (core:1654-1667)
        assert set(exp) <= set(got)
        assert got[0] == exp[0]
    else:
        assert got == exp


@pytest.mark.parametrize("name", gu.ALL)
def test_dedup_and_em(orc, name):
    """Class dict accumulation + single_abundance (typing_common.py:1282-1410), bit-identical floats."""
    fx = gu.load(name)
    t = tables.oracle_tables(fx["_locus"])
    A = t["n_alleles"]
    hla = fx["_locus"].base_fname == "hla"
    rows = np.stack([gu.class_bits(fx, p["exon_cls" if hla else "gene_cls"], A) for p in fx["pairs"]])
    ub, uc, fr = orc.dedup(rows)
    em0 = fx["em"][0]
    assert len(ub) == len(em0["cmpt"])
    for k, (cid, cnt) in enumerate(em0["cmpt"]):
        assert np.array_equal(ub[k], gu.class_bits(fx, cid, A)) and uc[k] == cnt
    names = t["names"]
    lengths = np.array([fx["_locus"].allele_length(n) for n in names], dtype=np.int32)
    for em in fx["em"]:
        classes = []
        for cid, _ in em["cmpt"]:
            key = gu.class_key(fx, cid)
            classes.append([t["aidx"][n] for n in key.split("-")])
        oa, op, it = orc.single_abundance(A, classes, [c for _, c in em["cmpt"]], em["remove_low"],
                                          lengths if em["use_length"] else None)
        assert it == em["n_iter"]
        assert [[names[a], repr(float(p))] for a, p in zip(oa, op)] == em["result"]


def test_filtered_reaccumulation(orc):
    """Gene_cmpt2 (typing_core.py:1752-1766) = dedup of the gene classes AND exon_alleles."""
    fx = gu.load("hla_mid_real")
    t = tables.oracle_tables(fx["_locus"])
    A = t["n_alleles"]
    assert len(fx["em"]) == 2
    rows = np.stack([gu.class_bits(fx, p["gene_cls"], A) for p in fx["pairs"]])
    ub, uc, _ = orc.dedup(rows)
    keep = set()
    for cid, _ in fx["em"][1]["cmpt"]:
        keep |= set(_bits_to_indices(gu.class_bits(fx, cid, A)))
    # exon_alleles is a superset of what survives in the classes; rebuild it the reference's way
    names = t["names"]
    res0 = fx["em"][0]["result"]
    exon_alleles = set()
    for i, (a, p) in enumerate(res0):
        if i >= 10 and float(p) < 0.03:
            break
        g = t["rep_groups"][a]
        if len(g) <= 1:
            continue
        exon_alleles |= {t["aidx"][x] for x in g}
    mask = np.zeros((A + 63) // 64, dtype=np.uint64)
    for a in exon_alleles:
        mask[a >> 6] |= np.uint64(1) << np.uint64(a & 63)
    ub2, uc2, _ = orc.dedup(ub, weight=uc, and_mask=mask)
    exp = fx["em"][1]["cmpt"]
    assert len(ub2) == len(exp)
    for k, (cid, cnt) in enumerate(exp):
        assert np.array_equal(ub2[k], gu.class_bits(fx, cid, A)) and uc2[k] == cnt


def test_oracle_chain_on_the_10k_read_reference_run():
    """BASELINE configs[0] at full size (fixture `hla_7000_10k`: the real reference on 10 k reads, 7 000 alleles): the oracle
    chain the config-sized GPU tests lean on -- pyref front-end feeding the C oracle's add_count / add_stat / dedup /
    single_abundance / hand-off (tests/oracle_util.py) -- reproduces the reference end to end: read and pair counts, the class
    dicts going into both EM calls (rows, counts, order), bit-identical abundances, and every (allele, count) line of the
    report in the reference's order."""
    import oracle_util as ou
    fx = gu.load("hla_7000_10k")
    loc = fx["_locus"]
    out = ou.oracle_type(loc.to_json(), fx["sam"])
    head = [l for l in fx["report"].split("\n") if "aligned" in l][0]
    assert head.strip() == "%d reads and %d pairs are aligned" % (out["num_reads"], out["num_pairs"])
    names = [n for n in loc.allele_names if "BACKBONE" not in n]
    A = len(names)
    assert [(c, it) for c, it, _ in out["em"]] == [(len(e["cmpt"]), e["n_iter"]) for e in fx["em"]]
    for (c, it, res), e in zip(out["em"], fx["em"]):
        assert [[a, repr(float(p))] for a, p in res] == e["result"]
    eb, ec = out["exon_classes"]
    for k, (cid, n) in enumerate(fx["em"][0]["cmpt"]):
        assert np.array_equal(eb[k], gu.class_bits(fx, cid, A)) and ec[k] == n
    gc, fp = out["gene_counts"], out["first_pair"]
    order = sorted([a for a in range(A) if gc[a] > 0], key=lambda a: (fp[a], a))
    order = sorted(order, key=lambda a: -gc[a])
    exp = []
    for l in fx["report"].split("\n"):
        if "(count:" in l:
            f = l.strip().split()
            exp.append((f[1], int(f[3].rstrip(")"))))
    assert [(names[a], int(gc[a])) for a in order] == exp and len(exp) > 1000
    assert fx["reference_timing"]["sam_records"] == 10000 and fx["reference_timing"]["seconds"] > 10


def test_oracle_chain_on_the_codis_10k_reference_run():
    """BASELINE configs[4]'s shape through the real reference (fixture `codis_10k`: one CODIS STR ladder, 10 k reads, timed): the
    same oracle chain reproduces the reference's counts, its single EM call bit for bit, and the (allele, count) report lines."""
    import oracle_util as ou
    fx = gu.load("codis_10k")
    loc = fx["_locus"]
    out = ou.oracle_type(loc.to_json(), fx["sam"])
    head = [l for l in fx["report"].split("\n") if "aligned" in l][0]
    assert head.strip() == "%d reads and %d pairs are aligned" % (out["num_reads"], out["num_pairs"])
    names = [n for n in loc.allele_names if "BACKBONE" not in n]
    assert [(c, it) for c, it, _ in out["em"]] == [(len(e["cmpt"]), e["n_iter"]) for e in fx["em"]]
    for (c, it, res), e in zip(out["em"], fx["em"]):
        assert [[a, repr(float(p))] for a, p in res] == e["result"]
    gc, fp = out["gene_counts"], out["first_pair"]
    order = sorted([a for a in range(len(names)) if gc[a] > 0], key=lambda a: (fp[a], a))
    order = sorted(order, key=lambda a: -gc[a])
    exp = []
    for l in fx["report"].split("\n"):
        if "(count:" in l:
            f = l.strip().split()
            exp.append((f[1], int(f[3].rstrip(")"))))
    assert [(names[a], int(gc[a])) for a in order] == exp and len(exp) == 13
    assert fx["reference_timing"]["sam_records"] == 10000
