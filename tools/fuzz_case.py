"""Details of one tools/fuzz_parity.py case: fuzz_case.py <first_seed> <case index>"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import hisatgenotype_amd as hgx
from hisatgenotype_amd import synth, locus as hl
import pyref
seed0, k = int(sys.argv[1]), int(sys.argv[2])
scale = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rng = random.Random(seed0 + k)
if rng.random() < 0.25:
    loc = synth.make_str_like_locus(gene=rng.choice(["D8S1179", "D18S51"]), unit=rng.choice(["TCTA", "AGAA"]), seed=seed0 + k,
                                    max_repeats=rng.randint(8, 18), min_repeats=rng.randint(3, 6))
    sample = synth.pick_sample(loc, seed0 + k)
    al = synth.simulate_pairs(loc, sample, scale * rng.randint(40, 160), read_len=100, frag_len=(250, 250), seed=k, err_rate=rng.choice([0.0, 0.002]))
else:
    loc = synth.make_hla_like_locus(n_alleles=rng.randint(30, 1200), n_vars=rng.randint(60, 900), seed=seed0 + k,
                                    insertion_frac=rng.choice([0.0, 0.03]), unlinked_vars=rng.randint(0, 4))
    sample = synth.pick_sample(loc, seed0 + k)
    al = synth.simulate_pairs(loc, sample, scale * rng.randint(60, 220), err_rate=rng.choice([0.0, 0.003, 0.01]), seed=k,
                              softclip_frac=rng.choice([0.0, 0.05]), novel_del_frac=rng.choice([0.0, 0.03]),
                              multi_hit_frac=rng.choice([0.0, 0.02]), dup_frac=rng.choice([0.0, 0.02]),
                              novel_ins_frac=rng.choice([0.0, 0.02]), single_end=rng.random() < 0.15)
sam = synth.sam_text(loc, al)
single = any(a.flag & 1 == 0 for a in al)
pl = hl.PackedLocus.from_synth(loc)
exp = pyref.RefLocus(loc, allow_discordant=single).run(sam)
res = hgx.type_locus(pl, sam, allow_discordant=single)
print("final:")
for i, ((a, p), (b, q)) in enumerate(zip(res.gene_prob, exp["gene_prob"])):
    print("  %2d %-18s %-22r | %-18s %-22r %s" % (i, a, p, b, q, "" if a == b else "<<<"))
for n, (g, e) in enumerate(zip(res.em, exp["em"])):
    print("EM", n, "iters", g["n_iter"], e["n_iter"], "classes", g["n_classes"])
    for i, ((a, p), (b, q)) in enumerate(zip(g["result"], e["result"])):
        if i < 14 or a != b:
            print("  %2d %-18s %-22r | %-18s %-22r %s" % (i, a, p, b, q, "" if a == b else "<<<"))
