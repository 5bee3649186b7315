"""Randomised end-to-end parity on the GPU box: type_locus (front-end + HIP path) against oracle/pyref.py (the pinned pure-Python
restatement of the reference) on freshly seeded loci and read sets.  Usage: tools/fuzz_parity.py [n_cases] [first_seed] [read-count scale]   (FUZZ_ONLY=k1,k2 replays cases k1, k2 of that batch)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import random
import hisatgenotype_amd as hgx
from hisatgenotype_amd import synth, locus as hl
import pyref

def make_case(seed0, k, scale=1):
    """Case k of the batch that starts at seed `seed0`: (locus, SAM text, single-end?).  tests/test_gpu_typing.py replays single cases."""
    rng = random.Random(seed0 + k)
    if rng.random() < 0.25:
        loc = synth.make_str_like_locus(gene=rng.choice(["D8S1179", "D18S51"]), unit=rng.choice(["TCTA", "AGAA"]), seed=seed0 + k,
                                        max_repeats=rng.randint(8, 18), min_repeats=rng.randint(3, 6))
        sample = synth.pick_sample(loc, seed0 + k)
        al = synth.simulate_pairs(loc, sample, scale * rng.randint(40, 160), read_len=100, frag_len=(250, 250), seed=k,
                                  err_rate=rng.choice([0.0, 0.002]))
    else:
        loc = synth.make_hla_like_locus(n_alleles=rng.randint(30, 1200), n_vars=rng.randint(60, 900), seed=seed0 + k,
                                        insertion_frac=rng.choice([0.0, 0.03]), unlinked_vars=rng.randint(0, 4))
        sample = synth.pick_sample(loc, seed0 + k)
        al = synth.simulate_pairs(loc, sample, scale * rng.randint(60, 220), err_rate=rng.choice([0.0, 0.003, 0.01]), seed=k,
                                  softclip_frac=rng.choice([0.0, 0.05]), novel_del_frac=rng.choice([0.0, 0.03]),
                                  multi_hit_frac=rng.choice([0.0, 0.02]), dup_frac=rng.choice([0.0, 0.02]),
                                  novel_ins_frac=rng.choice([0.0, 0.02]), single_end=rng.random() < 0.15)
    sam = synth.sam_text(loc, al)
    single = any(a.flag & 1 == 0 for a in al) if al else False
    return loc, sam, single


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    scale = int(sys.argv[3]) if len(sys.argv) > 3 else 1       # multiplies the number of read pairs per case
    bad = 0
    ties = 0
    worst = 0.0
    t0 = time.time()
    only = [int(x) for x in os.environ.get("FUZZ_ONLY", "").split(",") if x]      # replay single cases of a batch: FUZZ_ONLY=2754,9927
    for k in (only or range(n_cases)):
        loc, sam, single = make_case(seed0, k, scale)
        pl = hl.PackedLocus.from_synth(loc)
        try:
            exp = pyref.RefLocus(loc, allow_discordant=single).run(sam)
            e_err = None
        except Exception as e:
            exp, e_err = None, e
        try:
            res = hgx.type_locus(pl, sam, allow_discordant=single)
            g_err = None
        except Exception as e:
            res, g_err = None, e
        ok = True
        why = ""
        if (e_err is None) != (g_err is None):
            ok, why = False, "error mismatch: ref %r gpu %r" % (e_err, g_err)
        elif e_err is None:
            if (res.num_reads, res.num_pairs) != (exp["num_reads"], exp["num_pairs"]):
                ok, why = False, "read counts"
            elif res.num_reads > 0:
                if res.counts_sorted != exp["counts_sorted"]:
                    ok, why = False, "Gene_counts"
                elif [e["n_iter"] for e in res.em] != [e["n_iter"] for e in exp["em"]]:
                    ok, why = False, "EM iterations %s vs %s" % ([e["n_iter"] for e in res.em], [e["n_iter"] for e in exp["em"]])
                elif [a for a, _ in res.gene_prob] != [a for a, _ in exp["gene_prob"]]:
                    # same alleles, and every position that differs sits in a run of reference abundances equal to 1e-11
                    # relative: the reference's own order there is decided by the rounding noise of its summation order
                    ref = exp["gene_prob"]
                    same_set = sorted(a for a, _ in res.gene_prob) == sorted(a for a, _ in ref)
                    noise = same_set and all(
                        a == b or any(c == a and abs(q2 - q) <= 1e-11 * max(abs(q), 1e-300) for c, q2 in ref)
                        for (a, _), (b, q) in zip(res.gene_prob, ref))
                    if noise:
                        why = "(order inside a reference near-tie differs)"
                        ties += 1
                    else:
                        ok, why = False, "allele order"
                else:
                    dev = max([abs(p - q) for (_, p), (_, q) in zip(res.gene_prob, exp["gene_prob"])] or [0.0])
                    worst = max(worst, dev)
                    if dev > 1e-6:                            # north_star: 1e-5; short EMs agree to ~1e-15, a 41-iteration one to 5e-9
                        ok, why = False, "abundances (max deviation %.2e)" % dev
                    elif dev > 1e-9:
                        why = "(max abundance deviation %.1e over %s EM iterations)" % (dev, [e["n_iter"] for e in res.em])
        print("case %3d seed %d %-5s A=%-5d reads=%-4s %s %s" % (k, seed0 + k, loc.base_fname, len(loc.allele_names) - 1,
                                                              res.num_reads if res else "-", "ok" if ok else "MISMATCH", why), flush=True)
        bad += 0 if ok else 1
        pl.close()
    print("%d cases, %d mismatches, %d near-tie order differences, largest abundance deviation %.2e, %.0f s" % (
        n_cases, bad, ties, worst, time.time() - t0))
    sys.exit(1 if bad else 0)

