"""Randomised parity of the many-task path on the GPU box: for freshly seeded loci, several samples of one locus (different
individuals, read counts from 0 to a few hundred pairs, errors / indels / soft clips / duplicates as tools/fuzz_parity.py draws
them) are typed TOGETHER (hgx_type_many: one launch chain, EMs of all tasks in one k_emx launch) and compared
  (1) with the one-task path on every task: `==` on counts, EM lists (abundances, orders, iteration counts) and Gene_prob;
  (2) every `oracle_every`-th case also with oracle/pyref.py (the pinned restatement of the reference) on every task: `==` again --
      both paths run the EM in the reference's own order of operations.
A task on which the reference would raise must fail alone (its neighbours keep their results).
Usage: tools/fuzz_many.py [n_cases] [first_seed] [oracle_every]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import random
import hisatgenotype_amd as hgx
from hisatgenotype_amd import synth, locus as hl, engine
import pyref

htyping = sys.modules["hisatgenotype_amd.typing"]


def make_case(seed):
    rng = random.Random(seed)
    str_locus = rng.random() < 0.2
    if str_locus:
        loc = synth.make_str_like_locus(gene=rng.choice(["D8S1179", "D18S51"]), unit=rng.choice(["TCTA", "AGAA"]), seed=seed,
                                        max_repeats=rng.randint(8, 18), min_repeats=rng.randint(3, 6))
    else:
        loc = synth.make_hla_like_locus(n_alleles=rng.randint(30, 1500), n_vars=rng.randint(60, 900), seed=seed,
                                        insertion_frac=rng.choice([0.0, 0.03]), unlinked_vars=rng.randint(0, 4))
    sams = []
    for t in range(rng.randint(2, 7)):
        sample = synth.pick_sample(loc, seed * 31 + t)
        n = rng.choice([0, 1, 3, rng.randint(20, 80), rng.randint(60, 260), rng.randint(60, 260)])
        if n == 0:
            sams.append("")
            continue
        if str_locus:
            al = synth.simulate_pairs(loc, sample, n, read_len=100, frag_len=(250, 250), seed=seed + 7 * t, err_rate=rng.choice([0.0, 0.002]))
        else:
            al = synth.simulate_pairs(loc, sample, n, err_rate=rng.choice([0.0, 0.003, 0.01]), seed=seed + 7 * t,
                                      softclip_frac=rng.choice([0.0, 0.05]), novel_del_frac=rng.choice([0.0, 0.03]),
                                      multi_hit_frac=rng.choice([0.0, 0.02]), dup_frac=rng.choice([0.0, 0.02]),
                                      novel_ins_frac=rng.choice([0.0, 0.02]))
        sams.append(synth.sam_text(loc, al))
    return loc, sams


def summary(res):
    """what is compared, as plain Python values"""
    if isinstance(res, BaseException):
        return ("error",)                       # (which exception: pyref reports every failure of the reference as ReferenceError_)
    if isinstance(res, dict):                                  # pyref
        if res["num_reads"] == 0:
            return (0, 0)
        return (res["num_reads"], res["num_pairs"], res["counts_sorted"],
                [(e["n_iter"], [(a, p) for a, p in e["result"]]) for e in res["em"]], [(a, p) for a, p in res["gene_prob"]])
    if res.num_reads == 0:
        return (0, 0)
    return (res.num_reads, res.num_pairs, res.counts_sorted,
            [(e["n_iter"], [(a, p) for a, p in e["result"]]) for e in res.em], [(a, p) for a, p in res.gene_prob])


def first_difference(x, y):
    if type(x) != type(y) and not (isinstance(x, (list, tuple)) and isinstance(y, (list, tuple))):
        return "%r vs %r" % (x, y)
    if isinstance(x, (list, tuple)):
        if len(x) != len(y):
            return "length %d vs %d" % (len(x), len(y))
        for i, (a, b) in enumerate(zip(x, y)):
            d = first_difference(a, b)
            if d:
                return "[%d] %s" % (i, d)
        return None
    return None if x == y else "%r vs %r" % (x, y)


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 800000
    oracle_every = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    bad = n_tasks = n_oracle = n_err = 0
    t0 = time.time()
    for k in range(n_cases):
        loc, sams = make_case(seed0 + k)
        pl = hl.PackedLocus.from_synth(loc)
        batches = [pl.parse_sam(s) for s in sams]
        many = engine.ManyBatch(pl, batches)
        got = htyping.type_many(pl, many, return_errors=True, em_fast=False)
        why = []
        for t, (sam, b) in enumerate(zip(sams, batches)):
            try:
                one = htyping.type_locus(pl, sam) if sam else None
            except Exception as e:
                one = e
            s_many = summary(got[t])
            s_one = (0, 0) if one is None else summary(one)
            if s_many[0] == "error" or s_one[0] == "error":
                n_err += 1
            d = first_difference(s_many, s_one)
            if d:
                why.append("task %d many vs one-task: %s" % (t, d[:160]))
            if oracle_every and k % oracle_every == 0:
                try:
                    exp = pyref.RefLocus(loc).run(sam) if sam else {"num_reads": 0}
                except Exception as e:
                    exp = e
                n_oracle += 1
                d = first_difference(s_many, summary(exp))
                if d:
                    why.append("task %d many vs pyref: %s" % (t, d[:160]))
        n_tasks += len(sams)
        print("case %4d seed %d %-5s A=%-5d tasks=%d reads=%s %s" % (
            k, seed0 + k, loc.base_fname, len(loc.allele_names) - 1, len(sams), [b.n_reads for b in batches],
            "ok" if not why else "MISMATCH " + "; ".join(why)), flush=True)
        bad += 1 if why else 0
        many.close()
        pl.close()
    print("%d cases, %d tasks typed together and one by one (== on counts, every EM list, iteration counts, Gene_prob), %d of them also == "
          "oracle/pyref.py, %d tasks on which the reference raises; %d mismatching cases; %.0f s" % (n_cases, n_tasks, n_oracle, n_err, bad, time.time() - t0))
    sys.exit(1 if bad else 0)
