"""Large-problem EM fuzz on the GPU box: HLA-like loci of 3 000 - 8 000 alleles, 20 000 - 40 000 read pairs,
samples of 2 - 4 alleles with up to 1 % errors, pruning on / off -- problems of ~1 000 - 3 000 exon-level classes over thousands
of alleles, i.e. what tools/fuzz_parity.py's small cases never reach.  Front-end, scoring and dedup run on the GPU path (their
parity at size is pinned by the test suite); the class sets it produced go to the C oracle's single_abundance (the reference's
own order of operations, pinned to the real reference) on the host cores, and every EM result of the GPU path -- EM #1 AND the
hand-off EM #2, allele order, abundances as doubles, iteration counts -- must be EQUAL (`==`).  The same cases again with
hgx_type_opts.em_fast: same iteration counts and survivors, abundances within 1e-9.
Usage: tools/fuzz_em_large.py [n_cases] [first_seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import random
from concurrent.futures import ProcessPoolExecutor
import numpy as np


def oracle_em(n_alleles, names, bits, counts, remove_low, lengths):
    import orclib, orc_pipeline
    orc = orclib.load()
    t = {"names": names, "n_alleles": n_alleles}
    res, it = orc_pipeline.em_sorted(orc, t, bits, counts, remove_low, lengths)
    return res, it


def main():
    import hisatgenotype_amd as hgx
    from hisatgenotype_amd import synth, locus as hl
    htyping = sys.modules["hisatgenotype_amd.typing"]
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 500000
    any_size = len(sys.argv) > 3 and sys.argv[3] == "any"     # em_fast = -1: the reference's order at every size; 30-60 k pairs per case, so
                                                             # that most exon-level EMs have more than 4096 classes (k_emx in cluster mode)
    pool = ProcessPoolExecutor(max_workers=max(1, min(14, (os.cpu_count() or 2) - 2)))
    pending = []
    bad = fast_bad = 0
    n_exact = n_big = 0
    fast_order = [0]
    iters_hist = []
    worst_fast = 0.0
    t0 = time.time()

    def check(item):
        nonlocal bad, fast_bad, worst_fast, n_exact, n_big
        k, seed, A, n_pairs, low, res, fast, futs = item
        ok, why = True, ""
        for e, ef, f in zip(res.em, fast.em, futs):
            want, it = f.result()
            names_p = [(res._names[a], p) for a, p in want]
            got = [(a, p) for a, p in e["result"]]
            if e["n_classes"] > 4096:
                n_big += 1
                if not any_size:                            # beyond k_emx's default gate: the table-lookup path, within 1e-9 (not compared here)
                    continue
            n_exact += 1
            if it != e["n_iter"]:
                ok, why = False, "iterations %d vs %d" % (e["n_iter"], it)
            elif got != names_p:
                dev = max([abs(p - q) for (_, p), (_, q) in zip(got, names_p)] or [0.0])
                ok, why = False, "EM result differs (same alleles: %s, max dev %.2e)" % ([a for a, _ in got] == [a for a, _ in names_p], dev)
            if ef["n_iter"] != e["n_iter"]:
                fast_bad += 1
                why += " [fast mode: %d iterations]" % ef["n_iter"]
            elif [a for a, _ in ef["result"]] != [a for a, _ in e["result"]]:
                # the same alleles in another order (with pruning off the list holds thousands of alleles at ~0: their order is
                # decided by the last bits): counted, and the abundances compared allele by allele
                fast_order[0] += 1
                pe = dict(e["result"])
                if sorted(pe) != sorted(a for a, _ in ef["result"]):
                    fast_bad += 1
                    why += " [fast mode: other survivors]"
                else:
                    worst_fast = max(worst_fast, max(abs(p - pe[a]) for a, p in ef["result"]))
            else:
                worst_fast = max(worst_fast, max([abs(p - q) for (_, p), (_, q) in zip(ef["result"], e["result"])] or [0.0]))
        iters_hist.append(max(e["n_iter"] for e in res.em))
        bad += 0 if ok else 1
        print("case %4d seed %d A=%d pairs=%d low=%d classes=%s iters=%s %s %s" % (
            k, seed, A, n_pairs, low, [e["n_classes"] for e in res.em], [e["n_iter"] for e in res.em], "ok" if ok else "MISMATCH", why), flush=True)

    only = [int(x) for x in os.environ.get("FUZZ_ONLY", "").split(",") if x]      # replay single cases of a batch
    for k in (only or range(n_cases)):
        seed = seed0 + k
        rng = random.Random(seed)
        A = rng.randint(3000, 8000)
        loc = synth.make_hla_like_locus(n_alleles=A, n_vars=rng.randint(1500, 2800), seed=seed, sibling_frac=rng.choice([0.3, 0.5]))
        # two alleles, or a mixture (contaminated / pooled sample) with skewed proportions: the long EMs
        n_al = rng.choice([2, 2, 3, 4, 6, 8])
        sample = synth.pick_sample(loc, seed, n=n_al)
        if n_al > 2 and rng.random() < 0.7:
            sample = sample + [sample[0]] * rng.randint(1, 6) + [sample[1]] * rng.randint(0, 3)
        n_pairs = rng.randint(30000, 60000) if any_size else rng.randint(20000, 30000)    # (more pairs -> more than 4096 exon-level classes)
        sam = synth.simulate_sam_fast(loc, sample, n_pairs, err_rate=rng.choice([0.0, 0.002, 0.005, 0.01]), seed=k)
        low = rng.random() < 0.5
        pl = hl.PackedLocus.from_synth(loc)
        batch = pl.parse_sam(sam)
        if any_size:
            res = htyping.LocusResult()
            res.num_reads, res.num_pairs = batch.n_reads, batch.n_pairs
            res = htyping._type_batch(pl, batch, res, low, keep_classes=True, em_fast=-1)
        else:
            res = hgx.type_locus(pl, sam, remove_low_abundance_alleles=low, keep_classes=True)
        r2 = htyping.LocusResult()
        r2.num_reads, r2.num_pairs = batch.n_reads, batch.n_pairs
        fast = htyping._type_batch(pl, batch, r2, low, em_fast=True)
        names = [n for n in loc.allele_names if "BACKBONE" not in n]
        lengths = np.array([loc.allele_length(n) for n in names], np.int32)
        futs = []
        ebits, ecnt = res.exon_classes
        futs.append(pool.submit(oracle_em, len(names), names, ebits, ecnt, low, None))
        if len(res.em) > 1:
            # Gene_cmpt2 (core:1752-1766): gene classes filtered to exon_alleles, merged in first-seen order -- rebuilt here from the kept
            # gene class set with numpy, then the oracle's EM with lengths and pruning
            gbits, gcnt = res.gene_classes
            e1 = res.em[0]["result"]
            groups = pl.rep_groups()
            keep = set()
            for i, (a, p) in enumerate(e1):
                if i >= 10 and p < 0.03:
                    break
                g = groups.get(pl.aidx[a], [pl.aidx[a]])
                if len(g) > 1:
                    keep |= set(int(x) for x in g)
            mask = np.zeros(gbits.shape[1], np.uint64)
            for j in keep:
                mask[j >> 6] |= np.uint64(1) << np.uint64(j & 63)
            fb = gbits & mask
            order, merged = {}, []
            for row, c in zip(fb, gcnt):
                if not row.any():
                    continue
                key = row.tobytes()
                if key in order:
                    merged[order[key]][1] += int(c)
                else:
                    order[key] = len(merged)
                    merged.append([row, int(c)])
            futs.append(pool.submit(oracle_em, len(names), names, np.stack([m[0] for m in merged]), np.array([m[1] for m in merged], np.int64), True, lengths))
        pending.append((k, seed, A, n_pairs, int(low), res, fast, futs))
        pl.close()
        while len(pending) > 12:
            check(pending.pop(0))
    for item in pending:
        check(item)
    long_ = sum(1 for x in iters_hist if x >= 20)
    print("%d large cases%s, %d EM problems compared (== on every abundance, the allele order and the iteration count; %d had > 4096 classes), "
          "%d mismatches; cases with an EM of >= 20 iterations: %d, longest %d; fast mode (hgx_type_opts.em_fast): %d results with another "
          "iteration count or other survivors, %d with near-zero alleles in another order, largest abundance deviation %.2e; %.0f s" % (
              n_cases, " with em_fast = -1 (the reference's order at every size; lone large problems in cluster mode)" if any_size else "",
              n_exact, n_big, bad, long_, max(iters_hist or [0]), fast_bad, fast_order[0], worst_fast, time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
