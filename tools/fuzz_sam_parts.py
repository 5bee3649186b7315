"""Randomised check of the two-part SAM call (records_split of csrc/hgx_front.hip: the line table and record fields of the phases that
have landed run beside the last phase's transfer) on the GPU box: SAM files of 70-140 MB -- a random locus, depth, error rate, CRLF line
ends now and then, a last line with or without its newline, a region or none, the read groups in name order / rotated / with a few
groups swapped (not in name order: the call must fall back to the whole text and its sort) -- through hgx_parse_alignment_file_dev
by default and with front=sam_whole, and through the host reader + host front end: three batches, byte for byte.
usage: tools/fuzz_sam_parts.py [n_cases] [first_seed]"""
import os, sys, random, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hisatgenotype_amd  # noqa
from hisatgenotype_amd import engine, locus as hl, synth

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 880000
d = tempfile.mkdtemp(prefix="hgx_fuzz_parts_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
path = os.path.join(d, "c.sam")
bad = 0
took = {}
t0 = time.time()
for k in range(n_cases):
    rng = random.Random(seed0 + k)
    loc = synth.make_hla_like_locus(n_alleles=rng.randint(100, 900), n_vars=rng.randint(150, 700), seed=seed0 + k)
    pl = hl.PackedLocus.from_synth(loc)
    sam = synth.simulate_sam_fast(loc, synth.pick_sample(loc, seed0 + k), rng.randint(110000, 200000), err_rate=rng.choice([0.0, 0.003, 0.01]), seed=k)
    lines = sam.split("\n")[:-1]
    del sam
    groups = []
    for l in lines:
        q = l[:l.index("\t")]
        if groups and groups[-1][0] == q:
            groups[-1][1].append(l)
        else:
            groups.append((q, [l]))
    order = rng.choice(["sorted", "sorted", "rotated", "swapped"])
    if order == "rotated":
        c = rng.randrange(1, len(groups))
        groups = groups[c:] + groups[:c]
    elif order == "swapped":
        for _ in range(rng.randint(1, 4)):
            a, b = rng.randrange(len(groups)), rng.randrange(len(groups))
            groups[a], groups[b] = groups[b], groups[a]
    nl = rng.choice(["\n", "\n", "\r\n"])
    text = nl.join(l for _, ls in groups for l in ls) + (nl if rng.random() < 0.7 else "")
    if len(text) < (65 << 20):
        continue
    with open(path, "w", newline="") as f:
        f.write(text)
    del text, lines, groups
    regions = [loc.ref_allele] if rng.random() < 0.4 else None
    host = pl.parse_alignment_file(path, regions)
    with engine.test_switches(front="sam_whole"):
        whole = pl.parse_alignment_file_dev(path, regions=regions).to_host()
    parts_b = pl.parse_alignment_file_dev(path, regions=regions)
    n_parts, route = engine.front_last_parts(), engine.front_last()
    parts = parts_b.to_host()
    took[n_parts] = took.get(n_parts, 0) + 1
    ok = route == (2, 0) and (n_parts >= 2) == (order == "sorted")
    for other in (whole, parts):
        ok = ok and all(getattr(other, f).tobytes() == getattr(host, f).tobytes() for f in ("pieces", "masks", "pair_off", "pair_ref")) and other.n_reads == host.n_reads
    if not ok:
        bad += 1
        print("case %d seed %d (%s, region %s): MISMATCH (route %s, parts %d)" % (k, seed0 + k, order, bool(regions), route, n_parts), flush=True)
    pl.close()
    if (k + 1) % 10 == 0:
        print("%d cases, %d mismatches, %.0f s" % (k + 1, bad, time.time() - t0), flush=True)
os.remove(path) if os.path.exists(path) else None
print("%d cases, %d mismatches; calls by parts taken: %s; %.0f s" % (n_cases, bad, took, time.time() - t0))
