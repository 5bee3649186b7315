"""Randomised parity of hgx_alignment (ONE file read once, a locus at a time by region; round 6): 2-5 fuzz loci of tools/fuzz_parity.py
(HLA-like and STR loci, errors, soft clips, novel indels, duplicates, multi-hit and single-end records) concatenated into one
multi-reference file -- SAM text, a name-grouped BAM or a coordinate-sorted BAM in turn --, every locus' batch out of the resident
file (engine.Alignment.parse_dev, forced resident: the cases are small) against the pinned HOST reader + host front end on the same
file and region, byte for byte.  usage: tools/fuzz_alignment.py [n_files] [first_seed]"""
import os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import hisatgenotype_amd  # noqa
from hisatgenotype_amd import bamio, capi, engine, locus as hl
import fuzz_parity

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 660000
tmp = tempfile.mkdtemp(prefix="hgx_fuzz_al_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
bad, n_loci, routes = 0, 0, {}
t0 = time.time()
capi.set_device(0)
for k in range(n_files):
    n = 2 + k % 4
    cases = [fuzz_parity.make_case(seed0 + 1000 * k, j, 1 + (k + j) % 4) for j in range(n)]
    lines, refs = [], []
    for j, (loc, sam, single) in enumerate(cases):
        rname = "F%d_%s" % (j, loc.ref_allele)
        refs.append((rname, len(loc.backbone)))
        for l in sam.split("\n"):
            if l and not l.startswith("@"):
                f = l.split("\t")
                f[2] = rname
                lines.append("\t".join(f))
    text = "\n".join(lines) + "\n"
    kind = k % 3
    path = os.path.join(tmp, "f.sam" if kind == 0 else "f.bam")
    if kind == 0:
        with open(path, "w") as f:
            f.write("".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs) + text)
    else:
        bamio.write_bam_native(path, text.encode(), refs, sort_by_coordinate=(kind == 2))
    with engine.test_switches(front="device"):
        al = engine.Alignment(path)
        assert al.resident
        for j, (loc, sam, single) in enumerate(cases):
            pl = hl.PackedLocus.from_synth(loc)
            n_loci += 1
            try:
                host = pl.parse_alignment_file(path, [refs[j][0]], allow_discordant=single)
            except capi.HgxError:
                host = None
            try:
                dev = al.parse_dev(pl, [refs[j][0]], allow_discordant=single)
                route, code = engine.front_last()
            except capi.HgxError:
                if host is not None:
                    print("file %d locus %d: the resident file raised, the host reader did not" % (k, j)); bad += 1
                pl.close()
                continue
            routes[(kind, route, code)] = routes.get((kind, route, code), 0) + 1
            hb = dev.to_host()
            if host is None or not (all(getattr(hb, f).tobytes() == getattr(host, f).tobytes() for f in ("pieces", "masks", "pair_off", "pair_ref")) and
                                    hb.n_reads == host.n_reads):
                print("file %d (kind %d) locus %d seed %d: MISMATCH (route %d, code %d)" % (k, kind, j, seed0 + 1000 * k + j, route, code)); bad += 1
            pl.close()
        al.close()
    if (k + 1) % 50 == 0:
        print("%d files, %d loci, %d mismatches, %.0f s" % (k + 1, n_loci, bad, time.time() - t0), flush=True)
shutil.rmtree(tmp, ignore_errors=True)
print("%d multi-locus files (SAM text / BAM / coordinate-sorted BAM in turn), %d loci, %d mismatches; (kind, route, decline code) -> loci: %s; %.0f s" % (
    n_files, n_loci, bad, dict(sorted(routes.items())), time.time() - t0))
sys.exit(1 if bad else 0)
