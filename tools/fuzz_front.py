"""Randomised parity of the DEVICE front end on the GPU box: the batch the kernels build in HBM (record route and key route) against
the pinned host front end's batch, byte for byte, on the cases of tools/fuzz_parity.py (HLA-like and STR loci, errors, soft clips,
novel indels, duplicates, multi-hit and single-end records).  With `bam`: every case also as a coordinate-sorted BAM file through
hgx_parse_alignment_file_dev (BGZF inflate, record walk, region filter, name sort on the device too) against the host reader + host
front end.  usage: tools/fuzz_front.py [n_cases] [first_seed] [bam]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import hisatgenotype_amd  # noqa
from hisatgenotype_amd import capi, engine, locus as hl
import fuzz_parity

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 990000
with_bam = len(sys.argv) > 3 and sys.argv[3] == "bam"
if with_bam:
    import tempfile
    from hisatgenotype_amd import bamio
    tmp = tempfile.mkdtemp(prefix="hgx_fuzz_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
bam_routes = {}
bad = 0
routes = {}
t0 = time.time()
for k in range(n_cases):
    loc, sam, single = fuzz_parity.make_case(seed0, k, 1 + k % 4)
    pl = hl.PackedLocus.from_synth(loc)
    for ec in (True, False):
        try:
            host = pl.parse_sam(sam, error_correction=ec, allow_discordant=single)
        except capi.HgxError as e:
            for extra in ("device", "device,keys"):
                try:
                    with engine.test_switches(front=extra):
                        pl.parse_sam_dev(sam, error_correction=ec, allow_discordant=single)
                    print("case %d seed %d: host raised %r, device did not" % (k, seed0 + k, e)); bad += 1
                except capi.HgxError:
                    pass
            continue
        L = len(loc.backbone)
        for extra in ("device", "device,keys"):
            with engine.test_switches(front=extra):
                dev = pl.parse_sam_dev(sam, error_correction=ec, allow_discordant=single)
                route, code = engine.front_last()
            routes[(route, code)] = routes.get((route, code), 0) + 1
            hb = dev.to_host()
            same = all(getattr(hb, f).tobytes() == getattr(host, f).tobytes() for f in ("pieces", "masks", "pair_off", "pair_ref")) and \
                hb.n_reads == host.n_reads
            if same and route > 0:
                (na, ca), (nb, cb) = hb.pileup(L), host.pileup(L)
                same = np.array_equal(na, nb) and np.array_equal(ca, cb)
            if not same:
                print("case %d seed %d ec %s %s: MISMATCH (route %d, code %d)" % (k, seed0 + k, ec, extra, route, code)); bad += 1
    if with_bam:
        path = os.path.join(tmp, "c.bam")
        bamio.write_bam_native(path, sam.encode(), [(loc.ref_allele, len(loc.backbone))], sort_by_coordinate=(k % 2 == 0))
        regions = [loc.ref_allele] if k % 3 else None
        try:
            hostb = pl.parse_alignment_file(path, regions, allow_discordant=single)
        except capi.HgxError:
            hostb = None
        for sw in ("device", "device,host_inflate"):
            try:
                with engine.test_switches(front=sw):
                    devb = pl.parse_alignment_file_dev(path, regions=regions, allow_discordant=single)
                    route, code = engine.front_last()
            except capi.HgxError:
                if hostb is not None:
                    print("case %d seed %d BAM %s: device raised, host did not" % (k, seed0 + k, sw)); bad += 1
                continue
            bam_routes[(sw, route, code)] = bam_routes.get((sw, route, code), 0) + 1
            hb = devb.to_host()
            if hostb is None or not (all(getattr(hb, f).tobytes() == getattr(hostb, f).tobytes() for f in ("pieces", "masks", "pair_off", "pair_ref")) and
                                     hb.n_reads == hostb.n_reads):
                print("case %d seed %d BAM %s: MISMATCH (route %d, code %d)" % (k, seed0 + k, sw, route, code)); bad += 1
    pl.close()
    if (k + 1) % 100 == 0:
        print("%d cases, %d mismatches, %.0f s" % (k + 1, bad, time.time() - t0), flush=True)
print("%d cases (x 2 error-correction settings x 2 routes), %d mismatches; (route, decline code) -> inputs: %s; %.0f s" % (
    n_cases, bad, dict(sorted(routes.items())), time.time() - t0))
if with_bam:
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    print("BAM files: (switches, route, decline code) -> inputs: %s" % dict(sorted(bam_routes.items())))
sys.exit(1 if bad else 0)
