#!/usr/bin/env python3
"""hgx_many_create_files for K copies of a 64-sample locus side by side (a host thread and a stream per locus, as bench.py's panel
leg does) against one after the other; HGX_PARSE_PROFILE=1 for the last side-by-side round.  Usage: tools/prof_many_side.py [K] [n_samples]"""
import os, sys, time, tempfile, shutil
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hisatgenotype_amd import bamio, capi, engine, locus as hl, synth
K, n = int(sys.argv[1]) if len(sys.argv) > 1 else 6, int(sys.argv[2]) if len(sys.argv) > 2 else 64
thr = int(os.environ.get("THR", "10"))
loc = synth.make_hla_like_locus(gene="A", n_alleles=7000, length=3500, n_vars=2500, seed=500)
pl = hl.PackedLocus.from_synth(loc)
pl.index()
d = tempfile.mkdtemp(prefix="hgx_pms_", dir="/dev/shm")
try:
    paths = []
    for s in range(n):
        sam = synth.simulate_sam_fast(loc, synth.pick_sample(loc, 1000 * s), 5000, err_rate=0.002, seed=100 * s)
        paths.append(os.path.join(d, "s%02d.bam" % s))
        bamio.write_bam_native(paths[-1], sam.encode(), [(loc.ref_allele, len(loc.backbone))], sort_by_coordinate=True)
    def one(k):
        capi.set_stream_slot(("side", k))
        t0 = time.perf_counter()
        m = engine.ManyBatch.from_files(pl, paths, regions=[loc.ref_allele] * n, n_threads=thr, stream=capi.get_stream(1))
        dt = time.perf_counter() - t0
        m.close()
        return dt
    for rep in range(5):
        if rep == 4:
            os.environ["HGX_PARSE_PROFILE"] = "1"
        t0 = time.perf_counter()
        with ThreadPoolExecutor(K) as ex:
            dts = list(ex.map(one, range(K)))
        t1 = time.perf_counter()
        print("side by side: %.1f ms  (per call %s)" % ((t1 - t0) * 1e3, " ".join("%.1f" % (x * 1e3) for x in dts)), flush=True)
    os.environ.pop("HGX_PARSE_PROFILE", None)
    t0 = time.perf_counter()
    dts = [one(k) for k in range(K)]
    print("one after the other: %.1f ms  (per call %s)" % ((time.perf_counter() - t0) * 1e3, " ".join("%.1f" % (x * 1e3) for x in dts)))
finally:
    shutil.rmtree(d, ignore_errors=True)
