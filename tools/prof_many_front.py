#!/usr/bin/env python3
"""Where the time of hgx_many_create_files goes: 64 samples x 5000 pairs of a 7000-allele locus as BAM files, one locus at a time
(HGX_PARSE_PROFILE=1 prints the stages).  Usage: tools/prof_many_front.py [n_samples] [pairs]"""
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hisatgenotype_amd import bamio, engine, locus as hl, synth
if os.environ.get("HGX_BENCH_LIB"):          # an A/B against another build of libhgx
    from hisatgenotype_amd import capi as _capi
    _capi.LIB_PATH = os.environ["HGX_BENCH_LIB"]
n, pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[2]) if len(sys.argv) > 2 else 5000
loc = synth.make_hla_like_locus(gene="A", n_alleles=7000, length=3500, n_vars=2500, seed=500)
pl = hl.PackedLocus.from_synth(loc)
pl.index()
d = tempfile.mkdtemp(prefix="hgx_pmf_", dir="/dev/shm")
try:
    paths = []
    for s in range(n):
        sam = synth.simulate_sam_fast(loc, synth.pick_sample(loc, 1000 * s), pairs, err_rate=0.002, seed=100 * s)
        paths.append(os.path.join(d, "s%02d.bam" % s))
        bamio.write_bam_native(paths[-1], sam.encode(), [(loc.ref_allele, len(loc.backbone))], sort_by_coordinate=True)
    for rep in range(4):
        if rep == 3:
            os.environ["HGX_PARSE_PROFILE"] = "1"
        t0 = time.perf_counter()
        m = engine.ManyBatch.from_files(pl, paths, regions=[loc.ref_allele] * n)
        t1 = time.perf_counter()
        print("from_files %.1f ms  route %s  bytes %d  pairs %d pieces %d" % ((t1 - t0) * 1e3, engine.front_last(), engine.front_last_bytes(), m.n_pairs, m.n_pieces), flush=True)
        m.close()
finally:
    shutil.rmtree(d, ignore_errors=True)
