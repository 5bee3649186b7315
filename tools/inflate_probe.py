#!/usr/bin/env python3
"""k_bgzf_inflate alone: a coordinate-sorted BAM of 1 M reads (configs[1]'s file) and 64 BAMs of 10 000 reads back to back (a panel
locus' files), each through hgx_bgzf_inflate (host bytes in and out: run under rocprofv3 --kernel-trace for the kernel's own time;
`show` prints the launches by grid size).
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/inf -o t -- python3 tools/inflate_probe.py run
  python3 tools/inflate_probe.py show gpurun_out/inf/t_kernel_trace.csv"""
import os, sys, csv, time, zlib, tempfile, shutil, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "show":
    by = {}
    for r in csv.DictReader(open(sys.argv[2])):
        if "k_bgzf_" in r["Kernel_Name"]:
            by.setdefault((r["Kernel_Name"].split("(")[0].split("::")[-1], int(r["Grid_Size_X"]) // 64), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    for (kn, nb), ts in sorted(by.items()):
        print("%-18s %6d BGZF blocks: %d launches, median %.3f ms, min %.3f ms" % (kn, nb, len(ts), sorted(ts)[len(ts) // 2], min(ts)))
    sys.exit(0)
import numpy as np
from hisatgenotype_amd import bamio, capi, synth
if os.environ.get("INF_LIB"):
    capi.LIB_PATH = os.environ["INF_LIB"]
if os.environ.get("INF_FORM"):                     # inflate_v1 = round 4's kernel, inflate_prof = the default kernel with clock64 laps per phase
    from hisatgenotype_amd import engine
    engine.test_switch("front", os.environ["INF_FORM"])
loc = synth.make_hla_like_locus(gene="A", n_alleles=7000, length=3500, n_vars=2500, seed=500)
d = tempfile.mkdtemp(prefix="hgx_inf_", dir="/dev/shm")
try:
    datas = []
    sam = synth.simulate_sam_fast(loc, synth.pick_sample(loc, 1), 500000, err_rate=0.002, seed=1)
    p = os.path.join(d, "big.bam")
    bamio.write_bam_native(p, sam.encode(), [(loc.ref_allele, len(loc.backbone))], sort_by_coordinate=True)
    datas.append(("1 M reads, one file", open(p, "rb").read()))
    many = b""
    for s in range(64):
        sam = synth.simulate_sam_fast(loc, synth.pick_sample(loc, 1000 * s), 5000, err_rate=0.002, seed=100 * s)
        p = os.path.join(d, "s.bam")
        bamio.write_bam_native(p, sam.encode(), [(loc.ref_allele, len(loc.backbone))], sort_by_coordinate=True)
        many += open(p, "rb").read()
    datas.append(("64 files of 10 000 reads", many))
    for name, data in datas:
        nb = data.count(b"\x1f\x8b\x08\x04")
        cap = 64 * 1024 * (nb + 2)
        buf = np.zeros(cap, np.uint8)
        n_out, bad = C.c_size_t(0), C.c_int32(0)
        for rep in range(5):
            t0 = time.perf_counter()
            capi.check(capi.lib().hgx_bgzf_inflate(data, C.c_size_t(len(data)), capi.ptr(buf), C.c_size_t(cap), C.byref(n_out), C.byref(bad), None))
            dt = time.perf_counter() - t0
        assert bad.value == 0
        want = zlib.crc32(b"")
        print("%s: %.1f MB -> %.1f MB, ~%d blocks, call (with the copies) %.1f ms" % (name, len(data) / 1e6, n_out.value / 1e6, nb, dt * 1e3), flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
