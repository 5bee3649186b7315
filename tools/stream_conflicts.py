"""Which of the streams a run with N samples in flight uses share a hardware queue: the callers' main streams (capi.get_stream(2) per
worker slot), and the EM / gene-side streams of libhgx's stream sets.  Runs the configs1 steps (1 in flight, then N in flight) as
bench.py does, then probes every pair."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]] + sys.argv[1:]
import bench  # noqa: E402
from hisatgenotype_amd import capi, engine, synth, locus as hl  # noqa: E402

nf = int(os.environ.get("NF", "2"))
pairs = int(os.environ.get("PAIRS", "500000"))
if os.environ.get("HGX_STREAMS"):
    engine.test_switch("streams", os.environ["HGX_STREAMS"])
capi.set_device(0)
loc = synth.make_hla_like_locus(n_alleles=7000, n_vars=2500, seed=101)
pl = hl.PackedLocus.from_synth(loc)
pl.index()
sam = synth.simulate_sam_fast(loc, synth.pick_sample(loc, 101), pairs, err_rate=0.002, seed=100)
batch = pl.parse_sam(sam)
db = pl.parse_sam_dev(sam)
del sam
for n, steps in ((1, 10), (nf, 10 * nf), (nf, 10 * nf)):
    bench.run_steps(pl, batch, db, n, 3 * n, None, False, 0)
    capi.sync()
    t0 = time.perf_counter()
    bench.run_steps(pl, batch, db, n, steps, None, False, 0)
    capi.sync()
    print("%d in flight: %.3f ms per step" % (n, (time.perf_counter() - t0) / steps * 1e3))
names, streams = [], []
for key, p in capi._streams.items():
    names.append("main[slot %s, %d]" % (key[1], key[2]))
    streams.append(p)
buf = (C.c_void_p * 32)()
ns = C.c_int32()
capi.check(capi.lib().hgx_stream_sets_streams(buf, C.c_int32(32), C.byref(ns)))
for i in range(ns.value):
    names += ["em%d" % i, "gene%d" % i]
    streams += [C.c_void_p(buf[2 * i]), C.c_void_p(buf[2 * i + 1])]
print(engine.stream_sets_info())
print("streams:", names)
for i in range(len(streams)):
    row = []
    for j in range(len(streams)):
        same = C.c_int32()
        if i == j:
            row.append("-")
            continue
        capi.check(capi.lib().hgx_stream_probe_pair(streams[i], streams[j], C.byref(same)))
        row.append("X" if same.value else ".")
    print("%-22s %s" % (names[i], " ".join(row)))

import numpy as np  # noqa: E402
for mode, what in ((1, "the same chain on the COLUMN stream, launches interleaved: two chains on one LANE"), (0, "131 072 tiny workgroups on the COLUMN stream")):
    print("chain of 16 short kernels on the ROW stream beside %s (us alone -> us beside)" % what)
    for i in range(len(streams)):
        row = []
        for j in range(len(streams)):
            if i == j:
                row.append("      -     ")
                continue
            us = np.zeros(2)
            capi.check(capi.lib().hgx_stream_probe_chain(streams[i], streams[j], C.c_int32(mode), capi.ptr(us)))
            row.append("%4.0f->%-5.0f" % (us[0], us[1]))
        print("%-22s %s" % (names[i], " ".join(row)))
