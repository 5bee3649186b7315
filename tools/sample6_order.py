"""Does workloads.sample6 depend on what ran before it in the process?  (bench.py: 25 ms behind the other legs, 16-18 ms alone.)"""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import bench  # noqa: E402
from hisatgenotype_amd import capi, engine  # noqa: E402

before = os.environ.get("BEFORE", "")
args = bench.parse_args()
args.no_e2e, args.no_cpu_baseline, args.no_kernel_timing = True, True, True
capi.set_device(0)
for what in [w for w in before.split(",") if w]:
    a2 = copy.copy(args)
    a2.steps, a2.warmup, a2.workload = 3, 1, what
    if what == "class1":
        a2.pairs = 100000
        bench.run_class1(a2, 0, 0, 1, None)
    elif what == "panel64":
        a2.panel_pairs = 1000
        bench.run_panel64(a2, 0, 0, 1, None)
    elif what.startswith("streams"):
        import ctypes as C
        keep = []
        for k in range(int(what[7:] or 20)):
            p_ = C.c_void_p()
            capi.check(capi.lib().hgx_stream_create_prio(C.byref(p_), C.c_int(0)))
            keep.append(p_)
    elif what == "dropin":
        bench.run_dropin(a2, "hla_7000_10k")
    print("ran", what, engine.stream_sets_info())
r = bench.run_sample6(args)
print("BEFORE=%-22s sample6 %.2f ms  all %s  table-lookup %.2f  one-wg %.2f  sets %s" % (
    before, r["value"], r["all_five_repeated_calls_ms"], r["table_lookup_em_ms"], r["reference_order_em_on_one_workgroup_per_locus_ms"], engine.stream_sets_info()))
