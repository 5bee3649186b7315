#!/usr/bin/env python3
"""Copy the judged summaries of a gpurun profile directory (bench JSONs, rocprofv3 kernel stats, PMC passes) into profiles/
and rebuild profiles/traffic.json.  Usage: tools/install_profiles.py gpurun_out/<dir>"""
import collections, csv, json, os, shutil, sys
src = sys.argv[1]
R = sys.argv[2] if len(sys.argv) > 2 else "r02"            # round prefix of the files in profiles/
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
for f in os.listdir(dst):
    if f.startswith(R + "_final"):
        os.remove(os.path.join(dst, f))
shutil.copy(f"{src}/stats/{R}_kernel_stats.csv", f"{dst}/{R}_final_kernel_stats.csv")
shutil.copy(f"{src}/bench_default.json", f"{dst}/{R}_final_bench.json")
shutil.copy(f"{src}/bench_inflight2.json", f"{dst}/{R}_final_bench_inflight2.json")
shutil.copy(f"{src}/bench_under_rocprof.json", f"{dst}/{R}_final_bench_under_rocprofv3.json")
summ = {}
for name, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    rows = list(csv.DictReader(open(f"{src}/{name}/{R}_counter_collection.csv")))
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in rows:
        a = agg[r["Kernel_Name"]]
        v = float(r["Counter_Value"])
        a[0] += 1; a[1] += v; a[2] = max(a[2], v)
    with open(f"{dst}/{R}_final_pmc_{ctr}.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Dispatches", "Counter", ctr + "_avg_KB", ctr + "_max_KB", ctr + "_total_KB"])
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, a[0], ctr, round(a[1] / a[0], 1), round(a[2], 1), round(a[1], 1)])
    summ[ctr] = {k: a[1] / a[0] for k, a in agg.items()}
def get(ctr, sub):
    for k, v in summ[ctr].items():
        if sub in k:
            return v
d = json.load(open(f"{src}/bench_default.json"))
names = {"k_pair_classes": "k_pair_classes_x2<2, false>", "k_piece_compat": "k_piece_compat_tiled", "k_lutmatvec<0>": "k_lutmatvec<0>",
         "k_lutmatvec<1>": "k_lutmatvec<1>"}
raw, hb = {}, {}
for k, sub in names.items():
    f, w = get("FETCH_SIZE", sub), get("WRITE_SIZE", sub)
    raw[k] = {"FETCH_SIZE": f, "WRITE_SIZE": w}
    hb[k] = int((2 * f + w) * 1024)
json.dump({"n_pairs": d["config"]["pairs_per_gpu"], "a_pad": 7168,
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python3 bench.py --no-cpu-baseline --steps 3 "
                     "--warmup 1`; average per dispatch, KB -> bytes x1024; FETCH_SIZE doubled (gfx950 tallies 128-B read requests at "
                     "64 B, MI355X_MICROARCH.md HBM section)",
           "raw_KB": raw, "hbm_bytes_per_launch": hb}, open(f"{dst}/traffic.json", "w"), indent=1)
print(d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["traffic"])
print({k: (v["avg_ms"], v["total_ms_per_step"], v["GBps"]) for k, v in d["roofline"]["kernels"].items()})
u = json.load(open(f"{src}/bench_under_rocprof.json"))
n_steps = u["steps"] + u["warmup"]                      # the profiled run's steps (warm-up included: the profiler sees all)
tot = 0
for r in csv.DictReader(open(f"{dst}/{R}_final_kernel_stats.csv")):
    tot += int(r["TotalDurationNs"])
    n = r["Name"].split("(")[0][-48:]
    if int(r["TotalDurationNs"]) > 0.05e6 * n_steps:
        print("%-50s calls %5s avg %9.1f us total/step %7.3f ms" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, int(r["TotalDurationNs"]) / 1e6 / n_steps))
print("kernel time per step %.3f ms over %d steps" % (tot / 1e6 / n_steps, n_steps))
