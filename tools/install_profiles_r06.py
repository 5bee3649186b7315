#!/usr/bin/env python3
"""Copy the judged summaries of a round-6 gpurun profile directory (tools/run_profiles_r06.sh) into profiles/ and rebuild
profiles/traffic.json (per-launch fabric bytes of the hot kernels of BOTH driver-timed workloads).  Usage:
tools/install_profiles_r06.py gpurun_out/<dir>"""
import collections, csv, json, os, shutil, sys
src = sys.argv[1]
R = "r06"
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
if not os.path.exists(f"{src}/stats/r06_kernel_stats.csv"):
    sys.exit("no profile set under %s (usage: tools/install_profiles_r06.py gpurun_out/<dir>)" % src)
for f in os.listdir(dst):
    if f.startswith(R + "_final"):
        os.remove(os.path.join(dst, f))
shutil.copy(f"{src}/stats/r06_kernel_stats.csv", f"{dst}/{R}_final_kernel_stats.csv")
shutil.copy(f"{src}/stats_panel/r06p_kernel_stats.csv", f"{dst}/{R}_final_panel64_kernel_stats.csv")
shutil.copy(f"{src}/bench_default.json", f"{dst}/{R}_final_bench.json")
shutil.copy(f"{src}/bench_panel64_em_exact.json", f"{dst}/{R}_final_bench_panel64_em_exact.json")
shutil.copy(f"{src}/bench_under_rocprof.json", f"{dst}/{R}_final_bench_under_rocprofv3.json")
shutil.copy(f"{src}/panel_under_rocprof.json", f"{dst}/{R}_final_panel64_under_rocprofv3.json")
summ = {}
for tag, stem in (("", "r06"), ("_panel", "r06p")):
    for name, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        rows = list(csv.DictReader(open(f"{src}/{name}{tag}/{stem}_counter_collection.csv")))
        agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
        for r in rows:
            a = agg[r["Kernel_Name"]]
            v = float(r["Counter_Value"])
            a[0] += 1; a[1] += v; a[2] = max(a[2], v)
        with open(f"{dst}/{R}_final{'_panel64' if tag else ''}_pmc_{ctr}.csv", "w") as f:
            w = csv.writer(f)
            w.writerow(["Kernel_Name", "Dispatches", "Counter", ctr + "_avg_KB", ctr + "_max_KB", ctr + "_total_KB"])
            for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                w.writerow([k, a[0], ctr, round(a[1] / a[0], 1), round(a[2], 1), round(a[1], 1)])
        summ[(tag, ctr)] = {k: a[1] / a[0] for k, a in agg.items()}
def get(tag, ctr, sub):
    for k, v in summ[(tag, ctr)].items():
        if sub in k:
            return v
d = json.load(open(f"{src}/bench_default.json"))
names = {"k_pair_classes": ("", "k_pair_classes_x2<2, false>"), "k_piece_compat": ("", "k_piece_compat_pat"), "k_lutmatvec<0>": ("", "k_lutmatvec<0>"),
         "k_lutmatvec<1>": ("", "k_lutmatvec<1>"), "k_emx (table lookups)": ("_panel", "k_emx<true"), "k_emx (reference order)": ("_panel", "k_emx<false")}
raw, hb = {}, {}
for k, (tag, sub) in names.items():
    f, w = get(tag, "FETCH_SIZE", sub), get(tag, "WRITE_SIZE", sub)
    if f is None or w is None:
        continue
    raw[k] = {"FETCH_SIZE": f, "WRITE_SIZE": w}
    hb[k] = int((2 * f + w) * 1024)
json.dump({"n_pairs": d["config"]["pairs_per_gpu"], "a_pad": 7168,
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python3 bench.py --no-cpu-baseline --no-e2e --no-workloads "
                     "--steps 3 --warmup 1` (configs[1] kernels) and of `python3 bench.py --workload panel64 --no-cpu-baseline --steps 2 --warmup 1` "
                     "(k_emx); average per dispatch, KB -> bytes x1024; FETCH_SIZE doubled (gfx950 tallies 128-B read requests at 64 B, "
                     "MI355X_MICROARCH.md HBM section)",
           "raw_KB": raw, "hbm_bytes_per_launch": hb}, open(f"{dst}/traffic.json", "w"), indent=1)
print(d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["traffic"])
for k, w in d.get("workloads", {}).items():
    print(k, w.get("ms_per_step"), w.get("value"), (w.get("roofline") or {}).get("kernel"), (w.get("roofline") or {}).get("frac"))
for stats, u in ((f"{dst}/{R}_final_kernel_stats.csv", f"{src}/bench_under_rocprof.json"), (f"{dst}/{R}_final_panel64_kernel_stats.csv", f"{src}/panel_under_rocprof.json")):
    uj = json.load(open(u))
    n_steps = uj["steps"] + uj["warmup"]
    tot = 0
    for r in csv.DictReader(open(stats)):
        tot += int(r["TotalDurationNs"])
        n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-48:]
        if int(r["TotalDurationNs"]) > 0.05e6 * n_steps:
            print("%-50s calls %5s avg %9.1f us total/step %7.3f ms" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, int(r["TotalDurationNs"]) / 1e6 / n_steps))
    print("kernel time per step %.3f ms over %d steps" % (tot / 1e6 / n_steps, n_steps))
# the one-step timeline and the wave / busy counters
if os.path.exists(f"{src}/step_timeline.txt") and os.path.getsize(f"{src}/step_timeline.txt") > 0:
    shutil.copy(f"{src}/step_timeline.txt", f"{dst}/{R}_final_step_timeline.txt")
for name, stem in (("pmc_waves", "r06"), ("pmc_busy", "r06")):
    path = f"{src}/{name}/{stem}_counter_collection.csv"
    if not os.path.exists(path):
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(path)):
        a = agg[r["Kernel_Name"]][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
    with open(f"{dst}/{R}_final_{name}.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Counter", "Dispatches", "avg", "total"])
        for k, d in sorted(agg.items(), key=lambda kv: -sum(v[1] for v in kv[1].values())):
            for c, a in sorted(d.items()):
                w.writerow([k, c, a[0], round(a[1] / a[0], 1), round(a[1], 1)])
for name in ("step_profile.json", "inflate_forms.txt", "inflate_phases.txt"):
    if os.path.exists(f"{src}/{name}") and os.path.getsize(f"{src}/{name}") > 0:
        shutil.copy(f"{src}/{name}", f"{dst}/{'step_profile.json' if name == 'step_profile.json' else R + '_final_' + name}")
for kind in ("sam", "bam"):
    for stem in ("fe_%s_stages.log" % kind, "fe_many_stages.log"):
        if os.path.exists(f"{src}/{stem}"):
            shutil.copy(f"{src}/{stem}", f"{dst}/{R}_final_" + ("file_to_result_%s_stages.log" % kind if stem.startswith("fe_" + kind) else "many_files_stages.log"))
    p = f"{src}/stats_fe_{kind}/r06fe_kernel_stats.csv"
    if os.path.exists(p):
        shutil.copy(p, f"{dst}/{R}_final_file_to_result_{kind}_kernel_stats.csv")
    p = f"{src}/fe_{kind}.log"
    if os.path.exists(p):
        shutil.copy(p, f"{dst}/{R}_final_file_to_result_{kind}.log")
# dispatches and kernel time per file -> result call, from the traced calls' kernel stats (bench.py carries them as e2e.<kind>.profile).
# The number of calls in a trace = the launches of a kernel that runs once per call (k_fe_decode runs once per front-end pass).
fr = {}
for kind in ("sam", "bam"):
    p = f"{dst}/{R}_final_file_to_result_{kind}_kernel_stats.csv"
    if not os.path.exists(p):
        continue
    rows = list(csv.DictReader(open(p)))
    calls = [int(r["Calls"]) for r in rows if "k_fe_decode" in r["Name"]]              # (k_fe_records runs twice per SAM call since the two-part form)
    if not calls or not calls[0]:
        continue
    n_calls = calls[0]
    lib = [r for r in rows if "rocprim" in r["Name"] or "hipcub" in r["Name"]]
    fr[kind] = {"calls_traced": n_calls,
                "dispatches_per_call": round(sum(int(r["Calls"]) for r in rows) / n_calls, 1),
                "kernel_ms_per_call": round(sum(int(r["TotalDurationNs"]) for r in rows) / 1e6 / n_calls, 3),
                "library_sort_dispatches_per_call": round(sum(int(r["Calls"]) for r in lib) / n_calls, 1),
                "largest_kernels_ms_per_call": {r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-40:]:
                                                round(int(r["TotalDurationNs"]) / 1e6 / n_calls, 3)
                                                for r in sorted(rows, key=lambda r: -int(r["TotalDurationNs"]))[:6]},
                "source": f"profiles/{R}_final_file_to_result_{kind}_kernel_stats.csv (rocprofv3 --kernel-trace --stats of tools/e2e_{'bam' if kind == 'bam' else 'file'}.py)"}
sp = f"{dst}/step_profile.json"
if fr and os.path.exists(sp):
    sj = json.load(open(sp))
    sj["file_to_result"] = fr
    json.dump(sj, open(sp, "w"), indent=1)
    print("file -> result per call:", {k: (v["dispatches_per_call"], v["kernel_ms_per_call"]) for k, v in fr.items()})
import glob
for p in glob.glob(f"{src}/stats_dropin/**/*kernel_stats.csv", recursive=True)[:1]:
    shutil.copy(p, f"{dst}/{R}_final_dropin_config0_kernel_stats.csv")
    shutil.copy(f"{src}/dropin_config0.log", f"{dst}/{R}_final_dropin_config0.log")
for name in ("bench_2ranks_shared_gpu.json", "bench_class1_4ranks_shared_gpu.json", "stream_probe.txt", "stream_conflicts.txt", "stream_ab.txt",
             "front_gate.txt", "dropin_profile.txt", "bench_first_process.json", "bench_class1_4ranks_rccl_entry_points.json"):
    if os.path.exists(f"{src}/{name}") and os.path.getsize(f"{src}/{name}") > 0:
        shutil.copy(f"{src}/{name}", f"{dst}/{R}_final_{name}")
