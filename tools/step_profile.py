#!/usr/bin/env python3
"""Per-step dispatch statistics of the bench's default workload from a rocprofv3 kernel trace (tools/run_profiles_r05.sh):
dispatches per step, kernel time per step (sum of the dispatches' durations), time with at least one kernel running (union of the
intervals), the rest of the step's span (gaps: launch latency, host round trips), runtime copies / fills per step.  A step starts at a
k_piece_compat dispatch (the first kernel of the device path); the LAST `n` steps of the trace are averaged.
Usage: step_profile.py <kernel_trace.csv> <bench_under_trace.json> <out.json> [n_steps]"""
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
bench = json.load(open(sys.argv[2]))
n_avg = int(sys.argv[4]) if len(sys.argv) > 4 else 10
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
starts = [i for i, r in enumerate(rows) if "k_piece_compat" in r["Kernel_Name"]]
starts = starts[-(n_avg + 1):]
steps = [rows[a:b] for a, b in zip(starts, starts[1:])]
def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, iv[0][0], iv[0][1]
    for s_, e_ in iv[1:]:
        if s_ > ce:
            tot += ce - cs; cs, ce = s_, e_
        else:
            ce = max(ce, e_)
    return tot + ce - cs
acc = {"dispatches": 0, "kernel_ms": 0.0, "busy_ms": 0.0, "span_ms": 0.0, "runtime_copy_fill_dispatches": 0, "em_pass_dispatches": 0}
for st in steps:
    acc["dispatches"] += len(st)
    acc["kernel_ms"] += sum(r["e"] - r["s"] for r in st) / 1e6
    acc["busy_ms"] += union([(r["s"], r["e"]) for r in st]) / 1e6
    acc["span_ms"] += (max(r["e"] for r in st) - st[0]["s"]) / 1e6
    acc["runtime_copy_fill_dispatches"] += sum(1 for r in st if "__amd_rocclr" in r["Kernel_Name"])
    acc["em_pass_dispatches"] += sum(1 for r in st if "k_lut" in r["Kernel_Name"])
n = max(len(steps), 1)
per = {"dispatches_per_step": round(acc["dispatches"] / n, 1), "kernel_ms_per_step": round(acc["kernel_ms"] / n, 3),
       "ms_with_a_kernel_running_per_step": round(acc["busy_ms"] / n, 3), "step_span_ms_under_the_tracer": round(acc["span_ms"] / n, 3),
       "gap_ms_per_step": round((acc["span_ms"] - acc["busy_ms"]) / n, 3),
       "runtime_copy_fill_dispatches_per_step": round(acc["runtime_copy_fill_dispatches"] / n, 1),
       "em_pass_dispatches_per_step": round(acc["em_pass_dispatches"] / n, 1), "steps_averaged": len(steps),
       "ms_per_step_of_the_traced_run": bench["ms_per_step"]}
json.dump({"n_pairs": bench["config"]["pairs_per_gpu"], "a_pad": 7168,
           "source": "rocprofv3 --kernel-trace of `python3 bench.py --no-cpu-baseline --no-e2e --no-workloads` (tracing adds ~4 us per dispatch: "
                     "the traced step is longer than the untraced one; dispatch counts and kernel durations are what to read)",
           "per_step": per}, open(sys.argv[3], "w"), indent=1)
print(json.dumps(per))
