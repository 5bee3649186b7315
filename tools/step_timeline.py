#!/usr/bin/env python3
"""One bench step as a timeline from a rocprofv3 kernel trace (tools/run_profiles_r03.sh): every dispatch of the LAST step -- start
offset, duration, queue, gap to the previous dispatch on the same queue -- and the busy time per queue.  The step starts at the
last k_piece_compat_tiled dispatch (the first kernel of the device path).  Usage: step_timeline.py <kernel_trace.csv>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
starts = [i for i, r in enumerate(rows) if "k_piece_compat" in r["Kernel_Name"]]
i0 = starts[-1]
step = rows[i0:]
t0 = step[0]["s"]
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)[:44]
print("# dispatches of the last step: %d, span %.1f us (first start -> last end)" % (len(step), (max(r["e"] for r in step) - t0) / 1e3))
print("# %8s %8s %6s %8s  %s" % ("start_us", "dur_us", "queue", "gap_us", "kernel"))
last_end = {}
busy = {}
for r in step:
    q = r.get("Queue_Id", "?")
    gap = (r["s"] - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = r["e"]
    busy[q] = busy.get(q, 0) + (r["e"] - r["s"])
    print("  %8.1f %8.1f %6s %8.1f  %s" % ((r["s"] - t0) / 1e3, (r["e"] - r["s"]) / 1e3, q, gap, short(r["Kernel_Name"])))
print("# busy per queue (us): " + ", ".join("%s: %.1f" % (q, b / 1e3) for q, b in sorted(busy.items())))
# union of busy intervals = time with at least one kernel running
iv = sorted((r["s"], r["e"]) for r in step)
tot, cs, ce = 0, iv[0][0], iv[0][1]
for s_, e_ in iv[1:]:
    if s_ > ce:
        tot += ce - cs; cs, ce = s_, e_
    else:
        ce = max(ce, e_)
tot += ce - cs
print("# at least one kernel running: %.1f us; nothing running: %.1f us" % (tot / 1e3, (max(r["e"] for r in step) - t0 - tot) / 1e3))
