#!/usr/bin/env python3
"""Summarise one bench step from a rocprofv3 kernel trace: busy time per kernel and the largest idle gaps."""
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_piece_compat")]
a, b = idx[-2], idx[-1]
step = rows[a:b]
t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
busy = collections.Counter(); cnt = collections.Counter()
for r in step:
    k = r["Kernel_Name"].split("(")[0][-48:]
    busy[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cnt[k] += 1
print("step wall %.1f us, busy %.1f us, kernels %d" % ((t1 - t0) / 1e3, sum(busy.values()) / 1e3, len(step)))
for k, v in busy.most_common(14):
    print("  %-50s n=%4d %9.1f us" % (k, cnt[k], v / 1e3))
gaps = []
for i in range(1, len(step)):
    g = int(step[i]["Start_Timestamp"]) - int(step[i - 1]["End_Timestamp"])
    gaps.append((g, step[i - 1]["Kernel_Name"].split("(")[0][-30:], step[i]["Kernel_Name"].split("(")[0][-30:]))
print("gaps total %.1f us; >10us: %.1f us" % (sum(g for g, _, _ in gaps) / 1e3, sum(g for g, _, _ in gaps if g > 10000) / 1e3))
for g, x, y in sorted(gaps, reverse=True)[:14]:
    print("  %8.1f us  %s -> %s" % (g / 1e3, x, y))
