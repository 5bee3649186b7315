#!/usr/bin/env python3
"""Print the first part of one bench step from a rocprofv3 kernel trace: every kernel until the first EM mat-vec launch,
with start/end relative to the step's k_piece_compat launch and the queue it ran on."""
import csv, glob, re, sys
d = sys.argv[1]
f = glob.glob(d + "/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_piece_compat" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
n_lut = 0
for r in rows[a:b]:
    k = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"]).split("(")[0][:44]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print("%9.1f %9.1f  %7.1f  q%-3s %s" % (s, e, e - s, r.get("Queue_Id", "?"), k))
    if "k_lut" in k:
        n_lut += 1
        if n_lut >= 3 and "--all" not in sys.argv:
            break
print("step wall %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
