"""Summary of a rocprofv3 kernel trace: launches, summed kernel time, wall span, busy time per queue, the ten heaviest kernels."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
t1 = max(int(r["End_Timestamp"]) for r in rows)
tot = collections.Counter(); cnt = collections.Counter(); q = collections.Counter()
for r in rows:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:50]
    tot[n] += d; cnt[n] += 1; q[r.get("Queue_Id", "?")] += d
# union of busy intervals
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
busy = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
print("launches %d, span %.1f ms, some kernel running %.1f ms (%.0f %%), summed kernel time %.1f ms" % (
    len(rows), (t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), sum(tot.values()) / 1e6))
print("per queue busy ms:", {k: round(v / 1e6, 1) for k, v in q.items()})
for n, v in tot.most_common(12):
    print("  %-52s %7d launches %9.2f ms  avg %7.1f us" % (n, cnt[n], v / 1e6, v / cnt[n] / 1e3))
