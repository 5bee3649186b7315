"""Timeline of the LAST bench step in a rocprofv3 kernel trace: per kernel start (us since the step's first kernel), duration, queue."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steps start with k_piece_compat_tiled
starts = [i for i, r in enumerate(rows) if "k_piece_compat_tiled" in r["Kernel_Name"]]
i0 = starts[-1]
t0 = int(rows[i0]["Start_Timestamp"])
last_end = {}
busy = collections.defaultdict(int)
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    q = r.get("Queue_Id", "?")
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
    gap = s - last_end.get(q, s)
    print("%9.1f us  +%7.1f us  q%-3s gap %6.1f  %s" % (s / 1e3, (e - s) / 1e3, q, gap / 1e3, name))
    last_end[q] = e
    busy[q] += e - s
print({q: round(v / 1e3, 1) for q, v in busy.items()}, "total span %.1f us" % (max(int(r["End_Timestamp"]) for r in rows[i0:]) / 1e3 - t0 / 1e3))
