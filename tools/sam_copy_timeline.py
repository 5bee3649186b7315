"""Timeline of the host -> device copies and the first kernels of the file -> result call on SAM text, from a rocprofv3 trace
(--memory-copy-trace --kernel-trace of tools/e2e_file.py): when each phase of the text starts and ends its way up, at which rate,
and when the first kernel that reads it starts.  usage: python tools/sam_copy_timeline.py <trace dir>"""
import csv, glob, sys
d = sys.argv[1]
cp = [r for p in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True) for r in csv.DictReader(open(p))]
kr = [r for p in glob.glob(d + "/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(p))]
big = [r for r in cp if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) >= 400000 and "HOST_TO_DEVICE" in r["Direction"]]      # (this rocprofv3 writes no sizes: the long ones)
big.sort(key=lambda r: int(r["Start_Timestamp"]))
# the calls: groups of big copies less than 50 ms apart
calls, cur = [], []
for r in big:
    if cur and int(r["Start_Timestamp"]) - int(cur[-1]["End_Timestamp"]) > 50e6:
        calls.append(cur); cur = []
    cur.append(r)
if cur: calls.append(cur)
print("%d big copies in %d calls" % (len(big), len(calls)))
for c in calls[-2:]:
    t0 = int(c[0]["Start_Timestamp"])
    tot = 0
    for r in c:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        n = int(float(sys.argv[2]) * 1e6 / len(c)) if len(sys.argv) > 2 else 0          # (MB of text, shared out evenly over the phases)
        tot += n
        print("  copy  start %7.3f ms  end %7.3f ms  %6.3f ms  %5.1f GB/s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, n / max(e - s, 1)))
    t_end = int(c[-1]["End_Timestamp"])
    print("  %.1f MB in %.3f ms from the first copy's start = %.1f GB/s" % (tot / 1e6, (t_end - t0) / 1e6, tot / (t_end - t0)))
    ks = sorted((r for r in kr if t0 <= int(r["Start_Timestamp"]) <= t_end + 30e6), key=lambda r: int(r["Start_Timestamp"]))
    for r in (ks if len(sys.argv) > 3 else ks[:6]):
        if len(sys.argv) > 3 and (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) < 30000 and "k_" not in r["Kernel_Name"]: continue
        print("  kernel %-40s start %7.3f ms  dur %7.1f us" % (r["Kernel_Name"][:40], (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    if ks:
        print("  last kernel of the call ends at %.3f ms" % ((max(int(r["End_Timestamp"]) for r in ks) - t0) / 1e6))
