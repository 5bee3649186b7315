"""Where a BAM file -> result call idles: the kernels of the LAST call of a rocprofv3 --kernel-trace --memory-copy-trace of
tools/e2e_bam.py in start order, with every gap of 25 us and more during which NO kernel and no copy was running.
usage: python tools/bam_gap_timeline.py <trace dir> [marker kernel: a kernel that runs once per call, default k_bgzf_inflate]"""
import csv, glob, sys
d = sys.argv[1]
kr = [r for p in glob.glob(d + "/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(p))]
cp = [r for p in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True) for r in csv.DictReader(open(p))]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:44]) for r in kr]
ev += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r["Direction"].replace("MEMORY_COPY_", "")) for r in cp]
ev.sort()
# calls: the inflate kernel marks one per call
marker = sys.argv[2] if len(sys.argv) > 2 else "k_bgzf_inflate"
starts = [e[0] for e in ev if marker in e[2]]
if not starts:
    sys.exit("no %s kernel in the trace" % marker)
t_inf = starts[-1]
# the call's first event: walk back while gaps are below 5 ms
i = max(k for k, e in enumerate(ev) if e[0] == t_inf)
lo = i
while lo > 0 and ev[lo][0] - max(e[1] for e in ev[max(0, lo - 8):lo]) < 3e6:
    lo -= 1
hi = i
while hi + 1 < len(ev) and ev[hi + 1][0] - max(e[1] for e in ev[lo:hi + 1]) < 3e6:
    hi += 1
call = ev[lo:hi + 1]
t0 = call[0][0]
busy_end = call[0][0]
tot_gap = 0
min_idle = float(sys.argv[3]) * 1000 if len(sys.argv) > 3 else 25000          # [3] = idle stretches to print, us; [4] = kernels to print, us
min_kernel = float(sys.argv[4]) * 1000 if len(sys.argv) > 4 else 40000
print("%d events, span %.3f ms" % (len(call), (max(e[1] for e in call) - t0) / 1e6))
for s, e, nm in call:
    if s - busy_end >= min_idle:
        print("   ---- idle %7.1f us (from %.3f ms)" % ((s - busy_end) / 1e3, (busy_end - t0) / 1e6))
    if s > busy_end:
        tot_gap += s - busy_end
    if e - s >= min_kernel:
        print("%8.3f ms  %8.1f us  %s" % ((s - t0) / 1e6, (e - s) / 1e3, nm))
    busy_end = max(busy_end, e)
print("idle in all: %.3f ms" % (tot_gap / 1e6))
