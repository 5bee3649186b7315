#!/bin/bash
# per kernel of ONE BAM file -> result call: wavefront-cycles per instruction and the share spent waiting (kernels with few
# instructions and long-lived wavefronts are waiting on something serial: this is how k_fe_pair_count's atomics were found)
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/allpmc
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY"; do
  rm -rf gpurun_out/allpmc/a
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/allpmc/a -o t -- python3 tools/e2e_bam.py 500000 > gpurun_out/allpmc/a.log 2>&1
done
python3 - <<'P'
import csv, glob, collections
fs = glob.glob("gpurun_out/allpmc/a/**/t_counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": calls[k] += 1
rows = []
for k, d in acc.items():
    ins = d["SQ_INSTS_VALU"] + d["SQ_INSTS_SALU"]
    if d["SQ_WAVES"] == 0 or ins == 0: continue
    rows.append((d["SQ_WAVE_CYCLES"], k, calls[k], d["SQ_WAVES"], ins / d["SQ_WAVES"], 4 * d["SQ_WAVE_CYCLES"] / ins, d["SQ_WAIT_ANY"] / max(d["SQ_WAVE_CYCLES"], 1)))
rows.sort(reverse=True)
print("%-48s %6s %9s %10s %9s %6s" % ("kernel", "calls", "waves", "instr/wave", "clk/instr", "wait"))
for wc, k, c, w, ipw, cpi, wt in rows[:40]:
    print("%-48s %6d %9d %10.0f %9.1f %5.0f%%" % (k, c, w, ipw, cpi, 100 * wt))
P
