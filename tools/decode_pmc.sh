#!/bin/bash
# SQ counters of the record kernels of ONE BAM file -> result call (tools/e2e_bam.py), per launch
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/decpmc
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_SMEM" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/decpmc/$n -o t -- python3 tools/e2e_bam.py 500000 > gpurun_out/decpmc/$n.log 2>&1
  python3 - "$n" <<'P'
import csv, sys, glob, collections
n = sys.argv[1]
fs = glob.glob("gpurun_out/decpmc/%s/**/t_counter_collection.csv" % n, recursive=True)
if not fs: print(n, "no output"); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    for k in ("k_fe_decode", "k_fe_records", "k_fe_group_flags", "k_fe_pair_count", "k_bam_walk<0>", "k_fe_pileup"):
        if k in r["Kernel_Name"]:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k.ljust(18), "  ".join("%s %.4g" % (c, sum(v) / len(v)) for c, v in sorted(d.items())))
P
done
