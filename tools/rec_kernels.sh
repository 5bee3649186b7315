#!/bin/bash
# the record kernels of ONE BAM file -> result call (tools/e2e_bam.py 500000): average kernel times from rocprofv3 --stats
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
rm -rf gpurun_out/wk
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/wk -o t -- python3 tools/e2e_bam.py 500000 > gpurun_out/wk.log 2>&1
python3 - <<'P'
import csv
for r in csv.DictReader(open("gpurun_out/wk/t_kernel_stats.csv")):
    for k in ("k_bgzf_inflate", "k_bam_walk<0>", "k_fe_pair_count","k_bam_filter","k_bam_name_key","k_fe_group_flags","k_fe_records","k_fe_decode","k_fe_pair_emit","k_fe_rec_filter_insert","k_fe_pileup","k_fe_build"):
        if k in r["Name"]: print(k.ljust(24), r["Calls"], "%.1f us"%(float(r["AverageNs"])/1e3))
P
grep "^run" gpurun_out/wk.log | tail -2
