#!/bin/bash
# kernel stats of the file -> result call (device front end + typing) under rocprofv3: tools/prof_front.sh -> gpurun_out/prof_fe_{sam,bam}
R=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_fe_sam -o fe -- python3 tools/e2e_file.py 500000 0 > gpurun_out/prof_fe_sam.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_fe_bam -o fe -- python3 tools/e2e_bam.py 500000 > gpurun_out/prof_fe_bam.log 2>&1
rm -f gpurun_out/prof_fe_*/*kernel_trace.csv gpurun_out/prof_fe_*/*agent_info.csv
for d in gpurun_out/prof_fe_sam gpurun_out/prof_fe_bam; do f=$(find $d -name "*kernel_stats.csv" | head -1); echo "== $f"; head -30 "$f" | cut -c1-160; done
