#!/bin/bash
# one bench step as a kernel timeline, with and without the gene-side CU mask (HGX_GENE_CUS): where does the step's time go?
set -u
R=$(pwd); cd /tmp && export TMPDIR=/tmp && cd $R
for m in 0 160; do
  D=gpurun_out/trace_mask_$m; mkdir -p $D
  HGX_GENE_CUS=$m rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python3 bench.py --no-cpu-baseline --no-e2e --no-workloads --no-kernel-timing --steps 6 --warmup 4 > $D/bench.json 2> $D/err.log
  python3 tools/step_timeline.py $D/t_kernel_trace.csv > $D/timeline.txt
  python3 - <<PY
import json
d=json.load(open("$D/bench.json")); print("mask $m: ms_per_step", d["ms_per_step"])
PY
  rm -f $D/t_kernel_trace.csv $D/t_agent_info.csv
done
