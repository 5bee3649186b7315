#!/bin/bash
# kernel trace (start/end timestamps per dispatch) of a short bench run: gpurun_out/trace/  (analysed by tools/trace_timeline.py)
set -u
D=gpurun_out/trace
mkdir -p $D
R=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd $R
rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python3 bench.py --no-cpu-baseline --no-e2e --no-kernel-timing --steps 4 --warmup 3 > $D/bench.json 2> $D/err.log
ls -la $D
