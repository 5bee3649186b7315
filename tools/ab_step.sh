#!/bin/bash
# the default bench step with two builds of libhgx, alternating on the same box: tools/ab_step.sh <other libhgx.so> [rounds]
R=$(pwd)
for i in $(seq 1 ${2:-4}); do
  a=$(python3 bench.py --no-cpu-baseline --no-e2e --no-workloads --steps 100 --warmup 10 2>/dev/null | python3 -c "import json,sys; print(json.load(sys.stdin)['ms_per_step'])")
  b=$(python3 tools/bench_with_lib.py $1 --no-cpu-baseline --no-e2e --no-workloads --steps 100 --warmup 10 2>/dev/null | python3 -c "import json,sys; print(json.load(sys.stdin)['ms_per_step'])")
  echo "this tree $a ms | $1 $b ms"
done
