#!/bin/bash
# kernel averages of the configs[1] step (bench.py without the file legs) from rocprofv3 --stats, and the step time untraced
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
rm -rf gpurun_out/sk
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sk -o t -- python3 bench.py --no-cpu-baseline --no-e2e --no-workloads --no-kernel-timing --steps 30 --warmup 5 > gpurun_out/sk.log 2>&1
python3 - <<'P'
import csv
rows = list(csv.DictReader(open("gpurun_out/sk/t_kernel_stats.csv")))
for r in rows[:22]:
    print(r["Name"].split("(")[0].replace("void ", "")[:44].ljust(44), r["Calls"].rjust(6), "%8.1f us" % (float(r["AverageNs"]) / 1e3), "%5.1f%%" % float(r["Percentage"]))
P
for i in 1 2 3; do python3 bench.py --no-cpu-baseline --no-e2e --no-workloads --steps 50 --warmup 5 2>/dev/null | python3 -c "import json,sys; b=json.loads(sys.stdin.read()); print('step', b['ms_per_step'])"; done
