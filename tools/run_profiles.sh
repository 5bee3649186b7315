#!/bin/bash
# Produce the judged profile set on the GPU box: tools/run_profiles.sh <name>  ->  gpurun_out/<name>/ (then tools/install_profiles.py).
# rocprofv3 gets the program itself after `--`; counters are collected in their own passes (no trace options beside --pmc).
set -u
D=gpurun_out/$1
mkdir -p $D
R=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd $R
python3 bench.py > $D/bench_default.json 2> $D/bench_default.err
python3 bench.py --no-cpu-baseline --inflight 2 > $D/bench_inflight2.json 2>> $D/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -o r02 -- python3 bench.py --no-cpu-baseline > $D/bench_under_rocprof.json 2> $D/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/pmc_fetch -o r02 -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2> $D/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/pmc_write -o r02 -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2> $D/pmc_write.err
rm -f $D/stats/*kernel_trace.csv $D/stats/*agent_info.csv $D/pmc_*/*agent_info.csv
ls -la $D $D/stats $D/pmc_fetch | head -40
cat $D/bench_default.json | head -c 400
