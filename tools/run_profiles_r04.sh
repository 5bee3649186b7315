#!/bin/bash
# Round 4 profile set on the GPU box: tools/run_profiles_r04.sh <name>  ->  gpurun_out/<name>/ (then tools/install_profiles_r04.py).
# rocprofv3 gets the program itself after `--`; counters are collected in their own passes (no trace options beside --pmc).
set -u
D=gpurun_out/$1
mkdir -p $D
R=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd $R
python3 bench.py > $D/bench_default.json 2> $D/bench_default.err
python3 bench.py --workload panel64 --em-exact --no-cpu-baseline --steps 6 --warmup 2 > $D/bench_panel64_em_exact.json 2>> $D/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -o r04 -- python3 bench.py --no-cpu-baseline --no-e2e --no-workloads > $D/bench_under_rocprof.json 2> $D/stats.err
rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats_panel -o r04p -- python3 bench.py --workload panel64 --no-cpu-baseline --steps 8 --warmup 2 > $D/panel_under_rocprof.json 2> $D/stats_panel.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/pmc_fetch -o r04 -- python3 bench.py --no-cpu-baseline --no-e2e --no-workloads --steps 3 --warmup 1 > /dev/null 2> $D/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/pmc_write -o r04 -- python3 bench.py --no-cpu-baseline --no-e2e --no-workloads --steps 3 --warmup 1 > /dev/null 2> $D/pmc_write.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/pmc_fetch_panel -o r04p -- python3 bench.py --workload panel64 --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> $D/pmc_fetch_panel.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/pmc_write_panel -o r04p -- python3 bench.py --workload panel64 --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> $D/pmc_write_panel.err
# one step as a timeline: kernel trace of a 1-step run (tools/step_timeline.py turns the CSV into profiles/r04_step_timeline.txt)
rocprofv3 --kernel-trace --output-format csv -d $D/trace_step -o r04s -- python3 bench.py --no-cpu-baseline --no-e2e --no-workloads --steps 1 --warmup 3 > /dev/null 2> $D/trace_step.err
python3 tools/step_timeline.py $D/trace_step/r04s_kernel_trace.csv > $D/step_timeline.txt 2>> $D/trace_step.err
rocprofv3 --pmc SQ_WAVES --output-format csv -d $D/pmc_waves -o r04 -- python3 bench.py --no-cpu-baseline --no-e2e --no-workloads --steps 2 --warmup 1 > /dev/null 2> $D/pmc_waves.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $D/pmc_busy -o r04 -- python3 bench.py --no-cpu-baseline --no-e2e --no-workloads --steps 2 --warmup 1 > /dev/null 2> $D/pmc_busy.err
rm -f $D/trace_step/*agent_info.csv
rm -f $D/stats*/*kernel_trace.csv $D/stats*/*agent_info.csv $D/pmc_*/*agent_info.csv
ls -la $D $D/stats $D/stats_panel $D/pmc_fetch $D/pmc_fetch_panel | head -60
head -c 400 $D/bench_default.json
# the file -> result call (device front end + typing path) under the tracer: SAM text and coordinate-sorted BAM
rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats_fe_sam -o r04fe -- python3 tools/e2e_file.py 500000 0 > $D/fe_sam.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats_fe_bam -o r04fe -- python3 tools/e2e_bam.py 500000 > $D/fe_bam.log 2>&1
rm -f $D/stats_fe_*/*kernel_trace.csv $D/stats_fe_*/*agent_info.csv
# the stages of the calls as the library prints them (HGX_PARSE_PROFILE), untraced
HGX_PARSE_PROFILE=1 python3 tools/e2e_file.py 500000 0 > $D/fe_sam_stages.log 2>&1
HGX_PARSE_PROFILE=1 python3 tools/e2e_bam.py 500000 > $D/fe_bam_stages.log 2>&1
HGX_PARSE_PROFILE=1 python3 tools/prof_many_front.py > $D/fe_many_stages.log 2>&1
