#!/bin/bash
# Round 5 profile set on the GPU box: tools/run_profiles_r05.sh <name>  ->  gpurun_out/<name>/ (then tools/install_profiles_r05.py).
# rocprofv3 gets the program itself after `--`; counters are collected in their own passes (no trace options beside --pmc).
set -u
D=gpurun_out/$1
mkdir -p $D
R=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd $R
# a fresh box's FIRST process runs its multi-threaded phases slower (class1 5.7 instead of 4.4 ms per step, the single sample 2.13
# instead of 2.06: the image is still paging in): a throw-away run first, so that the committed line is the steady state
python3 bench.py --no-cpu-baseline --no-e2e --wl-steps 4 > $D/bench_first_process.json 2> /dev/null
python3 bench.py > $D/bench_default.json 2> $D/bench_default.err
python3 bench.py --workload panel64 --em-exact --no-cpu-baseline --steps 6 --warmup 2 > $D/bench_panel64_em_exact.json 2>> $D/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -o r05 -- python3 bench.py --no-cpu-baseline --no-e2e --no-workloads > $D/bench_under_rocprof.json 2> $D/stats.err
rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats_panel -o r05p -- python3 bench.py --workload panel64 --no-cpu-baseline --steps 8 --warmup 2 > $D/panel_under_rocprof.json 2> $D/stats_panel.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/pmc_fetch -o r05 -- python3 bench.py --no-cpu-baseline --no-e2e --no-workloads --steps 3 --warmup 1 > /dev/null 2> $D/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/pmc_write -o r05 -- python3 bench.py --no-cpu-baseline --no-e2e --no-workloads --steps 3 --warmup 1 > /dev/null 2> $D/pmc_write.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/pmc_fetch_panel -o r05p -- python3 bench.py --workload panel64 --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> $D/pmc_fetch_panel.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/pmc_write_panel -o r05p -- python3 bench.py --workload panel64 --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> $D/pmc_write_panel.err
# one step as a timeline: kernel trace of a 1-step run (tools/step_timeline.py turns the CSV into profiles/r05_step_timeline.txt)
rocprofv3 --kernel-trace --output-format csv -d $D/trace_step -o r05s -- python3 bench.py --no-cpu-baseline --no-e2e --no-workloads --steps 1 --warmup 3 > /dev/null 2> $D/trace_step.err
python3 tools/step_timeline.py $D/trace_step/r05s_kernel_trace.csv > $D/step_timeline.txt 2>> $D/trace_step.err
# dispatches / kernel time / gaps per step (bench.py carries these as roofline.step_profile): a trace of the bench's own command
rocprofv3 --kernel-trace --output-format csv -d $D/trace_steps -o r05t -- python3 bench.py --no-cpu-baseline --no-e2e --no-workloads > $D/bench_under_trace.json 2> $D/trace_steps.err
python3 tools/step_profile.py $D/trace_steps/r05t_kernel_trace.csv $D/bench_under_trace.json $D/step_profile.json 10 >> $D/trace_steps.err 2>&1
rm -f $D/trace_steps/*kernel_trace.csv $D/trace_steps/*agent_info.csv
rocprofv3 --pmc SQ_WAVES --output-format csv -d $D/pmc_waves -o r05 -- python3 bench.py --no-cpu-baseline --no-e2e --no-workloads --steps 2 --warmup 1 > /dev/null 2> $D/pmc_waves.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $D/pmc_busy -o r05 -- python3 bench.py --no-cpu-baseline --no-e2e --no-workloads --steps 2 --warmup 1 > /dev/null 2> $D/pmc_busy.err
rm -f $D/trace_step/*agent_info.csv
rm -f $D/stats*/*kernel_trace.csv $D/stats*/*agent_info.csv $D/pmc_*/*agent_info.csv
ls -la $D $D/stats $D/stats_panel $D/pmc_fetch $D/pmc_fetch_panel | head -60
head -c 400 $D/bench_default.json
# the file -> result call (device front end + typing path) under the tracer: SAM text and coordinate-sorted BAM
rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats_fe_sam -o r05fe -- python3 tools/e2e_file.py 500000 0 > $D/fe_sam.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats_fe_bam -o r05fe -- python3 tools/e2e_bam.py 500000 > $D/fe_bam.log 2>&1
rm -f $D/stats_fe_*/*kernel_trace.csv $D/stats_fe_*/*agent_info.csv
# the stages of the calls as the library prints them (HGX_PARSE_PROFILE), untraced
HGX_PARSE_PROFILE=1 python3 tools/e2e_file.py 500000 0 > $D/fe_sam_stages.log 2>&1
HGX_PARSE_PROFILE=1 python3 tools/e2e_bam.py 500000 > $D/fe_bam_stages.log 2>&1
HGX_PARSE_PROFILE=1 python3 tools/prof_many_front.py > $D/fe_many_stages.log 2>&1
# the device inflate's two forms on the two BAM workloads (kernel times) and the default form's phase laps
bash tools/inf_ab.sh > $D/inflate_forms.txt 2>&1
INF_FORM=inflate_prof python3 tools/inflate_probe.py run 2>&1 | grep "k_bgzf_inflate_w\|call (with" | sort -u > $D/inflate_phases.txt
