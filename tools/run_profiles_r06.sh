#!/bin/bash
# Round 6 profile set on the GPU box: tools/run_profiles_r06.sh <name>  ->  gpurun_out/<name>/ (then tools/install_profiles_r06.py).
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
rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -o r06 -- python3 bench.py --no-cpu-baseline --no-e2e --no-workloads > $D/bench_under_rocprof.json 2> $D/stats.err
rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats_panel -o r06p -- python3 bench.py --workload panel64 --no-cpu-baseline --steps 8 --warmup 2 > $D/panel_under_rocprof.json 2> $D/stats_panel.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/pmc_fetch -o r06 -- python3 bench.py --no-cpu-baseline --no-e2e --no-workloads --steps 3 --warmup 1 > /dev/null 2> $D/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/pmc_write -o r06 -- python3 bench.py --no-cpu-baseline --no-e2e --no-workloads --steps 3 --warmup 1 > /dev/null 2> $D/pmc_write.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/pmc_fetch_panel -o r06p -- python3 bench.py --workload panel64 --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> $D/pmc_fetch_panel.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/pmc_write_panel -o r06p -- python3 bench.py --workload panel64 --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> $D/pmc_write_panel.err
# one step as a timeline: kernel trace of a 1-step run (tools/step_timeline.py turns the CSV into profiles/r06_step_timeline.txt)
rocprofv3 --kernel-trace --output-format csv -d $D/trace_step -o r06s -- python3 bench.py --no-cpu-baseline --no-e2e --no-workloads --steps 1 --warmup 3 > /dev/null 2> $D/trace_step.err
python3 tools/step_timeline.py $D/trace_step/r06s_kernel_trace.csv > $D/step_timeline.txt 2>> $D/trace_step.err
# dispatches / kernel time / gaps per step (bench.py carries these as roofline.step_profile): a trace of the bench's own command
rocprofv3 --kernel-trace --output-format csv -d $D/trace_steps -o r06t -- python3 bench.py --no-cpu-baseline --no-e2e --no-workloads > $D/bench_under_trace.json 2> $D/trace_steps.err
python3 tools/step_profile.py $D/trace_steps/r06t_kernel_trace.csv $D/bench_under_trace.json $D/step_profile.json 10 >> $D/trace_steps.err 2>&1
rm -f $D/trace_steps/*kernel_trace.csv $D/trace_steps/*agent_info.csv
rocprofv3 --pmc SQ_WAVES --output-format csv -d $D/pmc_waves -o r06 -- python3 bench.py --no-cpu-baseline --no-e2e --no-workloads --steps 2 --warmup 1 > /dev/null 2> $D/pmc_waves.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $D/pmc_busy -o r06 -- python3 bench.py --no-cpu-baseline --no-e2e --no-workloads --steps 2 --warmup 1 > /dev/null 2> $D/pmc_busy.err
rm -f $D/trace_step/*agent_info.csv
rm -f $D/stats*/*kernel_trace.csv $D/stats*/*agent_info.csv $D/pmc_*/*agent_info.csv
ls -la $D $D/stats $D/stats_panel $D/pmc_fetch $D/pmc_fetch_panel | head -60
head -c 400 $D/bench_default.json
# the file -> result call (device front end + typing path) under the tracer: SAM text and coordinate-sorted BAM
rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats_fe_sam -o r06fe -- python3 tools/e2e_file.py 500000 0 > $D/fe_sam.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats_fe_bam -o r06fe -- python3 tools/e2e_bam.py 500000 > $D/fe_bam.log 2>&1
rm -f $D/stats_fe_*/*kernel_trace.csv $D/stats_fe_*/*agent_info.csv
# the stages of the calls as the library prints them (HGX_PARSE_PROFILE), untraced
HGX_PARSE_PROFILE=1 python3 tools/e2e_file.py 500000 0 > $D/fe_sam_stages.log 2>&1
HGX_PARSE_PROFILE=1 python3 tools/e2e_bam.py 500000 > $D/fe_bam_stages.log 2>&1
HGX_PARSE_PROFILE=1 python3 tools/prof_many_front.py > $D/fe_many_stages.log 2>&1
# the device inflate's two forms on the two BAM workloads (kernel times) and the default form's phase laps
bash tools/inf_ab.sh > $D/inflate_forms.txt 2>&1
INF_FORM=inflate_prof python3 tools/inflate_probe.py run 2>&1 | grep "k_bgzf_inflate_w\|call (with" | sort -u > $D/inflate_phases.txt
# ---- round 6 additions --------------------------------------------------------------------------------------------------------
# the multi-rank bench BODY on this one GPU (control plane over gloo; RCCL refuses two ranks on one device): configs1 with 2 ranks,
# class1 with 4 ranks (HLA-A's pairs sharded over ranks 0-1: device front end per shard, pileup all-reduce, class-table gather per step)
python3 bench.py --gpus 2 --backend gloo --share-gpu --pairs 100000 --steps 5 --warmup 2 --no-workloads > $D/bench_2ranks_shared_gpu.json 2> $D/bench_2ranks.err
python3 bench.py --gpus 4 --backend gloo --share-gpu --workload class1 --pairs 100000 --steps 5 --warmup 2 --check-unsharded --no-cpu-baseline > $D/bench_class1_4ranks_shared_gpu.json 2> $D/bench_class1_4ranks.err
# stream placement: what the hardware does with streams (queues per priority, lanes), and the A/B of the measured placement
python3 tools/stream_probe.py 12 HHHHHHHHHHHH > $D/stream_probe.txt 2>&1
python3 tools/stream_probe.py 12 HLHLHLHLLLLL >> $D/stream_probe.txt 2>&1
NF=3 python3 tools/stream_conflicts.py > $D/stream_conflicts.txt 2>&1
bash tools/stream_ab.sh > $D/stream_ab.txt 2>&1
# the device front end's size gate and the drop-in's stages
python3 tools/front_gate.py > $D/front_gate.txt 2>&1
python3 tools/dropin_profile.py > $D/dropin_profile.txt 2>&1
python3 tools/dropin_profile.py codis_10k >> $D/dropin_profile.txt 2>&1
# the drop-in call itself under the tracer: 23 typing() calls on the configs[0] fixture
mkdir -p $D/stats_dropin
rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats_dropin -o r06dropin -- python3 tools/dropin_calls.py 20 > $D/dropin_config0.log 2>&1
rm -f $D/stats_dropin/*kernel_trace.csv $D/stats_dropin/*agent_info.csv
# ... and the same class-I body with the ranks of the sharded locus exchanging through dist.RcclComm / the C-ABI collectives (a stand-in
# transport with RCCL's entry points for ranks that share one GPU: tests/fake_rccl)
/opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -shared -o /tmp/libfake_rccl.so tests/fake_rccl/fake_rccl.cpp
HGX_BENCH_RCCL_LIB=/tmp/libfake_rccl.so python3 bench.py --gpus 4 --backend gloo --share-gpu --comm rccl --workload class1 --pairs 100000 --steps 5 --warmup 2 --check-unsharded --no-cpu-baseline > $D/bench_class1_4ranks_rccl_entry_points.json 2> $D/bench_class1_4ranks_rccl.err
