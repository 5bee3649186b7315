cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_inflate.py -x -q 2>&1 | tail -3
rm -rf gpurun_out/inf
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/inf -o t -- python3 tools/inflate_probe.py run > gpurun_out/inf_one.log 2>&1
tail -2 gpurun_out/inf_one.log | cut -c1-150; python3 tools/inflate_probe.py show gpurun_out/inf/t_kernel_trace.csv
