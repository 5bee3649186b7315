#!/bin/bash
# the device BGZF inflate's forms on the two BAM workloads (1 M reads in one file; 64 files of 10 000 reads): kernel times from rocprofv3
#   default = k_bgzf_inflate_w (lane-parallel symbol loop), inflate_v1 = round 4's kernel
set -u
R=$(pwd)
mkdir -p gpurun_out/inf_ab
cd /tmp && export TMPDIR=/tmp && cd $R
for form in default inflate_v1; do
  if [ $form = default ]; then unset INF_FORM; else export INF_FORM=$form; fi
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/inf_ab/$form -o t -- python3 tools/inflate_probe.py run > gpurun_out/inf_ab/$form.log 2>&1
  echo "== $form"; grep "call (with" gpurun_out/inf_ab/$form.log
  python3 tools/inflate_probe.py show gpurun_out/inf_ab/$form/t_kernel_trace.csv
  rm -rf gpurun_out/inf_ab/$form
done
