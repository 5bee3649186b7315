#!/bin/bash
# k_bam_walk with other range sizes: needs a build whose W_all line reads the size from HGX_WALK_RANGE (a one-line lab edit of
# hgx_front.hip bam_lines_dev); round 4's result is in that function's comment.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_front.py -x -q -k "bam or BAM" 2>&1 | tail -2
for r in 49152 16384 8192 4096 2048; do
  export HGX_WALK_RANGE=$r; rm -rf gpurun_out/wk
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/wk -o t -- python3 tools/e2e_bam.py 500000 > gpurun_out/wk.log 2>&1
  echo "range $r: $(grep 'k_bam_walk<0>' gpurun_out/wk/t_kernel_stats.csv | awk -F, '{print $(NF-4)}') ns avg walk<0>; $(grep 'k_bam_walk<1>' gpurun_out/wk/t_kernel_stats.csv | awk -F, '{print $(NF-4)}') walk<1>; $(grep '^run 3' gpurun_out/wk.log)"
done
