#!/bin/bash
# k_bgzf_inflate with other ring / flush sizes (LDS per wavefront = occupancy: 16 KB ring 7 wavefronts per CU, 8 KB 12, 4 KB 18,
# 2 KB 24): builds libhgx variants with -DHGX_INF_RING / -DHGX_INF_FLUSH under csrc/lab/var/ and times the kernel on the two inputs
# of tools/inflate_probe.py.  Round 4's result (profiles/r04_inflate_ring_sweep.txt): no size beats 8 KB / 2 KB.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/hisat-genotype_amd/csrc
mkdir -p $C/lab/var gpurun_out
for v in "16384 2048" "8192 1024" "4096 2048" "4096 1024" "2048 512"; do
  set -- $v
  [ -f $C/lab/var/libhgx_$1_$2.so ] && continue
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $R/include -I $C -DHGX_INF_RING=$1 -DHGX_INF_FLUSH=$2 -c $C/hgx_inflate.hip -o $C/lab/var/inf_$1_$2.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $C/lab/var/libhgx_$1_$2.so $(ls $C/*.o | grep -v hgx_inflate.o) $C/lab/var/inf_$1_$2.o -lz -ldl
done
cd /tmp && export TMPDIR=/tmp && cd $R
for v in "" 16384_2048 8192_1024 4096_2048 4096_1024 2048_512; do
  if [ -n "$v" ]; then export INF_LIB=$C/lab/var/libhgx_$v.so; fi
  rm -rf gpurun_out/inf
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/inf -o t -- python3 tools/inflate_probe.py run > gpurun_out/inf_$v.log 2>&1
  echo "== ring_flush ${v:-8192_2048 (the product)}"; python3 tools/inflate_probe.py show gpurun_out/inf/t_kernel_trace.csv
done
