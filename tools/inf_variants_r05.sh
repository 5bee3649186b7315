#!/bin/bash
# k_bgzf_inflate_w with other ring / flush / table-index / window-output sizes: builds libhgx variants under csrc/lab/var/ and times
# the kernel on the two inputs of tools/inflate_probe.py (the default build first)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/hisat-genotype_amd/csrc
mkdir -p $C/lab/var gpurun_out
VARS=("8192 2048 10 1024 9" "8192 2048 10 1024 10" "8192 2048 11 1024 10" "4096 1024 10 512 9")
for v in "${VARS[@]}"; do
  set -- $v
  n=$1_$2_$3_$4_$5
  [ -f $C/lab/var/libhgx_$n.so ] && continue
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $R/include -I $C -DHGX_INF_RING=$1 -DHGX_INF_FLUSH=$2 -DHGX_INF_LITP=$3 -DHGX_INF_TMAX=$4 -DHGX_INF_DISTP=$5 -c $C/hgx_inflate.hip -o $C/lab/var/inf_$n.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $C/lab/var/libhgx_$n.so $(ls $C/*.o | grep -v hgx_inflate.o) $C/lab/var/inf_$n.o -lz -ldl
done
[ "${BUILD_ONLY:-}" = 1 ] && exit 0
cd /tmp && export TMPDIR=/tmp && cd $R
for v in "" "${VARS[@]}"; do
  n=$(echo $v | tr ' ' '_')
  if [ -n "$n" ]; then export INF_LIB=$C/lab/var/libhgx_$n.so; fi
  rm -rf gpurun_out/inf
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/inf -o t -- python3 tools/inflate_probe.py run > gpurun_out/inf_var.log 2>&1
  echo "== ring flush lit_p t_max dist_p: ${v:-8192 2048 10 1024 8 (the product)}"; python3 tools/inflate_probe.py show gpurun_out/inf/t_kernel_trace.csv
done
rm -rf gpurun_out/inf
