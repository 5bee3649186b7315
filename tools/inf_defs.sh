#!/bin/bash
# k_bgzf_inflate_w built with other -D settings (one quoted argument per variant, e.g. "-DHGX_INF_LITP=9 -DHGX_INF_SUBPOOL=256"):
# the kernel's time on the two inputs of tools/inflate_probe.py, the product build first.  BUILD_ONLY=1 stops after the builds.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/hisat-genotype_amd/csrc
mkdir -p $C/lab/var gpurun_out
i=0
for v in "$@"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $R/include -I $C $v -c $C/hgx_inflate.hip -o $C/lab/var/inf_d$i.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $C/lab/var/libhgx_d$i.so $(ls $C/*.o | grep -v hgx_inflate.o) $C/lab/var/inf_d$i.o -lz -ldl
done
[ "${BUILD_ONLY:-}" = 1 ] && exit 0
cd /tmp && export TMPDIR=/tmp && cd $R
i=0
for v in "" "$@"; do
  if [ -n "$v" ]; then i=$((i+1)); export INF_LIB=$C/lab/var/libhgx_d$i.so; fi
  rm -rf gpurun_out/inf
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/inf -o t -- python3 tools/inflate_probe.py run > gpurun_out/inf_var.log 2>&1
  echo "== ${v:-the product build}"; python3 tools/inflate_probe.py show gpurun_out/inf/t_kernel_trace.csv
  if [ -n "${WITH_PROF:-}" ]; then INF_FORM=inflate_prof python3 tools/inflate_probe.py run 2>&1 | grep k_bgzf | sort -u | tail -2; fi
done
rm -rf gpurun_out/inf
