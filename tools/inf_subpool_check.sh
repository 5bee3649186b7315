#!/bin/bash
# k_bgzf_inflate_w with a second-level pool of 32 / 64 entries instead of 512: most long codes no longer fit and go through the
# wave-uniform path -- the fuzzer must not notice (tools/fuzz_inflate.py against zlib)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/hisat-genotype_amd/csrc
mkdir -p $C/lab/var
for n in 32 64; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $R/include -I $C -DHGX_INF_SUBPOOL=$n -c $C/hgx_inflate.hip -o $C/lab/var/inf_sub$n.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $C/lab/var/libhgx_sub$n.so $(ls $C/*.o | grep -v hgx_inflate.o) $C/lab/var/inf_sub$n.o -lz -ldl
  [ "${BUILD_ONLY:-}" = 1 ] && continue
  echo "== pool of $n entries"; INF_LIB=$C/lab/var/libhgx_sub$n.so python3 tools/fuzz_inflate.py ${1:-1500} 333 | tail -1
done
