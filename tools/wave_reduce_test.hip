// The register-file butterfly reductions and the 64 x 64 bit transpose of hgx_common.hpp against the ds_bpermute (__shfl_xor) /
// ballot forms they replaced: bit-identical on random inputs.  hipcc --offload-arch=gfx950 -I include -I hisat-genotype_amd/csrc -o /tmp/wrt tools/wave_reduce_test.hip && /tmp/wrt
#include "hgx_common.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>

extern "C" void hgx_set_error(const char *, ...) {}

__global__ void k(uint64_t *out, const double *x, const uint64_t *u) {
    const int l = threadIdx.x;
    double v = x[l], s = v, m = v;
    uint64_t w = u[l], t = w;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        s += __shfl_xor(s, d, 64);
        m = fmax(m, __shfl_xor(m, d, 64));
        t += shfl_xor_u64(t, d);
    }
    out[l] = (uint64_t)__double_as_longlong(s);
    out[64 + l] = (uint64_t)__double_as_longlong(wave_sum_f64(v));
    out[128 + l] = (uint64_t)__double_as_longlong(m);
    out[192 + l] = (uint64_t)__double_as_longlong(wave_max_nonneg_f64(v));
    out[256 + l] = t;
    out[320 + l] = wave_sum_u64(w);
    uint64_t mine = 0;                      // 64 x 64 bit transpose: ballots against the block-swap form
    for (int b = 0; b < 64; ++b) {
        const uint64_t col = __ballot((w >> b) & 1ull);
        if (l == b) mine = col;
    }
    out[384 + l] = mine;
    out[448 + l] = wave_transpose64(w);
}

int main() {
    double *dx, hx[64];
    uint64_t *du, *dout, hu[64], ho[512];
    if (hipMalloc(&dx, 64 * 8) != hipSuccess || hipMalloc(&du, 64 * 8) != hipSuccess || hipMalloc(&dout, 512 * 8) != hipSuccess) return 2;
    int bad = 0;
    srand(7);
    for (int t = 0; t < 2000; ++t) {
        for (int i = 0; i < 64; ++i) {
            hx[i] = (t % 3 == 0 ? 1.0 : 1e-9) * (double)rand() / RAND_MAX * ((t & 1) && (i % 5 == 0) ? 0.0 : 1.0) + (t % 7 == 0 ? i * 1e3 : 0.0);
            hu[i] = ((uint64_t)rand() << 42) ^ ((uint64_t)rand() << 21) ^ (uint64_t)rand();
        }
        (void)hipMemcpy(dx, hx, 64 * 8, hipMemcpyHostToDevice);
        (void)hipMemcpy(du, hu, 64 * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dout, dx, du);
        if (hipMemcpy(ho, dout, 512 * 8, hipMemcpyDeviceToHost) != hipSuccess) return 2;
        for (int q = 0; q < 4; ++q)
            if (memcmp(ho + 128 * q, ho + 128 * q + 64, 64 * 8) != 0) { if (bad < 5) printf("mismatch t=%d kind=%d\n", t, q); ++bad; }
    }
    printf("bad=%d of 8000\n", bad);
    return bad != 0;
}
