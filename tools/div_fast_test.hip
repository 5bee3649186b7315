// tools/div_fast_test.hip -- is the quotient k_emx forms (csrc/hgx_emx.hip: reciprocal refined once per denominator, then
// q0 = x * r, e = fma(-s, q0, x), q = fma(e, r, q0)) the SAME double as the compiler's correctly rounded x / s wherever the
// kernel takes that path (0 or 2^-600 <= x <= 2^600, 2^-200 <= s <= 2^200)?  Random and adversarial operands; prints mismatches.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/div_fast_test tools/div_fast_test.hip && tools/bin/div_fast_test
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#pragma clang fp contract(off)
__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31;
    return x;
}
__device__ __forceinline__ double make(uint64_t bits, int emin, int emax, int mode) {
    uint64_t man = bits & ((1ull << 52) - 1);
    const int e = emin + (int)((bits >> 52) % (uint64_t)(emax - emin + 1));
    if (mode == 1) man = ((1ull << 52) - 1) ^ (bits >> 60);            // mantissa of (nearly) all ones
    if (mode == 2) man = bits >> 60;                                    // just above a power of two
    if (mode == 3) man &= ~((1ull << 26) - 1);                          // short mantissas (small integers, simple fractions)
    return __longlong_as_double((long long)(((uint64_t)(e + 1023) << 52) | man));
}
__global__ void k_test(uint64_t seed, unsigned long long *bad, unsigned long long *n_done, double *ex) {
    const uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long nb = 0;
    for (int it = 0; it < 4096; ++it) {
        const uint64_t a = mix(seed + id * 4096 + it), b = mix(a + 0x9e3779b97f4a7c15ull), c = mix(b ^ seed);
        const int mode = (int)(c & 3), mode2 = (int)((c >> 2) & 3);
        const double s = make(a, -200, 200, mode);
        double x = make(b, -600, 600, mode2);
        if (((c >> 4) & 15) == 0) x = (double)(1 + (c >> 40) % 300) * make(b, -60, 0, mode2);     // count * prob, as the EM forms it
        if (((c >> 8) & 255) == 0) x = 0.0;
        const double r0 = __builtin_amdgcn_rcp(s);
        const double e0 = __builtin_fma(-s, r0, 1.0);
        const double r1 = __builtin_fma(r0, e0, r0);
        const double e1 = __builtin_fma(-s, r1, 1.0);
        const double r2 = __builtin_fma(r1, e1, r1);
        const double q0 = x * r2;
        const double e = __builtin_fma(-s, q0, x);
        const double q = __builtin_fma(e, r2, q0);
        const double want = x / s;
        if (__double_as_longlong(q) != __double_as_longlong(want)) {
            if (nb == 0 && atomicAdd(bad, 0ull) == 0) { ex[0] = x; ex[1] = s; ex[2] = q; ex[3] = want; }
            ++nb;
        }
    }
    if (nb) atomicAdd(bad, nb);
    atomicAdd(n_done, 4096ull);
}
int main() {
    unsigned long long *d, h[2] = {0, 0};
    double *dex, hex[4] = {0, 0, 0, 0};
    hipMalloc(&d, 16); hipMalloc(&dex, 32);
    hipMemset(d, 0, 16); hipMemset(dex, 0, 32);
    for (int round = 0; round < 16; ++round) k_test<<<4096, 256>>>(0x1234567ull + round * 0x100000000ull, d, d + 1, dex);
    hipDeviceSynchronize();
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    hipMemcpy(hex, dex, 32, hipMemcpyDeviceToHost);
    printf("div_fast_test: %llu quotients, %llu differ from x / s\n", h[1], h[0]);
    if (h[0]) printf("  first: x = %a, s = %a: fast %a, x / s %a\n", hex[0], hex[1], hex[2], hex[3]);
    return h[0] ? 1 : 0;
}
