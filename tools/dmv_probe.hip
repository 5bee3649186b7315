// dmv_probe.hip -- is a K-COMPLETE direct bit mat-vec (no LUT slabs, no slab partials, no ticket, no combine) shorter per launch than
// k_lutmatvec's 13-14 us at the EM #1 shape of BASELINE configs[1] (16 098 classes x 4 549 active alleles)?  Standalone probe:
//   rows:  w_c = n_c / sum_a M[c][a] x_a     one workgroup per tile of 64 classes, its 16 wavefronts split the alleles, LDS reduce
//   cols:  t_a = sum_c M[c][a] w_c           one workgroup per tile of 16 alleles, lane = (allele, quarter of the classes), 16 wavefronts split each quarter
// The wave-uniform mask word of a step is a scalar load; the add is fma(b, x, acc) with b = 1.0 / 0.0 picked by ONE v_cndmask.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/dmv_probe tools/dmv_probe.hip ; run: tools/bin/dmv_probe [density]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int C = 16098, CP = 16128, A = 4549, AP = 4608;
constexpr int NW = 16;

// R[ct][a]: bit l = class 64 ct + l contains allele a
__global__ __launch_bounds__(1024) void k_rows(const uint64_t *__restrict__ R, const double *__restrict__ x, const long long *__restrict__ cnt,
                                                double *__restrict__ w) {
    __shared__ double sh[NW][64];
    const int ct = blockIdx.x, lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int a0 = wv * (AP / NW);
    const uint64_t *Rr = R + (size_t)ct * AP + a0;
    const double *xx = x + a0;
    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll 8
    for (int j = 0; j < AP / NW; j += 2) {
        const uint64_t m0 = Rr[j], m1 = Rr[j + 1];
        const double x0 = xx[j], x1 = xx[j + 1];
        const double b0 = __builtin_amdgcn_inverse_ballot_w64(m0) ? 1.0 : 0.0;
        const double b1 = __builtin_amdgcn_inverse_ballot_w64(m1) ? 1.0 : 0.0;
        acc0 = fma(b0, x0, acc0);
        acc1 = fma(b1, x1, acc1);
    }
    sh[wv][lane] = acc0 + acc1;
    __syncthreads();
    if (threadIdx.x < 64) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < NW; ++k) s += sh[k][lane];
        const int c = ct * 64 + lane;
        if (c < C) w[c] = s > 0.0 ? (double)cnt[c] / s : 0.0;
        else if (c < CP) w[c] = 0.0;
    }
}

// Q[t][j]: step j of allele tile t (16 alleles): bits [16 q, 16 q + 16) = alleles of the tile in class q * (CP / 4) + j
__global__ __launch_bounds__(1024) void k_cols(const uint64_t *__restrict__ Q, const double *__restrict__ w, const double *__restrict__ x,
                                                double *__restrict__ y, double *__restrict__ part_tot) {
    extern __shared__ double ws[];                 // w, all CP classes
    __shared__ double sh[NW][64];
    const int t = blockIdx.x, lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int c = threadIdx.x; c < CP; c += 1024) ws[c] = w[c];
    __syncthreads();
    constexpr int QK = CP / 4, STEPS = QK / NW;    // 4032 classes per quarter, 252 steps per wavefront
    const int q = lane >> 4;
    const uint64_t *Qr = Q + (size_t)t * QK + wv * STEPS;
    const double *wl = ws + q * QK + wv * STEPS;
    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll 6
    for (int j = 0; j < STEPS; j += 2) {
        const uint64_t m0 = Qr[j], m1 = Qr[j + 1];
        const double b0 = __builtin_amdgcn_inverse_ballot_w64(m0) ? 1.0 : 0.0;
        const double b1 = __builtin_amdgcn_inverse_ballot_w64(m1) ? 1.0 : 0.0;
        acc0 = fma(b0, wl[j], acc0);
        acc1 = fma(b1, wl[j + 1], acc1);
    }
    sh[wv][lane] = acc0 + acc1;
    __syncthreads();
    if (threadIdx.x < 16) {
        double s = 0.0;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
#pragma unroll
            for (int k = 0; k < NW; ++k) s += sh[k][qq * 16 + lane];
        const int a = t * 16 + lane;
        const double v = a < A ? x[a] * s : 0.0;
        if (a < AP) y[a] = v;
        double tot = v;
        for (int off = 8; off; off >>= 1) tot += __shfl_xor(tot, off, 16);
        if (lane == 0) part_tot[t] = tot;
    }
}

__global__ void k_norm(double *y, const double *part_tot, int n_part) {       // stands in for the consumer's normalisation
    __shared__ double tot;
    if (threadIdx.x == 0) { double s = 0.0; for (int k = 0; k < n_part; ++k) s += part_tot[k]; tot = s; }
    __syncthreads();
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a < AP) y[a] = y[a] / tot;
}
__global__ void k_empty(double *p) { if (p && threadIdx.x == 9999) p[0] = 0; }

int main(int argc, char **argv) {
    const double dens = argc > 1 ? atof(argv[1]) : 0.3;
    std::vector<uint64_t> R((size_t)(CP / 64) * AP, 0), Q((size_t)(AP / 16) * (CP / 4), 0);
    std::vector<uint8_t> M((size_t)C * A);
    srand(7);
    for (auto &b : M) b = (rand() / (double)RAND_MAX) < dens;
    for (int c = 0; c < C; ++c)
        for (int a = 0; a < A; ++a)
            if (M[(size_t)c * A + a]) {
                R[(size_t)(c / 64) * AP + a] |= 1ull << (c % 64);
                const int q = c / (CP / 4), j = c % (CP / 4);
                Q[(size_t)(a / 16) * (CP / 4) + j] |= 1ull << (16 * q + (a % 16));
            }
    std::vector<double> x(AP, 0.0), wref(CP, 0.0), yref(AP, 0.0);
    std::vector<long long> cnt(CP, 0);
    for (int a = 0; a < A; ++a) x[a] = 1.0 / A;
    for (int c = 0; c < C; ++c) cnt[c] = 1 + rand() % 50;
    uint64_t *dR, *dQ; double *dx, *dw, *dy, *dpt; long long *dc;
    CHK(hipMalloc(&dR, R.size() * 8)); CHK(hipMalloc(&dQ, Q.size() * 8)); CHK(hipMalloc(&dx, AP * 8)); CHK(hipMalloc(&dw, CP * 8));
    CHK(hipMalloc(&dy, AP * 8)); CHK(hipMalloc(&dpt, 512 * 8)); CHK(hipMalloc(&dc, CP * 8));
    CHK(hipMemcpy(dR, R.data(), R.size() * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(dQ, Q.data(), Q.size() * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dx, x.data(), AP * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(dc, cnt.data(), CP * 8, hipMemcpyHostToDevice));
    CHK(hipFuncSetAttribute((const void *)k_cols, hipFuncAttributeMaxDynamicSharedMemorySize, CP * 8));
    hipStream_t st; CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    // correctness of one application against the host
    k_rows<<<CP / 64, 1024, 0, st>>>(dR, dx, dc, dw);
    k_cols<<<AP / 16, 1024, CP * 8, st>>>(dQ, dw, dx, dy, dpt);
    CHK(hipStreamSynchronize(st));
    std::vector<double> w(CP), y(AP);
    CHK(hipMemcpy(w.data(), dw, CP * 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(y.data(), dy, AP * 8, hipMemcpyDeviceToHost));
    double maxe = 0.0;
    for (int c = 0; c < C; ++c) { double s = 0; for (int a = 0; a < A; ++a) if (M[(size_t)c * A + a]) s += x[a]; wref[c] = s > 0 ? cnt[c] / s : 0; maxe = fmax(maxe, fabs(wref[c] - w[c]) / fmax(1e-300, fabs(wref[c]))); }
    for (int a = 0; a < A; ++a) { double s = 0; for (int c = 0; c < C; ++c) if (M[(size_t)c * A + a]) s += wref[c]; yref[a] = x[a] * s; maxe = fmax(maxe, fabs(yref[a] - y[a]) / fmax(1e-300, fabs(yref[a]))); }
    printf("one application vs host: max relative error %.3g\n", maxe);
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    auto time_it = [&](const char *what, int n, auto body) {
        for (int i = 0; i < 20; ++i) body();
        CHK(hipStreamSynchronize(st));
        CHK(hipEventRecord(e0, st));
        for (int i = 0; i < n; ++i) body();
        CHK(hipEventRecord(e1, st));
        CHK(hipStreamSynchronize(st));
        float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-58s %8.2f us per iteration\n", what, ms * 1e3 / n);
    };
    time_it("empty 252-WG x 1024 kernel (launch floor)", 500, [&] { k_empty<<<252, 1024, 0, st>>>(nullptr); });
    time_it("rows only", 500, [&] { k_rows<<<CP / 64, 1024, 0, st>>>(dR, dx, dc, dw); });
    time_it("cols only", 500, [&] { k_cols<<<AP / 16, 1024, CP * 8, st>>>(dQ, dw, dx, dy, dpt); });
    time_it("rows + cols (one application of the map)", 500, [&] { k_rows<<<CP / 64, 1024, 0, st>>>(dR, dx, dc, dw); k_cols<<<AP / 16, 1024, CP * 8, st>>>(dQ, dw, dx, dy, dpt); });
    time_it("rows + cols + norm (x <- map(x))", 500, [&] { k_rows<<<CP / 64, 1024, 0, st>>>(dR, dx, dc, dw); k_cols<<<AP / 16, 1024, CP * 8, st>>>(dQ, dw, dx, dy, dpt);
                                                              k_norm<<<AP / 256, 256, 0, st>>>(dy, dpt, AP / 16); });
    return 0;
}
