// hgx_counts.hip -- per-allele sums over a class table (8a-7): Gene_counts (core:1700-1710: sum of the counts of the classes that
// contain the allele) and the first class containing each allele (the tie order of the reference's stable sorts), for the one-task
// path.  (The many-task path has its own task-aware kernel, k_many_counts in hgx_many.hip.)
#include <algorithm>
#include <cstdint>
#include <vector>

#include <hip/hip_runtime.h>

#include "hgx_common.hpp"

namespace {
__device__ __forceinline__ uint64_t lane_u64(uint64_t v, int l) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)v, l);
}
}   // namespace

// first class containing each compact allele = first set bit of its row in the transposed matrix (one wavefront per row)
__global__ __launch_bounds__(256) void k_first_set_rows(const uint64_t *__restrict__ BT, int n_rows, int c64, int32_t *__restrict__ first) {
    const int lane = threadIdx.x & 63;
    const long row = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (row >= n_rows) return;
    int best = 0x7fffffff;
    for (int w0 = 0; w0 < c64; w0 += 64) {
        const int w = w0 + lane;
        const uint64_t x = w < c64 ? BT[(size_t)row * c64 + w] : 0ull;
        const uint64_t hit = __ballot(x != 0ull);
        if (hit) {
            const int l = __builtin_ctzll(hit);
            const uint64_t xl = lane_u64(x, l);
            best = 64 * (w0 + l) + __builtin_ctzll(xl);
            break;
        }
    }
    if (lane == 0) first[row] = best == 0x7fffffff ? -1 : best;
}

// first class containing an allele, for a handful of alleles: one workgroup per allele walks the allele's bit column
__global__ __launch_bounds__(256) void k_first_classes(const uint64_t *__restrict__ B, int n_classes, int w64,
                                                       const int32_t *__restrict__ alleles, int32_t *__restrict__ first) {
    __shared__ int best;
    const int a = alleles[blockIdx.x];
    if (threadIdx.x == 0) best = n_classes;
    __syncthreads();
    const int word = a >> 6;
    const uint64_t bit = 1ull << (a & 63);
    for (int c0 = 0; c0 < n_classes; c0 += 256) {
        const int c = c0 + threadIdx.x;
        const bool hit = c < n_classes && (B[(size_t)c * w64 + word] & bit);
        if (__syncthreads_or(hit)) {
            if (hit) atomicMin(&best, c);
            __syncthreads();
            break;
        }
    }
    if (threadIdx.x == 0) first[blockIdx.x] = best < n_classes ? best : -1;
}

extern "C" int hgx_first_classes(const hgx_classes *c, const int32_t *alleles_host, int32_t n, int32_t *first_host, void *stream) {
    ARGCHK(c && n >= 0);
    if (n == 0) return HGX_OK;
    hgx_classes_order_after(c, (hipStream_t)stream);
    ARGCHK(alleles_host && first_host);
    for (int i = 0; i < n; ++i) ARGCHK(alleles_host[i] >= 0 && alleles_host[i] < c->a_pad);
    if (c->n_classes == 0) { for (int i = 0; i < n; ++i) first_host[i] = -1; return HGX_OK; }
    hipStream_t st = (hipStream_t)stream;
    DevBuf b_a, b_f;
    ALLOC(b_a, (size_t)n * 4); ALLOC(b_f, (size_t)n * 4);
    { int rc_ = hgx_h2d(b_a.p, alleles_host, (size_t)n * 4, st); if (rc_) return rc_; }
    hipLaunchKernelGGL(k_first_classes, dim3(n), dim3(256), 0, st, c->d_bits, c->n_classes, c->w64, b_a.as<int32_t>(), b_f.as<int32_t>());
    HIPCHK(hipGetLastError());
    { int rc_ = hgx_d2h(first_host, b_f.p, (size_t)n * 4, st); if (rc_) return rc_; }
    { int rc_ = hgx_sync(st); if (rc_) return rc_; }
    return HGX_OK;
}

// Gene_counts (typing_core.py:1187-1190): per allele the summed count of the classes containing it, and the first
// such class (dict insertion order for ties).
// Direct form, straight from the row-major class matrix (no transposed copy, no FP64 mat-vec passes): a lane owns one 32-bit
// half of an allele word and a wavefront walks a range of classes -- a 256-byte coalesced load per class, the class count
// wave-uniform -- keeping 32 integer column sums in registers (one v_bfe + one v_mad_u32_u24 per bit and class; 64-bit
// multiply-adds made the kernel 114 us instead of the mat-vec form's 190).  First
// classes: classes are visited in ascending order, so a bit that a lane sees for the first time records the class index
// (kept in LDS, touched only when some lane of the wave has a new bit).  The four waves of a workgroup take consecutive class
// ranges of the same words and are added / min-ed in LDS; the workgroup's partial columns go to memory with plain coalesced
// stores and a second small kernel adds the <= 64 partials per allele (a global atomic per column and workgroup instead --
// 1 M atomics on 7 168 hot addresses -- was most of a 101 us launch).  Integers: exact.  Needs every class count < 2^24 and
// their sum < 2^32 (checked on the device; otherwise the mat-vec form below runs).
constexpr int AC_WAVES = 4;
__global__ __launch_bounds__(64 * AC_WAVES) void k_allele_counts_direct(const uint64_t *__restrict__ bits, int C, int w64,
                                                                         const int64_t *__restrict__ count, int per_wave,
                                                                         unsigned long long *__restrict__ cnt_part,
                                                                         int *__restrict__ first_part, int A) {
    __shared__ int s_first[AC_WAVES][32][64];                 // [wave][bit][lane]
    __shared__ uint32_t s_cnt[AC_WAVES][32][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n_half = 2 * w64;
    const int hw = blockIdx.x * 64 + lane;                    // my 32-bit half-word of the allele row
    const bool live = hw < n_half;
    const int c0 = (blockIdx.y * AC_WAVES + wv) * per_wave, c1 = min(C, c0 + per_wave);
    const uint32_t *rows = (const uint32_t *)bits;
    uint32_t acc[32];
#pragma unroll
    for (int b = 0; b < 32; ++b) { acc[b] = 0u; s_first[wv][b][lane] = 0x7fffffff; }
    uint32_t seen = 0u;
    constexpr int U = 8;                                       // class rows in flight per wave (one wave per SIMD: no other latency hiding)
    uint32_t xn[U], nn[U];                                      // the next batch: requested before the current one is worked through
    auto fetch = [&](int cb) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = cb + u;
            xn[u] = (live && c < c1) ? rows[(size_t)c * n_half + hw] : 0u;
            nn[u] = c < c1 ? (uint32_t)count[c] : 0u;
        }
    };
    fetch(c0);
    for (int cb = c0; cb < c1; cb += U) {
        uint32_t xs[U], ns[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { xs[u] = xn[u]; ns[u] = nn[u]; }
        fetch(cb + U);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t x = xs[u];
            const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)ns[u]);
#pragma unroll
            for (int b = 0; b < 32; ++b) acc[b] = __umul24((x >> b) & 1u, n) + acc[b];      // v_bfe_u32 + v_mad_u32_u24
            uint32_t fresh = x & ~seen;
            if (__any(fresh != 0u)) {
                seen |= x;
                while (fresh) {
                    const int b = __builtin_ctz(fresh);
                    s_first[wv][b][lane] = cb + u;
                    fresh &= fresh - 1;
                }
            }
        }
    }
#pragma unroll
    for (int b = 0; b < 32; ++b) s_cnt[wv][b][lane] = acc[b];
    __syncthreads();
    // thread (wv, lane) finishes bits 8 wv .. 8 wv + 7 of every lane's half-word: add / min over the four waves
    if (live) {
#pragma unroll
        for (int k = 0; k < 32 / AC_WAVES; ++k) {
            const int b = wv * (32 / AC_WAVES) + k;
            unsigned long long t = 0ull;
            int f = 0x7fffffff;
#pragma unroll
            for (int q = 0; q < AC_WAVES; ++q) { t += (unsigned long long)s_cnt[q][b][lane]; f = min(f, s_first[q][b][lane]); }
            const int a = hw * 32 + b;
            cnt_part[(size_t)blockIdx.y * A + a] = t;
            first_part[(size_t)blockIdx.y * A + a] = f;
        }
    }
}
__global__ void k_allele_counts_reduce(const unsigned long long *__restrict__ cnt_part, const int *__restrict__ first_part, int groups,
                                       int A, int64_t *__restrict__ cnt, int *__restrict__ first) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= A) return;
    unsigned long long t = 0ull;
    int f = 0x7fffffff;
    for (int g0 = 0; g0 < groups; g0 += 8) {                  // eight partials in flight
        unsigned long long tv[8];
        int fv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const bool in = g0 + k < groups;
            tv[k] = in ? cnt_part[(size_t)(g0 + k) * A + a] : 0ull;
            fv[k] = in ? first_part[(size_t)(g0 + k) * A + a] : 0x7fffffff;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { t += tv[k]; f = min(f, fv[k]); }
    }
    cnt[a] = (int64_t)t;
    first[a] = f == 0x7fffffff ? -1 : f;
}
__global__ void k_allele_counts_check(const int64_t *__restrict__ count, int C, unsigned long long *__restrict__ chk) {
    // chk[0] = some count outside [0, 2^24), chk[1] = sum of the counts; 16 classes per thread: a few dozen atomics in all
    unsigned long long s = 0ull;
    bool bad = false;
    for (int k = 0; k < 16; ++k) {
        const long c = ((long)blockIdx.x * 16 + k) * blockDim.x + threadIdx.x;
        const long long n = c < C ? count[c] : 0;
        bad = bad || n < 0 || n >= (1ll << 24);
        s += (unsigned long long)(n < 0 ? 0 : n);
    }
    if (bad) chk[0] = 1ull;
    s = wave_sum_u64(s);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(&chk[1], s);
}

// The mat-vec form (two passes of the bit mat-vec over the transposed class matrix), for class counts beyond 32 bits.
// bitsT [a_pad][c64]: row a = the classes containing allele a
__global__ void k_allele_counts_wide(const uint64_t *__restrict__ bitsT, int a_pad, int c64, int n_classes, const int64_t *__restrict__ count,
                                     int64_t *__restrict__ out_count, int32_t *__restrict__ out_first) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= a_pad) return;
    int64_t sum = 0;
    int32_t first = -1;
    for (int w = 0; w < c64; ++w) {
        uint64_t m = bitsT[(size_t)a * c64 + w];
        while (m) {
            const int cls = w * 64 + __builtin_ctzll(m);
            m &= m - 1;
            if (cls >= n_classes) break;
            sum += count[cls];
            if (first < 0) first = cls;
        }
    }
    out_count[a] = sum;
    out_first[a] = first;
}

extern "C" int hgx_allele_counts_on(const hgx_classes *cc, int64_t *count_host, int32_t *first_host, void *stream);
extern "C" int hgx_allele_counts(const hgx_classes *cc, int64_t *count_host, int32_t *first_host) {
    return hgx_allele_counts_on(cc, count_host, first_host, nullptr);
}
extern "C" int hgx_allele_counts_on(const hgx_classes *cc, int64_t *count_host, int32_t *first_host, void *stream) {
    ARGCHK(cc && count_host && first_host);
    hipStream_t st = (hipStream_t)stream;
    hgx_classes_order_after(cc, st);
    hgx_classes *c = const_cast<hgx_classes *>(cc);
    const int A = c->a_pad;
    if (c->n_classes == 0) {
        for (int a = 0; a < A; ++a) { count_host[a] = 0; first_host[a] = -1; }
        return HGX_OK;
    }
    {
        const int C = c->n_classes;
        DevBuf b_cnt, b_first, b_big, b_cp, b_fp;
        ALLOC(b_cnt, (size_t)A * 8); ALLOC(b_first, (size_t)A * 4); ALLOC(b_big, 16);
        HIPCHK(hipMemsetAsync(b_big.p, 0, 16, st));
        hipLaunchKernelGGL(k_allele_counts_check, dim3(nblk(C, 256 * 16)), dim3(256), 0, st, c->d_count, C, b_big.as<unsigned long long>());
        // ~1024 wavefronts: 64 half-words each, four consecutive class ranges per workgroup
        const int spans = (2 * c->w64 + 63) / 64;
        const int groups = std::max(1, std::min(256 / std::max(spans, 1), (C + 64 * AC_WAVES - 1) / (64 * AC_WAVES)));
        const int per_wave = (C + groups * AC_WAVES - 1) / (groups * AC_WAVES);
        ALLOC(b_cp, (size_t)groups * A * 8); ALLOC(b_fp, (size_t)groups * A * 4);
        hipLaunchKernelGGL(k_allele_counts_direct, dim3(spans, groups), dim3(64 * AC_WAVES), 0, st, c->d_bits, C, c->w64, c->d_count,
                           per_wave, b_cp.as<unsigned long long>(), b_fp.as<int>(), A);
        hipLaunchKernelGGL(k_allele_counts_reduce, dim3(nblk(A, 64)), dim3(64), 0, st, b_cp.as<unsigned long long>(), b_fp.as<int>(),
                           groups, A, b_cnt.as<int64_t>(), b_first.as<int>());
        HIPCHK(hipGetLastError());
        unsigned long long big[2] = {0, 0};
        { int rc_ = hgx_d2h(count_host, b_cnt.p, (size_t)A * 8, st); if (rc_) return rc_; }
        { int rc_ = hgx_d2h(first_host, b_first.p, (size_t)A * 4, st); if (rc_) return rc_; }
        { int rc_ = hgx_d2h(big, b_big.p, 16, st); if (rc_) return rc_; }
        { int rc_ = hgx_sync(st); if (rc_) return rc_; }
        if (!big[0] && big[1] < (1ull << 32)) return HGX_OK;
    }
    int rc = hgx_ensure_transposed(c, st);
    if (rc) return rc;
    // class counts beyond 32 bits (hand-made class sets: a batch has fewer pairs): plain 64-bit sums, one allele per lane
    DevBuf b_c, b_i;
    ALLOC(b_c, (size_t)A * 8); ALLOC(b_i, (size_t)A * 4);
    hipLaunchKernelGGL(k_allele_counts_wide, dim3(nblk(A, 256)), dim3(256), 0, st, c->d_bitsT, A, c->c64, c->n_classes, c->d_count,
                       b_c.as<int64_t>(), b_i.as<int32_t>());
    HIPCHK(hipGetLastError());
    { int rc_ = hgx_d2h(count_host, b_c.p, (size_t)A * 8, st); if (rc_) return rc_; }
    { int rc_ = hgx_d2h(first_host, b_i.p, (size_t)A * 4, st); if (rc_) return rc_; }
    { int rc_ = hgx_sync(st); if (rc_) return rc_; }
    return HGX_OK;
}

