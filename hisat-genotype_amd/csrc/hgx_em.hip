// hgx_em.hip -- 8a-8: SQUAREM-accelerated EM (single_abundance, typing_common.py:1282-1410) on gfx950, FP64.
//
//   T(p)_a  proportional to  p_a * sum_{c contains a, s_c > 0} n_c / s_c,     s_c = sum_{b in c, b present} p_b
//
// Both halves of T are products of a 0/1 matrix with a dense FP64 vector:
//   rows pass   s_c = sum_a B[c][a] p_a      over the class matrix      B  [C][a_pad bits]
//   cols pass   t_a = sum_c B[c][a] w_c      over its transpose         Bt [a_pad][C bits]
// k_bitmatvec keeps the VECTOR in registers (thread t of a 1024-thread workgroup owns elements
// t, t+1024, ...: exactly the bits `lane` of words w, w+16, ... of every matrix row for wave w), streams
// matrix rows as wave-uniform 64-bit words (scalar loads) and reduces eight rows at a time with a
// transposed butterfly (10 cross-lane steps per 8 rows instead of 48).  The matrix is read once per pass
// (C * a_pad / 8 bytes); nothing else moves.  No MFMA: the contraction is 1 bit x FP64 and changes matrix
// every locus, so it is bound by streaming the bit matrix, not by FLOPs.
//
// The host enqueues iterations in batches without waiting: every kernel starts by reading a device-side
// `done` word, so iterations queued past convergence fall through in a few microseconds.
#include <algorithm>
#include <vector>

#include "hgx_common.hpp"

namespace {

// state words (double) shared by the EM kernels
enum {
    S_TOT_A = 0,     // sum of the vector the last normalising rows pass consumed
    S_FLAG = 1,      // 1 = SQUAREM extrapolation happened this iteration (sum v^2 > 0)
    S_DIFF = 2,
    S_KEYERR = 3,
    S_DONE = 4,
    S_ITER = 5,
    S_NROWS = 6,     // rows passes that really ran (not gated / past convergence)
    S_NCOLS = 7,
    S_N = 8
};

constexpr int BLOCK = 1024;
constexpr int NWAVE = BLOCK / 64;
constexpr int RB = 8;            // rows reduced together

__device__ __forceinline__ double block_sum(double v, double *sh) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    v = wave_sum_f64(v);
    __syncthreads();
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < NWAVE; ++i) t += sh[i];
    return t;
}
__device__ __forceinline__ double block_max(double v, double *sh) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m, 64));
    __syncthreads();
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    double t = sh[0];
#pragma unroll
    for (int i = 1; i < NWAVE; ++i) t = fmax(t, sh[i]);
    return t;
}

enum { MODE_ROWS = 0, MODE_COLS = 1, MODE_SUM = 2, MODE_MIN = 3 };

// acc (op)= x on the lanes whose bit is set in the wave-uniform 64-bit `word`: the word goes straight into
// EXEC, so one matrix word costs ONE vector instruction (plus two scalar ones) instead of shift/and/select.
__device__ __forceinline__ void masked_add(double &acc, double x, uint64_t word) {
    uint64_t saved;
    asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\tv_add_f64 %[a], %[a], %[x]\n\ts_mov_b64 exec, %[sv]"
                 : [a] "+v"(acc), [sv] "=&s"(saved)
                 : [x] "v"(x), [m] "s"(word)
                 : "scc");
}
__device__ __forceinline__ void masked_min(double &acc, double x, uint64_t word) {
    uint64_t saved;
    asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\tv_min_f64 %[a], %[a], %[x]\n\ts_mov_b64 exec, %[sv]"
                 : [a] "+v"(acc), [sv] "=&s"(saved)
                 : [x] "v"(x), [m] "s"(word)
                 : "scc");
}

template <bool MIN>
__device__ __forceinline__ double comb(double a, double b) { return MIN ? fmin(a, b) : a + b; }

// reduce RB=8 per-lane partials over the 64 lanes of a wave; afterwards every lane holds the result of
// row ((lane>>3)&1)*4 + ((lane>>4)&1)*2 + ((lane>>5)&1)
template <bool MIN>
__device__ __forceinline__ double reduce8(const double (&v)[RB], int lane) {
    double u[4], t[2], s;
    const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double mine = b5 ? v[2 * i + 1] : v[2 * i];
        const double other = b5 ? v[2 * i] : v[2 * i + 1];
        u[i] = comb<MIN>(mine, __shfl_xor(other, 32, 64));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const double mine = b4 ? u[2 * i + 1] : u[2 * i];
        const double other = b4 ? u[2 * i] : u[2 * i + 1];
        t[i] = comb<MIN>(mine, __shfl_xor(other, 16, 64));
    }
    {
        const double mine = b3 ? t[1] : t[0];
        const double other = b3 ? t[0] : t[1];
        s = comb<MIN>(mine, __shfl_xor(other, 8, 64));
    }
    s = comb<MIN>(s, __shfl_xor(s, 4, 64));
    s = comb<MIN>(s, __shfl_xor(s, 2, 64));
    s = comb<MIN>(s, __shfl_xor(s, 1, 64));
    return s;
}

constexpr int MAX_RPB = 64;      // matrix rows per workgroup

// y = f(B x) for a bit matrix B [n_rows][n_words * 64 bits] (n_words = row stride, a multiple of 8, zero padded)
// and a dense vector x [n_k].
// A workgroup owns rows_per_block consecutive rows and walks the K dimension in chunks of KPT*1024 elements;
// inside a chunk x lives in registers and matrix words are wave-uniform scalar loads.
//   MODE_ROWS: x = vec (raw) or vec / sum(vec) (normalise) or all-ones (init);  y[c] = s > 0 ? count[c] / s : 0
//   MODE_COLS: x = w;  t = B x;  q_out[a] = pres_in[a] ? p_a * t / len[a] : 0 with p_a = q_in[a] / *tot
//              (init: q_out = t / len),  pres_out[a] = pres_in[a] && t > 0
//   MODE_SUM : x = (double)count (exact below 2^53), y = B x            -> Gene_counts
//   MODE_MIN : x = element index, y = min over set bits (+inf if none)  -> first class containing the allele
template <int KPT, int MODE>
__global__ __launch_bounds__(BLOCK) void k_bitmatvec(const uint64_t *__restrict__ B, int n_rows, int n_words, int n_k,
                                                     int rows_per_block, const double *__restrict__ vec,
                                                     const uint8_t *__restrict__ vec_pres,
                                                     int x_mode /* rows: 0 raw, 1 normalise, 2 ones */,
                                                     const int64_t *__restrict__ count, const double *__restrict__ q_in,
                                                     const uint8_t *__restrict__ pres_in, const double *__restrict__ len,
                                                     double *__restrict__ y, uint8_t *__restrict__ pres_out,
                                                     double *__restrict__ scal, int gate /* 0 always, 1 needs S_FLAG */) {
    constexpr bool MIN = MODE == MODE_MIN;
    __shared__ double sh[NWAVE];
    __shared__ double part[MAX_RPB][NWAVE];
    if (MODE == MODE_ROWS || MODE == MODE_COLS) {
        if (scal[S_DONE] != 0.0) return;
        if (gate && scal[S_FLAG] == 0.0) return;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * rows_per_block;
    const int nrow = min(rows_per_block, n_rows - row0);
    if (nrow <= 0) return;
    // every (row, wave) slot of `part` is owned by one lane of that wave: no cross-wave traffic before the final barrier
    for (int i = tid; i < MAX_RPB * NWAVE; i += BLOCK) (&part[0][0])[i] = MIN ? __builtin_inf() : 0.0;
    double tot = 1.0;
    if (MODE == MODE_ROWS && x_mode == 1) {
        double s = 0.0;
        for (int e = tid; e < n_k; e += BLOCK) if (vec_pres[e]) s += vec[e];
        tot = block_sum(s, sh);
    }
    if (MODE == MODE_ROWS && blockIdx.x == 0 && tid == 0) { scal[S_TOT_A] = tot; scal[S_NROWS] += 1.0; }
    if (MODE == MODE_COLS && blockIdx.x == 0 && tid == 0) scal[S_NCOLS] += 1.0;
    if (MODE == MODE_COLS) tot = scal[S_TOT_A];
    __syncthreads();
    // Wave w owns KPT consecutive 64-bit words of every row chunk (elements 64*(KPT*w + k) + lane), fetched with
    // one 64-byte scalar load per 8 words.  No validity tests in the hot loop: addresses are clamped into the
    // (8-word padded, zero filled) row and out-of-range elements carry x = 0 (+inf for MIN), so whatever bits a
    // clamped load returns contribute nothing.
    struct W8 { uint64_t w[8]; };
    for (int k0 = 0; k0 < n_k; k0 += KPT * BLOCK) {
        double x[KPT];
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            const int e = k0 + 64 * (KPT * wv + k) + lane;
            double v = MIN ? __builtin_inf() : 0.0;
            if (e < n_k) {
                if (MODE == MODE_ROWS) v = (x_mode == 2) ? 1.0 : (vec_pres[e] ? vec[e] / tot : 0.0);
                else if (MODE == MODE_COLS) v = vec[e];
                else if (MODE == MODE_SUM) v = (double)count[e];
                else v = (double)e;
            }
            x[k] = v;
        }
        const int wbase = min((k0 >> 6) + KPT * wv, n_words - KPT);
        for (int rb = 0; rb < nrow; rb += RB) {
            double acc[RB];
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                acc[r] = MIN ? __builtin_inf() : 0.0;
                const uint64_t *brow = B + (size_t)min(row0 + rb + r, n_rows - 1) * n_words + wbase;
#pragma unroll
                for (int h = 0; h < KPT / 8; ++h) {
                    const W8 ww = *reinterpret_cast<const W8 *>(brow + 8 * h);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        if (MIN) masked_min(acc[r], x[8 * h + k], ww.w[k]);
                        else masked_add(acc[r], x[8 * h + k], ww.w[k]);
                    }
                }
            }
            const double s = reduce8<MIN>(acc, lane);
            if ((lane & 7) == 0) {
                const int r = rb + ((lane >> 3) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 5) & 1);
                part[r][wv] = comb<MIN>(part[r][wv], s);
            }
        }
    }
    __syncthreads();
    if (tid < nrow) {
        double t = part[tid][0];
#pragma unroll
        for (int i = 1; i < NWAVE; ++i) t = comb<MIN>(t, part[tid][i]);
        const int row = row0 + tid;
        if (MODE == MODE_ROWS) {
            y[row] = t > 0.0 ? (double)count[row] / t : 0.0;
        } else if (MODE == MODE_COLS) {
            const bool init = x_mode == 2;
            const bool in = init || pres_in[row];
            double v = 0.0;
            if (in && t > 0.0) {
                v = init ? t : (q_in[row] / tot) * t;
                if (len) v = v / len[row];
            }
            y[row] = v;
            pres_out[row] = (in && t > 0.0) ? 1 : 0;
        } else {
            y[row] = t;
        }
    }
}

// SQUAREM extrapolation (common:1361-1380).  p = pq (normalised already), p1 = q1/sum(q1), p2 = q2/sum(q2).
// Writes q2 <- max(0, p - 2 g r + g^2 v) (used raw by the third map) when sum v^2 > 0.
__global__ __launch_bounds__(BLOCK) void k_em_squarem(const double *__restrict__ p, const uint8_t *__restrict__ pres,
                                                      const double *__restrict__ q1, const uint8_t *__restrict__ pres1,
                                                      double *__restrict__ q2, uint8_t *__restrict__ pres2, int a_pad,
                                                      double *__restrict__ scal) {
    __shared__ double sh[NWAVE];
    if (scal[S_DONE] != 0.0) return;
    double s1 = 0.0, s2 = 0.0;
    for (int a = threadIdx.x; a < a_pad; a += BLOCK) {
        if (pres1[a]) s1 += q1[a];
        if (pres2[a]) s2 += q2[a];
    }
    const double tot1 = block_sum(s1, sh), tot2 = block_sum(s2, sh);
    double sr = 0.0, sv = 0.0, key = 0.0;
    for (int a = threadIdx.x; a < a_pad; a += BLOCK) {
        if (!pres[a]) continue;
        if (!pres1[a] || !pres2[a]) { key = 1.0; continue; }
        const double p1 = q1[a] / tot1, p2 = q2[a] / tot2;
        const double r = p1 - p[a];
        const double v = p2 - p1 - r;
        sr += r * r;
        sv += v * v;
    }
    const double tsr = block_sum(sr, sh), tsv = block_sum(sv, sh), tkey = block_sum(key, sh);
    if (tsv > 0.0 && tkey == 0.0) {
        const double g = -sqrt(tsr / tsv);
        for (int a = threadIdx.x; a < a_pad; a += BLOCK) {
            if (!pres[a]) continue;
            const double p1 = q1[a] / tot1, p2 = q2[a] / tot2;
            const double r = p1 - p[a];
            const double v = p2 - p1 - r;
            q2[a] = fmax(0.0, p[a] - 2 * g * r + g * g * v);
            pres2[a] = 1;
        }
    }
    if (threadIdx.x == 0) {
        scal[S_FLAG] = (tsv > 0.0 && tkey == 0.0) ? 1.0 : 0.0;
        if (tkey != 0.0) { scal[S_KEYERR] = 1.0; scal[S_DONE] = 1.0; }
    }
}

// diff = prob_diff(p, pn) (common:1272-1279) with pn = (flag ? q3 : q1) normalised; p <- pn; pruning
// (common:1338-1346, from iteration 10); stopping rule (common:1351)
__global__ __launch_bounds__(BLOCK) void k_em_advance(double *__restrict__ p, uint8_t *__restrict__ pres,
                                                      const double *__restrict__ q1, const uint8_t *__restrict__ pres1,
                                                      const double *__restrict__ q3, const uint8_t *__restrict__ pres3,
                                                      int a_pad, int remove_low, double *__restrict__ scal) {
    __shared__ double sh[NWAVE];
    if (scal[S_DONE] != 0.0) return;
    const bool ext = scal[S_FLAG] != 0.0;
    const double *qn = ext ? q3 : q1;
    const uint8_t *prn = ext ? pres3 : pres1;
    double s = 0.0;
    for (int a = threadIdx.x; a < a_pad; a += BLOCK) if (prn[a]) s += qn[a];
    const double tot = block_sum(s, sh);
    double d = 0.0, mx = 0.0;
    for (int a = threadIdx.x; a < a_pad; a += BLOCK) {
        const double pn = prn[a] ? qn[a] / tot : 0.0;
        if (pres[a]) d += prn[a] ? fabs(p[a] - pn) : p[a];
        if (prn[a]) mx = fmax(mx, pn);
    }
    const double td = block_sum(d, sh);
    const double tm = block_max(mx, sh);
    const int iter = (int)scal[S_ITER];
    const bool prune = remove_low && iter >= 10;
    for (int a = threadIdx.x; a < a_pad; a += BLOCK) {
        const double pn = prn[a] ? qn[a] / tot : 0.0;
        bool keep = prn[a];
        if (prune && keep) keep = pn >= tm / 10.0;
        pres[a] = keep ? 1 : 0;
        p[a] = keep ? pn : 0.0;
    }
    if (threadIdx.x == 0) {
        scal[S_DIFF] = td;
        scal[S_ITER] = (double)(iter + 1);
        if (!(td > 0.0001) || iter + 1 >= 1000) scal[S_DONE] = 1.0;
    }
}

// initial estimate (common:1299-1309): normalise the mass vector in place
__global__ __launch_bounds__(BLOCK) void k_em_init_norm(double *__restrict__ p, const uint8_t *__restrict__ pres, int a_pad) {
    __shared__ double sh[NWAVE];
    double s = 0.0;
    for (int a = threadIdx.x; a < a_pad; a += BLOCK) if (pres[a]) s += p[a];
    const double tot = block_sum(s, sh);
    for (int a = threadIdx.x; a < a_pad; a += BLOCK) p[a] = pres[a] ? p[a] / tot : 0.0;
}

// final select_alleles + normalise (common:1402-1407); out = -1 for alleles not in the dict
__global__ __launch_bounds__(BLOCK) void k_em_finish(const double *__restrict__ p, const uint8_t *__restrict__ pres,
                                                     const double *__restrict__ len, int a_pad, int prune,
                                                     double *__restrict__ out) {
    __shared__ double sh[NWAVE];
    double mx = 0.0;
    for (int a = threadIdx.x; a < a_pad; a += BLOCK) if (pres[a]) mx = fmax(mx, p[a]);
    const double tm = block_max(mx, sh);
    double s = 0.0;
    for (int a = threadIdx.x; a < a_pad; a += BLOCK) {
        const bool keep = pres[a] && (!prune || p[a] >= tm / 10.0);
        if (keep) s += len ? p[a] / len[a] : p[a];
    }
    const double tot = block_sum(s, sh);
    for (int a = threadIdx.x; a < a_pad; a += BLOCK) {
        const bool keep = pres[a] && (!prune || p[a] >= tm / 10.0);
        out[a] = keep ? (len ? p[a] / len[a] / tot : p[a] / tot) : -1.0;
    }
}

struct MatVec {
    const uint64_t *B;
    int n_rows, n_words, n_k;
};

inline int rows_per_block(int n_rows) {
    int rpb = ((n_rows + 511) / 512 + RB - 1) / RB * RB;
    return std::max(RB, std::min(MAX_RPB, rpb));
}

template <int MODE>
int launch_matvec(const MatVec &m, hipStream_t st, const double *vec, const uint8_t *vec_pres, int x_mode,
                  const int64_t *count, const double *q_in, const uint8_t *pres_in, const double *len, double *y,
                  uint8_t *pres_out, double *scal, int gate) {
    const int rpb = rows_per_block(m.n_rows);
    const int grid = (m.n_rows + rpb - 1) / rpb;
    if (m.n_k <= 8 * BLOCK)
        hipLaunchKernelGGL((k_bitmatvec<8, MODE>), dim3(grid), dim3(BLOCK), 0, st, m.B, m.n_rows, m.n_words, m.n_k, rpb, vec,
                           vec_pres, x_mode, count, q_in, pres_in, len, y, pres_out, scal, gate);
    else
        hipLaunchKernelGGL((k_bitmatvec<16, MODE>), dim3(grid), dim3(BLOCK), 0, st, m.B, m.n_rows, m.n_words, m.n_k, rpb, vec,
                           vec_pres, x_mode, count, q_in, pres_in, len, y, pres_out, scal, gate);
    return HGX_OK;
}

__global__ void k_counts_out(const double *__restrict__ sum, const double *__restrict__ first, int n, int64_t *__restrict__ out_count,
                             int32_t *__restrict__ out_first) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n) return;
    out_count[a] = (int64_t)sum[a];
    out_first[a] = first[a] < 1e300 ? (int32_t)first[a] : -1;
}

// Per-kernel timing for bench.py's roofline object.  When enabled (hgx_em_set_timing) every bit-mat-vec launch of
// hgx_em is bracketed by HIP events on its stream; totals accumulate per kernel instantiation until reset.
struct PassStats { double ms = 0; int64_t launches = 0, executed = 0, bytes = 0; };
thread_local PassStats g_stats[4];   // [0] <8,ROWS> [1] <16,ROWS> [2] <8,COLS> [3] <16,COLS>
thread_local int g_timing = 0;
struct Timed { hipEvent_t a, b; int slot; };

}   // namespace

extern "C" int hgx_em_set_timing(int on) {
    g_timing = on;
    for (auto &s : g_stats) s = PassStats();
    return HGX_OK;
}
// slot: 0 <8,ROWS>, 1 <16,ROWS>, 2 <8,COLS>, 3 <16,COLS>
extern "C" int hgx_em_get_timing(int slot, double *ms_total, int64_t *launches, int64_t *executed, int64_t *bytes_total) {
    ARGCHK(slot >= 0 && slot < 4);
    if (ms_total) *ms_total = g_stats[slot].ms;
    if (launches) *launches = g_stats[slot].launches;
    if (executed) *executed = g_stats[slot].executed;
    if (bytes_total) *bytes_total = g_stats[slot].bytes;
    return HGX_OK;
}

extern "C" int hgx_em(const hgx_classes *cc, int32_t n_alleles, int32_t remove_low, const int32_t *allele_len,
                      double *prob_host, int32_t *n_iter_host, void *stream) {
    ARGCHK(cc && prob_host && n_alleles > 0 && n_alleles <= cc->a_pad);
    hgx_classes *c = const_cast<hgx_classes *>(cc);
    hipStream_t st = (hipStream_t)stream;
    const int A = c->a_pad, C = c->n_classes;
    if (n_iter_host) *n_iter_host = 0;
    if (C == 0) {
        for (int a = 0; a < n_alleles; ++a) prob_host[a] = -1.0;
        return HGX_OK;
    }
    int rc = hgx_ensure_transposed(c, st);
    if (rc) return rc;
    DevBuf b_p, b_q1, b_q2, b_q3, b_wc, b_pr, b_pr1, b_pr2, b_pr3, b_len, b_scal, b_out;
    ALLOC(b_p, A * 8); ALLOC(b_q1, A * 8); ALLOC(b_q2, A * 8); ALLOC(b_q3, A * 8); ALLOC(b_out, A * 8);
    ALLOC(b_wc, (size_t)C * 8); ALLOC(b_pr, A); ALLOC(b_pr1, A); ALLOC(b_pr2, A); ALLOC(b_pr3, A); ALLOC(b_scal, S_N * 8);
    double *d_len = nullptr;
    if (allele_len) {
        std::vector<double> l(A, 1.0);
        for (int a = 0; a < n_alleles; ++a) l[a] = (double)allele_len[a];
        ALLOC(b_len, A * 8);
        HIPCHK(hipMemcpyAsync(b_len.p, l.data(), A * 8, hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
        d_len = b_len.as<double>();
    }
    double *p = b_p.as<double>(), *q1 = b_q1.as<double>(), *q2 = b_q2.as<double>(), *q3 = b_q3.as<double>();
    uint8_t *pr = b_pr.as<uint8_t>(), *pr1 = b_pr1.as<uint8_t>(), *pr2 = b_pr2.as<uint8_t>(), *pr3 = b_pr3.as<uint8_t>();
    double *wc = b_wc.as<double>(), *scal = b_scal.as<double>();
    HIPCHK(hipMemsetAsync(scal, 0, S_N * 8, st));
    const MatVec rows{c->d_bits, C, c->w64, A};
    const MatVec cols{c->d_bitsT, A, c->c64, C};

    std::vector<Timed> timed;
    const int slot_rows = A <= 8 * BLOCK ? 0 : 1, slot_cols = C <= 8 * BLOCK ? 2 : 3;
    auto stamp = [&](int slot, bool begin) {
        if (!g_timing) return;
        if (begin) {
            Timed t;
            (void)hipEventCreate(&t.a);
            (void)hipEventCreate(&t.b);
            t.slot = slot;
            (void)hipEventRecord(t.a, st);
            timed.push_back(t);
        } else (void)hipEventRecord(timed.back().b, st);
    };
    // one application of the EM map: (vec, pres_v) -> (q_out, pres_out)
    auto next_prob = [&](const double *vec, const uint8_t *pres_v, int x_mode, double *q_out, uint8_t *pres_out, int gate) -> int {
        stamp(slot_rows, true);
        int r = launch_matvec<MODE_ROWS>(rows, st, vec, pres_v, x_mode, c->d_count, nullptr, nullptr, nullptr, wc, nullptr, scal, gate);
        stamp(slot_rows, false);
        if (r) return r;
        stamp(slot_cols, true);
        r = launch_matvec<MODE_COLS>(cols, st, wc, nullptr, x_mode == 2 ? 2 : 0, nullptr, vec, pres_v, d_len, q_out, pres_out, scal, gate);
        stamp(slot_cols, false);
        return r;
    };
    // initial mass sum_c n_c / |S_c|, normalised (common:1299-1309)
    rc = next_prob(p, pr, 2, p, pr, 0);
    if (rc) return rc;
    hipLaunchKernelGGL(k_em_init_norm, dim3(1), dim3(BLOCK), 0, st, p, pr, A);
    double h_scal[S_N];
    const int batch = 4;
    for (;;) {
        for (int b = 0; b < batch; ++b) {
            if ((rc = next_prob(p, pr, 0, q1, pr1, 0))) return rc;        // Gene_prob_next  (p is used raw)
            if ((rc = next_prob(q1, pr1, 1, q2, pr2, 0))) return rc;      // Gene_prob_next2 (normalised on the fly)
            hipLaunchKernelGGL(k_em_squarem, dim3(1), dim3(BLOCK), 0, st, p, pr, q1, pr1, q2, pr2, A, scal);
            if ((rc = next_prob(q2, pr2, 0, q3, pr3, 1))) return rc;      // only if extrapolated
            hipLaunchKernelGGL(k_em_advance, dim3(1), dim3(BLOCK), 0, st, p, pr, q1, pr1, q3, pr3, A, remove_low ? 1 : 0, scal);
        }
        HIPCHK(hipMemcpyAsync(h_scal, scal, S_N * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (h_scal[S_DONE] != 0.0) break;
    }
    if (g_timing) {
        const int64_t rows_bytes = (int64_t)C * c->w64 * 8 + (int64_t)A * 9 + (int64_t)C * 16;
        const int64_t cols_bytes = (int64_t)A * c->c64 * 8 + (int64_t)C * 8 + (int64_t)A * 26;
        for (auto &t : timed) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, t.a, t.b);
            g_stats[t.slot].ms += ms;
            g_stats[t.slot].launches += 1;
            (void)hipEventDestroy(t.a);
            (void)hipEventDestroy(t.b);
        }
        g_stats[slot_rows].executed += (int64_t)h_scal[S_NROWS];
        g_stats[slot_rows].bytes += (int64_t)h_scal[S_NROWS] * rows_bytes;
        g_stats[slot_cols].executed += (int64_t)h_scal[S_NCOLS];
        g_stats[slot_cols].bytes += (int64_t)h_scal[S_NCOLS] * cols_bytes;
    }
    if (h_scal[S_KEYERR] != 0.0) {
        hgx_set_error("EM: allele missing from the next estimate (the reference raises KeyError here, common:1365-1369)");
        return HGX_EKEY;
    }
    hipLaunchKernelGGL(k_em_finish, dim3(1), dim3(BLOCK), 0, st, p, pr, d_len, A, remove_low ? 1 : 0, b_out.as<double>());
    HIPCHK(hipGetLastError());
    std::vector<double> out(A);
    HIPCHK(hipMemcpyAsync(out.data(), b_out.p, A * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int a = 0; a < n_alleles; ++a) prob_host[a] = out[a];
    if (n_iter_host) *n_iter_host = (int)h_scal[S_ITER];
    return HGX_OK;
}

// Gene_counts (typing_core.py:1187-1190): per allele the summed count of the classes containing it, and the first
// such class (dict insertion order for ties) -- two passes of the bit mat-vec over the transposed class matrix.
extern "C" int hgx_allele_counts(const hgx_classes *cc, int64_t *count_host, int32_t *first_host) {
    ARGCHK(cc && count_host && first_host);
    hgx_classes *c = const_cast<hgx_classes *>(cc);
    const int A = c->a_pad;
    if (c->n_classes == 0) {
        for (int a = 0; a < A; ++a) { count_host[a] = 0; first_host[a] = -1; }
        return HGX_OK;
    }
    int rc = hgx_ensure_transposed(c, nullptr);
    if (rc) return rc;
    DevBuf b_s, b_f, b_c, b_i;
    ALLOC(b_s, (size_t)A * 8); ALLOC(b_f, (size_t)A * 8); ALLOC(b_c, (size_t)A * 8); ALLOC(b_i, (size_t)A * 4);
    const MatVec cols{c->d_bitsT, A, c->c64, c->n_classes};
    launch_matvec<MODE_SUM>(cols, nullptr, nullptr, nullptr, 0, c->d_count, nullptr, nullptr, nullptr, b_s.as<double>(), nullptr, nullptr, 0);
    launch_matvec<MODE_MIN>(cols, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, b_f.as<double>(), nullptr, nullptr, 0);
    hipLaunchKernelGGL(k_counts_out, dim3(nblk(A, 256)), dim3(256), 0, nullptr, b_s.as<double>(), b_f.as<double>(), A,
                       b_c.as<int64_t>(), b_i.as<int32_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(count_host, b_c.p, (size_t)A * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(first_host, b_i.p, (size_t)A * 4, hipMemcpyDeviceToHost));
    return HGX_OK;
}
