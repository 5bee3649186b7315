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
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include <hip/hip_ext.h>

#include "hgx_common.hpp"
#include "hgx_emx.hpp"

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
    S_NPRES = 8,     // alleles still in the estimate after the last advance
    S_TAIL = 9,      // 1 = the compact tail kernel finished the EM, -1 = it could not take over
    S_N = 10
};

constexpr int BLOCK = 1024;
constexpr int NWAVE = BLOCK / 64;

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
    v = wave_max_nonneg_f64(v);
    __syncthreads();
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    double t = sh[0];
#pragma unroll
    for (int i = 1; i < NWAVE; ++i) t = fmax(t, sh[i]);
    return t;
}

enum { MODE_ROWS = 0, MODE_COLS = 1, MODE_SUM = 2, MODE_MIN = 3 };

// N block-wide sums with a single barrier pair (fixed summation order: identical in every launch)
template <int N>
__device__ __forceinline__ void block_sum_n(double (&v)[N], double (*sh)[NWAVE]) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int n = 0; n < N; ++n) v[n] = wave_sum_f64(v[n]);
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int n = 0; n < N; ++n) sh[n][wv] = v[n];
    __syncthreads();
#pragma unroll
    for (int n = 0; n < N; ++n) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < NWAVE; ++i) t += sh[n][i];
        v[n] = t;
    }
}

#ifdef HGX_LAB
#include "lab/hgx_em_bitmatvec.inc"        // round 1's VALU bit mat-vec (backend 1) and its helpers: lab build only
#endif

constexpr int EPT = 8;     // vector elements per thread kept in registers by the single-workgroup kernels (a_pad <= 8192)

// SQUAREM extrapolation (common:1361-1380).  p = pq (normalised already), p1 = q1/sum(q1), p2 = q2/sum(q2).
// Writes q2 <- max(0, p - 2 g r + g^2 v) (used raw by the third map) when sum v^2 > 0.
// One workgroup; every operand is read once into registers, two reduction rounds.
__global__ __launch_bounds__(BLOCK) void k_em_squarem(const double *__restrict__ p, const uint8_t *__restrict__ pres,
                                                      const double *__restrict__ q1, const uint8_t *__restrict__ pres1,
                                                      double *__restrict__ q2, uint8_t *__restrict__ pres2, int a_pad,
                                                      double *__restrict__ scal) {
    __shared__ double sh[3][NWAVE];
    const double done = scal[S_DONE];
    double vp[EPT], v1[EPT], v2[EPT];
    uint8_t f0[EPT], f1[EPT], f2[EPT];
    const bool small = a_pad <= EPT * BLOCK;
    double red[2] = {0.0, 0.0};
    if (small) {
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int a = threadIdx.x + BLOCK * k;
            const bool in = a < a_pad;
            f0[k] = in ? pres[a] : 0; f1[k] = in ? pres1[a] : 0; f2[k] = in ? pres2[a] : 0;
            vp[k] = in ? p[a] : 0.0; v1[k] = in ? q1[a] : 0.0; v2[k] = in ? q2[a] : 0.0;
        }
    }
    if (done != 0.0) return;
    if (small) {
#pragma unroll
        for (int k = 0; k < EPT; ++k) { if (f1[k]) red[0] += v1[k]; if (f2[k]) red[1] += v2[k]; }
    } else {
        for (int a = threadIdx.x; a < a_pad; a += BLOCK) { if (pres1[a]) red[0] += q1[a]; if (pres2[a]) red[1] += q2[a]; }
    }
    block_sum_n<2>(red, sh);
    const double tot1 = red[0], tot2 = red[1];
    double acc[3] = {0.0, 0.0, 0.0};   // sum r^2, sum v^2, key error
    auto term = [&](bool a0, bool a1, bool a2, double x0, double x1, double x2) {
        if (!a0) return;
        if (!a1 || !a2) { acc[2] = 1.0; return; }
        const double p1 = x1 / tot1, p2 = x2 / tot2;
        const double r = p1 - x0;
        const double v = p2 - p1 - r;
        acc[0] += r * r;
        acc[1] += v * v;
    };
    if (small) {
#pragma unroll
        for (int k = 0; k < EPT; ++k) term(f0[k], f1[k], f2[k], vp[k], v1[k], v2[k]);
    } else {
        for (int a = threadIdx.x; a < a_pad; a += BLOCK) term(pres[a], pres1[a], pres2[a], p[a], q1[a], q2[a]);
    }
    block_sum_n<3>(acc, sh);
    const double tsr = acc[0], tsv = acc[1], tkey = acc[2];
    if (tsv > 0.0 && tkey == 0.0) {
        const double g = -sqrt(tsr / tsv);
        auto upd = [&](int a, double x0, double x1, double x2) {
            const double p1 = x1 / tot1, p2 = x2 / tot2;
            const double r = p1 - x0;
            const double v = p2 - p1 - r;
            q2[a] = fmax(0.0, x0 - 2 * g * r + g * g * v);
            pres2[a] = 1;
        };
        if (small) {
#pragma unroll
            for (int k = 0; k < EPT; ++k) if (f0[k]) upd(threadIdx.x + BLOCK * k, vp[k], v1[k], v2[k]);
        } else {
            for (int a = threadIdx.x; a < a_pad; a += BLOCK) if (pres[a]) upd(a, p[a], q1[a], q2[a]);
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
    __shared__ double sh[2][NWAVE];
    __shared__ double shm[NWAVE];
    const double done = scal[S_DONE], flag = scal[S_FLAG], it_d = scal[S_ITER];
    // both candidate sources are fetched before the flag is known (one memory round trip instead of two)
    double vp[EPT], va[EPT], vb[EPT];
    uint8_t f0[EPT], fa[EPT], fb[EPT];
    const bool small = a_pad <= EPT * BLOCK;
    if (small) {
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int a = threadIdx.x + BLOCK * k;
            const bool in = a < a_pad;
            f0[k] = in ? pres[a] : 0; fa[k] = in ? pres1[a] : 0; fb[k] = in ? pres3[a] : 0;
            vp[k] = in ? p[a] : 0.0; va[k] = in ? q1[a] : 0.0; vb[k] = in ? q3[a] : 0.0;
        }
    }
    if (done != 0.0) return;
    const bool ext = flag != 0.0;
    const double *qn = ext ? q3 : q1;
    const uint8_t *prn = ext ? pres3 : pres1;
    double s[1] = {0.0};
    if (small) {
#pragma unroll
        for (int k = 0; k < EPT; ++k) { const bool pr_ = ext ? fb[k] : fa[k]; if (pr_) s[0] += ext ? vb[k] : va[k]; }
    } else {
        for (int a = threadIdx.x; a < a_pad; a += BLOCK) if (prn[a]) s[0] += qn[a];
    }
    block_sum_n<1>(s, sh);
    const double tot = s[0];
    double d[1] = {0.0}, mx = 0.0;
    auto term = [&](bool a0, bool an, double x0, double xn) {
        const double pn = an ? xn / tot : 0.0;
        if (a0) d[0] += an ? fabs(x0 - pn) : x0;
        if (an) mx = fmax(mx, pn);
    };
    if (small) {
#pragma unroll
        for (int k = 0; k < EPT; ++k) term(f0[k], ext ? fb[k] : fa[k], vp[k], ext ? vb[k] : va[k]);
    } else {
        for (int a = threadIdx.x; a < a_pad; a += BLOCK) term(pres[a], prn[a], p[a], qn[a]);
    }
    const double tm = block_max(mx, shm);
    block_sum_n<1>(d, sh);
    const double td = d[0];
    const int iter = (int)it_d;
    const bool prune = remove_low && iter >= 10;
    double kept[1] = {0.0};
    auto store = [&](int a, bool an, double xn) {
        const double pn = an ? xn / tot : 0.0;
        bool keep = an;
        if (prune && keep) keep = pn >= tm / 10.0;
        pres[a] = keep ? 1 : 0;
        p[a] = keep ? pn : 0.0;
        if (keep) kept[0] += 1.0;
    };
    if (small) {
#pragma unroll
        for (int k = 0; k < EPT; ++k) { const int a = threadIdx.x + BLOCK * k; if (a < a_pad) store(a, ext ? fb[k] : fa[k], ext ? vb[k] : va[k]); }
    } else {
        for (int a = threadIdx.x; a < a_pad; a += BLOCK) store(a, prn[a], qn[a]);
    }
    block_sum_n<1>(kept, sh);
    if (threadIdx.x == 0) {
        scal[S_NPRES] = kept[0];
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



#ifdef HGX_LAB
#include "lab/hgx_em_small.inc"          // one-workgroup EM for <= 64 classes over many alleles (superseded by k_emx)
#endif

// ------------------------------------------------------------------------------------------------------------
// Whole EM in ONE WAVEFRONT when at most 64 classes involve at most 64 distinct alleles (the exon->gene hand-off EM,
// STR loci): the alleles that occur at all are renumbered 0..A'-1, lane j is allele j AND class j at the same time, the
// class matrix becomes 64-bit row masks (per class, over alleles) and column masks (per allele, over classes) held
// in registers, and both halves of the map are loops of <= 64 "broadcast one scalar, add it under an EXEC mask" steps.
// No LDS traffic, no barriers, one launch.  scal[S_FALLBACK] = 1 if more than 64 alleles occur (caller falls back).
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t lane_u64(uint64_t v, int l) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)v, l);
}
// acc += sx on the lanes selected by `mask` (both wave-uniform, in scalar registers): the mask register pair is the lane
// selector of a v_cndmask (sx or +0.0), followed by an unconditional add -- adding +0.0 leaves the non-negative partial sums
// bit-identical.  An EXEC-masked add (s_and_saveexec / v_add_f64 / restore, as k_bitmatvec does with streamed matrix words)
// serialises on the EXEC write hazards when it is the whole loop body: 125 cycles per step on the single wave of the small-EM
// kernels, against ~12 for this form, where the next steps' v_readlanes overlap the dependent adds.
__device__ __forceinline__ void masked_add_s(double &acc, uint64_t sx_bits, uint64_t mask) {
    acc += __builtin_amdgcn_inverse_ballot_w64(mask) ? __longlong_as_double((long long)sx_bits) : 0.0;
}
__device__ __forceinline__ double wave_max_f64(double v) { return wave_max_nonneg_f64(v); }      // abundances: >= 0

struct WaveEM {
    uint64_t R, K;        // as class `lane`: members over alleles; as allele `lane`: classes containing it
    double n;             // class count (0 for lanes >= C)
    double len;           // allele length (1 if unused)
    int C, A1;
};

// out = T(x / scale) restricted to `in_pres`; returns the presence of every allele-lane in out_pres
__device__ __forceinline__ double wave_map(const WaveEM &E, double x, bool in_pres, double scale, bool init, bool use_len,
                                           bool &out_pres) {
    const double xs = init ? 1.0 : (in_pres ? x / scale : 0.0);
    double s = 0.0;
    for (int j = 0; j < E.A1; ++j) {                        // rows half: lane = class, broadcast allele j
        const uint64_t xj = lane_u64((uint64_t)__double_as_longlong(xs), j);
        masked_add_s(s, xj, lane_u64(E.K, j));
    }
    const double w = s > 0.0 ? E.n / s : 0.0;
    double t = 0.0;
    for (int c = 0; c < E.C; ++c) {                         // cols half: lane = allele, broadcast class c
        const uint64_t wc = lane_u64((uint64_t)__double_as_longlong(w), c);
        masked_add_s(t, wc, lane_u64(E.R, c));
    }
    const bool in = init || in_pres;
    double v = 0.0;
    out_pres = in && t > 0.0;
    if (out_pres) {
        v = init ? t : xs * t;
        if (use_len) v = v / E.len;
    }
    return v;
}

// The SQUAREM loop + final selection (common:1311-1410) on one wavefront, starting from estimate (p, pr) at iteration
// `iter`; lane j < E.A1 writes its allele's result to out[g] (the caller pre-fills out with -1).
__device__ __forceinline__ void wave_em_run(const WaveEM &E, double p, bool pr, int iter, int remove_low, bool use_len, int g,
                                            double *__restrict__ out, double *__restrict__ scal) {
    const int lane = threadIdx.x & 63;
    bool pr1, pr2, pr3;
    double diff = 1.0;
    bool keyerr = false;
    while (diff > 0.0001 && iter < 1000) {
        const double q1 = wave_map(E, p, pr, 1.0, false, use_len, pr1);
        const double tot1 = wave_sum_f64(pr1 ? q1 : 0.0);
        double q2 = wave_map(E, q1, pr1, tot1, false, use_len, pr2);
        const double tot2 = wave_sum_f64(pr2 ? q2 : 0.0);
        double r = 0.0, v = 0.0;
        const bool bad = pr && (!pr1 || !pr2);
        if (pr && !bad) {
            const double p1 = q1 / tot1, p2 = q2 / tot2;
            r = p1 - p;
            v = p2 - p1 - r;
        }
        if (__any(bad)) { keyerr = true; break; }
        const double sr = wave_sum_f64(r * r), sv = wave_sum_f64(v * v);
        double pn;
        bool prn;
        if (sv > 0.0) {
            const double gm = -sqrt(sr / sv);
            if (pr) { q2 = fmax(0.0, p - 2 * gm * r + gm * gm * v); pr2 = true; }
            const double q3 = wave_map(E, q2, pr2, 1.0, false, use_len, pr3);
            const double tot3 = wave_sum_f64(pr3 ? q3 : 0.0);
            prn = pr3;
            pn = prn ? q3 / tot3 : 0.0;
        } else {
            prn = pr1;
            pn = prn ? q1 / tot1 : 0.0;
        }
        diff = wave_sum_f64(pr ? (prn ? fabs(p - pn) : p) : 0.0);
        const double mx = wave_max_f64(prn ? pn : 0.0);
        bool keep = prn;
        if (remove_low && iter >= 10 && keep) keep = pn >= mx / 10.0;
        pr = keep;
        p = keep ? pn : 0.0;
        iter += 1;
    }
    const double mx = wave_max_f64(pr ? p : 0.0);
    const bool keep = pr && (!remove_low || p >= mx / 10.0);
    const double tl = wave_sum_f64(keep ? (use_len ? p / E.len : p) : 0.0);
    if (lane < E.A1 && keep) out[g] = use_len ? p / E.len / tl : p / tl;
    if (lane == 0) { scal[S_ITER] = (double)iter; scal[S_KEYERR] = keyerr ? 1.0 : 0.0; scal[S_DONE] = 1.0; }
}

// ------------------------------------------------------------------------------------------------------------
// The same EM in the REFERENCE'S OWN ORDER of floating-point operations (single wavefront; lanes = alleles in class-key
// order = name order, classes in dict order): every sum is sequential in the order typing_common.py walks its dicts, every
// term is formed as the reference forms it (count * prob / alleles_prob), nothing is contracted -- so the result is
// bit-identical to the reference's, and the decisions it takes on rounding noise (the pruning test p >= max / 10 on an exact
// 10 : 1 ratio of small rational abundances) come out the same.  Used for the hand-off EM when the caller supplied the
// alleles' name order; follows oracle/hgx_oracle.c orc_single_abundance line by line.
// A dict is, per allele-lane: value, membership, position in the insertion order (positions may have gaps after pruning).
// ------------------------------------------------------------------------------------------------------------
struct RefDict { double v; bool in; int pos; int npos; };
struct RefOrder { uint64_t in_mask, cls_mask; int pos, npos; bool set; };      // the insertion order last computed (it rarely changes)
// Plain operators under contract(off): the compiler must not fuse a - b * c into one FMA (one rounding instead of the
// reference's two) -- seen as 2e-13 relative differences on 1e-9 abundances after the SQUAREM step.  (HIP's __dadd_rn /
// __dmul_rn are themselves plain operators defined under the default contraction mode, so they do not help.)
#pragma clang fp contract(off)

__device__ __forceinline__ double lane_f64(double v, int l) {
    return __longlong_as_double((long long)lane_u64((uint64_t)__double_as_longlong(v), l));
}
// sum of f over the dict's members in insertion order
template <class F> __device__ __forceinline__ double ref_seq_sum(const RefDict &d, F f) {
    const double mine = f();
    double t = 0.0;
    for (int r = 0; r < d.npos; ++r) {
        const uint64_t hit = __ballot(d.in && d.pos == r);
        if (hit) t = ((t) + (lane_f64(mine, __builtin_ctzll(hit))));
    }
    return t;
}
// insertion positions when the dict is filled class by class (dict order), alleles in key order: (first class, lane)
__device__ __forceinline__ void ref_positions(RefDict &d, uint64_t classes_of_me, int C, int lane) {
    const int fc = classes_of_me ? __builtin_ctzll(classes_of_me) : 64;
    int base = 0;
    d.pos = 0;
    for (int c = 0; c < C; ++c) {
        const uint64_t m = __ballot(d.in && fc == c);
        if (d.in && fc == c) d.pos = base + __popcll(m & ((1ull << lane) - 1ull));
        base += __popcll(m);
    }
    d.npos = base;
}
__device__ __forceinline__ void ref_normalize(RefDict &d, bool use_len, double len) {       // common:1285-1297
    const double total = ref_seq_sum(d, [&]() { return use_len ? ((d.v) / (len)) : d.v; });
    if (d.in) d.v = use_len ? ((((d.v) / (len))) / (total)) : ((d.v) / (total));
}
// Gene_prob_next (common:1311-1336): E.R = members of class `lane`, E.K = classes of allele `lane`, E.n = class counts
__device__ __forceinline__ RefDict ref_next(const WaveEM &E, const RefDict &prob, bool use_len, int lane, RefOrder &ord) {
    double s = 0.0;                                            // lane = class: alleles_prob, alleles in key order
    const uint64_t inmask = __ballot(prob.in);
    for (int j = 0; j < E.A1; ++j) {
        const double xj = lane_f64(prob.v, j);
        if (((inmask >> j) & 1ull) && ((E.R >> j) & 1ull)) s = ((s) + (xj));
    }
    const uint64_t valid = __ballot(lane < E.C && s > 0.0);   // classes with alleles_prob <= 0 are skipped
    RefDict next;
    next.v = 0.0;
    // lane = allele: += count * prob / alleles_prob, classes in dict order.  The quotients of four classes are formed side by
    // side (a double-precision division is a long dependent chain) and then added in order.
    for (int c0 = 0; c0 < E.C; c0 += 4) {
        double term[4];
        bool on[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c0 + k;
            const double sc = lane_f64(s, c & 63), nc = lane_f64(E.n, c & 63);
            on[k] = c < E.C && ((valid >> c) & 1ull) && prob.in && ((E.K >> c) & 1ull);
            term[k] = (nc * prob.v) / (((valid >> (c & 63)) & 1ull) ? sc : 1.0);      // formed on every lane (no branch between the four
                                                                                   // division chains); used where on[k]
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (on[k]) next.v = ((next.v) + (term[k]));
    }
    next.in = prob.in && (E.K & valid) != 0ull;
    if (!next.in) next.v = 0.0;
    // the order depends only on who is in the dict and which classes were walked: re-derive it only when either changed
    const uint64_t in_mask = __ballot(next.in);
    if (ord.set && ord.in_mask == in_mask && ord.cls_mask == valid) { next.pos = ord.pos; next.npos = ord.npos; }
    else {
        ref_positions(next, E.K & valid, E.C, lane);
        ord.in_mask = in_mask; ord.cls_mask = valid; ord.pos = next.pos; ord.npos = next.npos; ord.set = true;
    }
    ref_normalize(next, use_len, E.len);
    return next;
}
__device__ __forceinline__ void ref_select(RefDict &d) {      // common:1338-1346
    const double mx = wave_max_f64(d.in ? d.v : 0.0);
    if (d.in && !(d.v >= ((mx) / (10.0)))) { d.in = false; d.v = 0.0; }
}
__device__ __forceinline__ void ref_em_run(const WaveEM &E, int remove_low, bool use_len, int g, double *__restrict__ out,
                                           double *__restrict__ scal) {
    const int lane = threadIdx.x & 63;
    // initial estimate (common:1300-1309): prob[a] += count / |class| over the classes in dict order
    const int nal = __popcll(E.R);
    RefDict prob;
    prob.v = 0.0;
    for (int c = 0; c < E.C; ++c) {
        const double nc = lane_f64(E.n, c);
        const int nalc = __builtin_amdgcn_readlane(nal, c);
        if ((E.K >> c) & 1ull) prob.v = ((prob.v) + (((nc) / ((double)nalc))));
    }
    prob.in = lane < E.A1 && E.K != 0ull;
    ref_positions(prob, E.K, E.C, lane);
    ref_normalize(prob, use_len, E.len);
    double diff = 1.0;
    int iter = 0;
    bool keyerr = false;
    RefOrder ord;
    ord.set = false;
    while (diff > 0.0001 && iter < 1000) {                     // common:1351
        RefDict next = ref_next(E, prob, use_len, lane, ord);
        RefDict next2 = ref_next(E, next, use_len, lane, ord);
        if (__any(prob.in && (!next.in || !next2.in))) { keyerr = true; break; }      // the reference's KeyError (Q6)
        const double p_r = ((next.v) - (prob.v));
        const double p_v = ((((next2.v) - (next.v))) - (p_r));
        // the two sums advance together in the reference's loop; they are independent accumulators
        const double ssr = ref_seq_sum(prob, [&]() { return ((p_r) * (p_r)); });
        const double ssv = ref_seq_sum(prob, [&]() { return ((p_v) * (p_v)); });
        if (ssv > 0.0) {                                       // common:1370-1383
            const double gamma = -sqrt(((ssr) / (ssv)));
            if (prob.in) {
                const double x = ((((prob.v) - (((((2.0) * (gamma))) * (p_r))))) + (((((gamma) * (gamma))) * (p_v))));
                next2.v = 0.0 > x ? 0.0 : x;
            }
            next = ref_next(E, next2, use_len, lane, ord);
        }
        diff = ref_seq_sum(prob, [&]() { return next.in ? fabs(((prob.v) - (next.v))) : prob.v; });     // prob_diff, common:1272-1279
        prob = next;
        if (iter >= 10 && remove_low) ref_select(prob);
        iter += 1;
    }
    if (!keyerr) {
        if (remove_low) ref_select(prob);                      // common:1402-1407
        ref_normalize(prob, use_len, E.len);
        if (lane < E.A1 && prob.in) out[g] = prob.v;
    }
    if (lane == 0) { scal[S_ITER] = (double)iter; scal[S_KEYERR] = keyerr ? 1.0 : 0.0; scal[S_DONE] = 1.0; }
}

#pragma clang fp contract(fast)
constexpr int S_FALLBACK = 6;       // (re-uses the S_NROWS word: the wave kernel launches no mat-vec)

__global__ __launch_bounds__(64) void k_em_wave(const uint64_t *__restrict__ B, int C, int n_words, int a_pad,
                                                const int64_t *__restrict__ count, const double *__restrict__ len,
                                                int remove_low, double *__restrict__ out, double *__restrict__ scal,
                                                int32_t *__restrict__ first_out, const int32_t *__restrict__ rank) {
    // rank != NULL (the alleles' name order): lanes take the alleles in that order and the EM follows the reference's own
    // order of operations (ref_em_run: bit-identical abundances)
    __shared__ int gidx[64], gsorted[64];
    const int lane = threadIdx.x;
    // ---- which alleles occur at all ------------------------------------------------------------------------
    uint64_t u0 = 0, u1 = 0;
    const bool h0 = lane < n_words, h1 = lane + 64 < n_words;
    for (int c = 0; c < C; ++c) {
        if (h0) u0 |= B[(size_t)c * n_words + lane];
        if (h1) u1 |= B[(size_t)c * n_words + lane + 64];
    }
    const int c0 = __popcll(u0), c1 = __popcll(u1);
    int inc0 = c0, inc1 = c1;                               // inclusive scans over lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int a = __shfl_up(inc0, d, 64), b = __shfl_up(inc1, d, 64);
        if (lane >= d) { inc0 += a; inc1 += b; }
    }
    const int tot0 = __shfl(inc0, 63, 64), tot1 = __shfl(inc1, 63, 64);
    const int A1 = tot0 + tot1;
    if (A1 > 64 || n_words > 128) {
        if (lane == 0) { scal[S_FALLBACK] = 1.0; scal[S_DONE] = 1.0; }
        return;
    }
    {
        int id = inc0 - c0;
        for (uint64_t m = u0; m; m &= m - 1) gidx[id++] = 64 * lane + __builtin_ctzll(m);
        id = tot0 + inc1 - c1;
        for (uint64_t m = u1; m; m &= m - 1) gidx[id++] = 64 * (lane + 64) + __builtin_ctzll(m);
    }
    __syncthreads();
    int g = lane < A1 ? gidx[lane] : 0;
    if (rank) {                                             // key order = name order
        const int r = lane < A1 ? rank[g] : 0x7fffffff;
        int pos = 0;
        for (int k = 0; k < A1; ++k) {
            const int rk = __builtin_amdgcn_readlane(r, k);
            pos += (rk < r) || (rk == r && k < lane);
        }
        if (lane < A1) gsorted[pos] = g;
        __syncthreads();
        g = lane < A1 ? gsorted[lane] : 0;
    }
    // ---- row and column masks ----------------------------------------------------------------------------------
    WaveEM E;
    E.R = 0; E.K = 0; E.C = C; E.A1 = A1;
    for (int c = 0; c < C; ++c) {
        const bool bit = lane < A1 && ((B[(size_t)c * n_words + (g >> 6)] >> (g & 63)) & 1ull);
        const uint64_t row = __ballot(bit);
        if (lane == c) E.R = row;
        if (bit) E.K |= 1ull << c;
    }
    E.n = lane < C ? (double)count[lane] : 0.0;
    const bool use_len = len != nullptr;
    E.len = (use_len && lane < A1) ? len[g] : 1.0;
    // ---- EM (common:1299-1410) ----------------------------------------------------------------------------------
    for (int a = lane; a < a_pad; a += 64) out[a] = -1.0;
    if (first_out && lane < A1) first_out[g] = E.K ? __builtin_ctzll(E.K) : -1;     // first class (dict order) containing the allele
    __syncthreads();
    if (rank) {
        ref_em_run(E, remove_low, use_len, g, out, scal);
        return;
    }
    bool pr;
    double p = wave_map(E, 0.0, false, 1.0, true, use_len, pr);
    const double tot = wave_sum_f64(pr ? p : 0.0);
    p = pr ? p / tot : 0.0;
    wave_em_run(E, p, pr, 0, remove_low, use_len, g, out, scal);
}

#ifdef HGX_LAB
#include "lab/hgx_em_ref.inc"            // round 2's mid-size reference-order EM (superseded by k_emx)
#endif

// ------------------------------------------------------------------------------------------------------------
// Compact tail of a big EM.  Once pruning (common:1338-1346) has left <= 64 alleles in the estimate -- alleles never
// come back -- every class collapses to a 64-bit mask over the survivors, classes with equal masks merge (their counts
// add exactly) and, if <= 64 distinct masks remain, the rest of the EM runs on ONE wavefront (wave_em_run) in this
// launch instead of eight kernel launches over the whole matrix per iteration.
// One workgroup: survivors list -> per-class masks merged in an LDS hash table -> wave 0 sorts the masks (the summation
// order must not depend on the insertion race) and iterates.  scal[S_TAIL] = -1 if the masks do not fit.
// ------------------------------------------------------------------------------------------------------------
constexpr int TAIL_SLOTS = 2048;

__global__ __launch_bounds__(BLOCK) void k_em_tail(const uint64_t *__restrict__ BT, int C, int c64, int a_pad,
                                                   const int64_t *__restrict__ count, const double *__restrict__ p,
                                                   const uint8_t *__restrict__ pres, const double *__restrict__ len,
                                                   int remove_low, double *__restrict__ out, double *__restrict__ scal) {
    // launched speculatively right behind a batch of iterations: nothing to do if that batch already converged
    if (scal[S_DONE] != 0.0) {
        if (threadIdx.x == 0) scal[S_TAIL] = -2.0;
        return;
    }
    __shared__ int gidx[64], gsort[64];
    __shared__ int n_g, n_keys, n_list;
    __shared__ unsigned long long keys[TAIL_SLOTS], cnts[TAIL_SLOTS];
    __shared__ unsigned long long lk[64], lc[64];
    const int tid = threadIdx.x;
    if (tid == 0) { n_g = 0; n_keys = 0; n_list = 0; }
    for (int i = tid; i < TAIL_SLOTS; i += BLOCK) { keys[i] = 0; cnts[i] = 0; }
    __syncthreads();
    for (int a = tid; a < a_pad; a += BLOCK) {
        if (pres[a]) {
            const int k = atomicAdd(&n_g, 1);
            if (k < 64) gidx[k] = a;
        } else out[a] = -1.0;
    }
    __syncthreads();
    const int A1 = n_g;
    if (A1 > 64 || A1 == 0) {
        if (tid == 0) scal[S_TAIL] = -1.0;
        return;
    }
    if (tid < A1) {                                          // ascending allele order
        const int mine = gidx[tid];
        int rank = 0;
        for (int j = 0; j < A1; ++j) rank += gidx[j] < mine;
        gsort[rank] = mine;
    }
    __syncthreads();
    // Class masks from the TRANSPOSED matrix: lane j of a wave loads survivor j's word of 64 classes (A1 loads per 64 classes
    // instead of A1 per class); the words are then broadcast one survivor at a time (v_readlane) and lane b picks bit b of
    // each: mask of class 64w + b, A1 x 5 instructions per 64 classes.  One workgroup has only 16 waves to hide memory
    // latency with, so every wave first issues the loads of TAIL_U class words (matrix words and counts) and then works
    // through them.  (Measured at C = 16 k, 2 survivors: per-class bit gathers from the row-major matrix 64 us of a 77 us
    // launch; 64 ballots per class word 28 us; merging equal masks inside the wave before the LDS table 24 us more.)
    {
        constexpr int TAIL_U = 8;
        const int lane = tid & 63, n_cw = (C + 63) / 64;
        const size_t my_row = lane < A1 ? (size_t)gsort[lane] * c64 : 0;
        bool full = false;
        for (int w0 = tid >> 6; w0 < n_cw && !full; w0 += (BLOCK / 64) * TAIL_U) {
            unsigned long long xs[TAIL_U], cs[TAIL_U];
#pragma unroll
            for (int k = 0; k < TAIL_U; ++k) {
                const int w = w0 + k * (BLOCK / 64), c = w * 64 + lane;
                xs[k] = (lane < A1 && w < n_cw) ? BT[my_row + w] : 0ull;
                cs[k] = (w < n_cw && c < C) ? (unsigned long long)count[c] : 0ull;
            }
#pragma unroll
            for (int k = 0; k < TAIL_U; ++k) {
                if (w0 + k * (BLOCK / 64) >= n_cw) break;
                if (*(volatile int *)&n_keys > 64) { full = true; break; }
                unsigned long long m = 0;
                for (int j = 0; j < A1; ++j) m |= ((lane_u64(xs[k], j) >> lane) & 1ull) << j;
                if (m == 0) continue;
                unsigned h = (unsigned)(mix64(m) & (TAIL_SLOTS - 1));
                for (;;) {
                    const unsigned long long old = atomicCAS(&keys[h], 0ull, m);
                    if (old == 0ull) atomicAdd(&n_keys, 1);
                    if (old == 0ull || old == m) { atomicAdd(&cnts[h], cs[k]); break; }
                    h = (h + 1) & (TAIL_SLOTS - 1);
                }
            }
        }
    }
    __syncthreads();
    const int C1 = n_keys;
    if (C1 > 64) {
        if (tid == 0) scal[S_TAIL] = -1.0;
        return;
    }
    if (tid >= 64) return;
    const int lane = tid;
    for (int i = lane; i < TAIL_SLOTS; i += 64) {
        if (keys[i]) {
            const int k = atomicAdd(&n_list, 1);
            lk[k] = keys[i];
            lc[k] = cnts[i];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    unsigned long long myk = lane < C1 ? lk[lane] : ~0ull, myc = lane < C1 ? lc[lane] : 0ull;
    int rank = 0;
    for (int j = 0; j < C1; ++j) rank += lk[j] < myk;
    // move (key, count) to lane `rank`
    WaveEM E;
    E.R = 0; E.K = 0; E.n = 0.0; E.C = C1; E.A1 = A1;
    for (int j = 0; j < C1; ++j) {
        const int rj = __builtin_amdgcn_readlane(rank, j);
        const uint64_t kj = lane_u64(myk, j), cj = lane_u64(myc, j);
        if (lane == rj) { E.R = kj; E.n = (double)(long long)cj; }
    }
    E.K = wave_transpose64(E.R);              // allele j's classes = column j of the class masks (zero beyond A1 / C1)
    const int g = lane < A1 ? gsort[lane] : 0;
    const bool use_len = len != nullptr;
    E.len = (use_len && lane < A1) ? len[g] : 1.0;
    const double x = lane < A1 ? p[g] : 0.0;
    if (lane < A1) out[g] = -1.0;
    if (lane == 0) scal[S_TAIL] = 1.0;
    wave_em_run(E, x, lane < A1, (int)scal[S_ITER], remove_low, use_len, g, out, scal);
}

// ------------------------------------------------------------------------------------------------------------
// Hand-off EM in one launch (core:1752-1782: Gene_cmpt2 = every gene-level class filtered to exon_alleles, empty ones
// dropped, counts merged; then single_abundance with lengths).  When the filter keeps <= 64 alleles the whole thing --
// filter, merge, EM -- fits the single-wavefront machinery: every class collapses to a 64-bit mask over the kept alleles,
// equal masks merge in an LDS hash table (count added, FIRST class index kept), wave 0 orders the merged classes by that
// first index (= the dict order a dedup of the filtered rows produces, so the sums come out bit-identical to the
// dedup + k_em_wave path) and runs the EM from the initial estimate.  scal[S_FALLBACK] = 1 if > 64 distinct masks remain.
// ------------------------------------------------------------------------------------------------------------
constexpr int S_NCLS = 7;           // (re-uses the S_NCOLS word) number of merged classes

// Several workgroups scan the class rows (one LDS merge table each); every workgroup publishes its <= 64 merged entries with
// device-scope stores and takes a ticket; the last one merges the published entries and runs the EM (cf. k_lutmatvec's tail).
struct MaskedEntry { unsigned long long key, cnt; unsigned int first, pad; };

__global__ __launch_bounds__(BLOCK) void k_em_masked(const uint64_t *__restrict__ B, int C, int n_words,
                                                     const int64_t *__restrict__ count, const int32_t *__restrict__ al, int A1,
                                                     const double *__restrict__ lenc, int remove_low,
                                                     double *__restrict__ out, int32_t *__restrict__ first_out,
                                                     double *__restrict__ scal, MaskedEntry *__restrict__ pub,
                                                     unsigned *__restrict__ ticket, int exact) {
    __shared__ int n_keys, n_list, is_last;
    __shared__ unsigned long long keys[TAIL_SLOTS], cnts[TAIL_SLOTS];
    __shared__ unsigned int firsts[TAIL_SLOTS];
    __shared__ unsigned long long lk[64], lc[64];
    __shared__ unsigned int lf[64];
    __shared__ int sal[64];
    const int tid = threadIdx.x, nb = gridDim.x, bid = blockIdx.x;
    auto reset_table = [&]() {
        if (tid == 0) { n_keys = 0; n_list = 0; }
        for (int i = tid; i < TAIL_SLOTS; i += BLOCK) { keys[i] = 0; cnts[i] = 0; firsts[i] = 0xFFFFFFFFu; }
    };
    auto insert = [&](unsigned long long m, unsigned long long cn, unsigned int fi) {
        unsigned h = (unsigned)(mix64(m) & (TAIL_SLOTS - 1));
        for (;;) {
            const unsigned long long old = atomicCAS(&keys[h], 0ull, m);
            if (old == 0ull) atomicAdd(&n_keys, 1);
            if (old == 0ull || old == m) {
                atomicAdd(&cnts[h], cn);
                atomicMin(&firsts[h], fi);
                return;
            }
            h = (h + 1) & (TAIL_SLOTS - 1);
        }
    };
    reset_table();
    if (tid < 64) sal[tid] = tid < A1 ? al[tid] : 0;
    __syncthreads();
    const int per = (C + nb - 1) / nb, c_lo = bid * per, c_hi = min(C, c_lo + per);
    for (int c = c_lo + tid; c < c_hi; c += BLOCK) {
        const uint64_t *row = B + (size_t)c * n_words;
        unsigned long long m = 0;
        for (int j = 0; j < A1; ++j) {
            const int g = sal[j];
            m |= ((row[g >> 6] >> (g & 63)) & 1ull) << j;
        }
        if (m == 0) continue;
        if (*(volatile int *)&n_keys > 64) break;
        insert(m, (unsigned long long)count[c], (unsigned int)c);
    }
    __syncthreads();
    // publish this workgroup's entries (device-scope stores), then the ticket
    const int mine = n_keys;
    if (tid < 64) {
        for (int i = tid; i < TAIL_SLOTS; i += 64) {
            if (keys[i]) {
                const int k = atomicAdd(&n_list, 1);
                if (k < 64) { lk[k] = keys[i]; lc[k] = cnts[i]; lf[k] = firsts[i]; }
            }
        }
    }
    __syncthreads();
    if (tid < 64) {
        MaskedEntry *e = pub + (size_t)bid * 64 + tid;
        const bool have = mine <= 64 && tid < mine;
        __hip_atomic_store(&e->key, have ? lk[tid] : (mine > 64 ? ~0ull : 0ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&e->cnt, have ? lc[tid] : 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&e->first, have ? lf[tid] : 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = old == (unsigned)(nb - 1);
    }
    __syncthreads();
    if (!is_last) return;
    // ---- last workgroup: merge what everybody published ----
    reset_table();
    __syncthreads();
    bool overflow = false;
    for (int i = tid; i < nb * 64; i += BLOCK) {
        const MaskedEntry *e = pub + i;
        const unsigned long long k = __hip_atomic_load(&e->key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (k == ~0ull) { overflow = true; continue; }
        if (k == 0ull) continue;
        const unsigned long long cn = __hip_atomic_load(&e->cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int fi = __hip_atomic_load(&e->first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        insert(k, cn, fi);
    }
    const int any_over = __syncthreads_or(overflow);
    const int C1 = n_keys;
    if (any_over || C1 > 64) {
        if (tid == 0) { scal[S_FALLBACK] = 1.0; scal[S_DONE] = 1.0; }
        return;
    }
    if (tid >= 64) return;
    const int lane = tid;
    for (int i = lane; i < TAIL_SLOTS; i += 64) {
        if (keys[i]) {
            const int k = atomicAdd(&n_list, 1);
            lk[k] = keys[i];
            lc[k] = cnts[i];
            lf[k] = firsts[i];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const unsigned long long myk = lane < C1 ? lk[lane] : 0ull, myc = lane < C1 ? lc[lane] : 0ull;
    const unsigned int myf = lane < C1 ? lf[lane] : 0xFFFFFFFFu;
    int rank = 0;                                            // first-seen order: rank by the first class index (all distinct)
    for (int j = 0; j < C1; ++j) rank += lf[j] < myf;
    WaveEM E;
    E.R = 0; E.K = 0; E.n = 0.0; E.C = C1; E.A1 = A1;
    for (int j = 0; j < C1; ++j) {
        const int rj = __builtin_amdgcn_readlane(rank, j);
        const uint64_t kj = lane_u64(myk, j), cj = lane_u64(myc, j);
        if (lane == rj) { E.R = kj; E.n = (double)(long long)cj; }
    }
    E.K = wave_transpose64(E.R);              // allele j's classes = column j of the class masks (zero beyond A1 / C1)
    const bool use_len = lenc != nullptr;
    E.len = (use_len && lane < A1) ? lenc[lane] : 1.0;
    if (lane < A1) { out[lane] = -1.0; first_out[lane] = E.K ? __builtin_ctzll(E.K) : -1; }
    if (lane == 0) scal[S_NCLS] = (double)C1;
    if (exact) {        // alleles arrived in name order: the reference's own summation order (bit-identical results)
        ref_em_run(E, remove_low, use_len, lane, out, scal);
        return;
    }
    bool pr;
    double p = wave_map(E, 0.0, false, 1.0, true, use_len, pr);
    const double tot = wave_sum_f64(pr ? p : 0.0);
    p = pr ? p / tot : 0.0;
    wave_em_run(E, p, pr, 0, remove_low, use_len, lane, out, scal);      // lane j writes out[j]: the compact result vector
}

// ------------------------------------------------------------------------------------------------------------
// Active-allele compaction.  At the exon level only group representatives occur in classes (core:86-115), so a third
// or more of the allele columns are all-zero; the EM matrices are restricted to the alleles that occur at all.
// ------------------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------------------
// Table-lookup form of the bit mat-vec ("four Russians").  The EXEC-masked kernel above spends one vector instruction
// (plus two scalar ones and a share of a cross-lane reduction) per 64 matrix bits of ONE row; here a lane owns a row
// and consumes 8 matrix bits per LDS lookup:
//   * K is cut into slabs of 512 elements = 64 groups of 8; a workgroup (slab, chunk of 1024 rows) first tabulates the
//     256 subset sums of every group of its slab (128 KB of LDS, ~1.3 additions per entry),
//   * then every lane walks its row's 8 words of the slab -- the matrix is stored word-transposed, M[k_word][row],
//     so a wave reads 512 contiguous bytes per word -- and adds table[group][byte]; all-zero words are skipped per wave,
//   * the per-slab partial sums go to memory and the LAST workgroup of a row chunk to finish (device-scope counter)
//     adds them in slab order and applies the same epilogue as k_bitmatvec.
// Summation order is fixed (groups ascending, slabs ascending), so results are reproducible from launch to launch.
// ------------------------------------------------------------------------------------------------------------
constexpr int LUT_G = 64;                  // groups per slab
constexpr int LUT_SLAB = LUT_G * 8;        // elements per slab
constexpr size_t LUT_LDS = (size_t)(LUT_G * 256 + LUT_SLAB) * 8;

// 256 subset sums of each of the 64 groups of xs[] (see k_lutmatvec)
__device__ __forceinline__ void lut_build(const double *xs, double *T, int tid) {
    const int g = tid >> 4, lo = tid & 15;
    const double x0 = xs[8 * g], x1 = xs[8 * g + 1], x2 = xs[8 * g + 2], x3 = xs[8 * g + 3];
    const double x4 = xs[8 * g + 4], x5 = xs[8 * g + 5], x6 = xs[8 * g + 6], x7 = xs[8 * g + 7];
    const double L = (((lo & 1 ? x0 : 0.0) + (lo & 2 ? x1 : 0.0)) + (lo & 4 ? x2 : 0.0)) + (lo & 8 ? x3 : 0.0);
    double *Tg = T + g * 256 + lo;
#pragma unroll
    for (int hi = 0; hi < 16; ++hi) {
        const double H = (((hi & 1 ? x4 : 0.0) + (hi & 2 ? x5 : 0.0)) + (hi & 4 ? x6 : 0.0)) + (hi & 8 ? x7 : 0.0);
        Tg[hi * 16] = L + H;
    }
}
__device__ __forceinline__ double lut_row(const double *T, const uint64_t (&w)[8]) {
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (__ballot(w[i] != 0ull) == 0ull) continue;
        const uint32_t wl = (uint32_t)w[i], wh = (uint32_t)(w[i] >> 32);
        const double *Ti = T + i * 8 * 256;
        acc += Ti[0 * 256 + (wl & 255u)];
        acc += Ti[1 * 256 + ((wl >> 8) & 255u)];
        acc += Ti[2 * 256 + ((wl >> 16) & 255u)];
        acc += Ti[3 * 256 + (wl >> 24)];
        acc += Ti[4 * 256 + (wh & 255u)];
        acc += Ti[5 * 256 + ((wh >> 8) & 255u)];
        acc += Ti[6 * 256 + ((wh >> 16) & 255u)];
        acc += Ti[7 * 256 + (wh >> 24)];
    }
    return acc;
}
// lut_row for matrix words that live in registers across many passes (k_em_grid): the words are passed through an empty
// asm so that the compiler cannot hoist the 64 table addresses of a row out of the iteration loop (it did: 128 loop-invariant
// VGPRs, spilled to scratch and re-loaded before every lookup)
[[maybe_unused]] __device__ __forceinline__ double lut_row_resident(const double *T, const uint64_t (&w)[8]) {
    uint64_t v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint32_t lo = (uint32_t)w[i], hi = (uint32_t)(w[i] >> 32);
        asm volatile("" : "+v"(lo), "+v"(hi));
        v[i] = (uint64_t)hi << 32 | lo;
    }
    return lut_row(T, v);
}
template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_lutmatvec(const uint64_t *__restrict__ M, int N, int Npad, int n_k,
                                                     const double *__restrict__ vec, const uint8_t *__restrict__ vec_pres,
                                                     int x_mode, const int64_t *__restrict__ count,
                                                     const double *__restrict__ q_in, const uint8_t *__restrict__ pres_in,
                                                     const double *__restrict__ len, double *__restrict__ y,
                                                     uint8_t *__restrict__ pres_out, double *__restrict__ scal, int gate,
                                                     double *__restrict__ part, unsigned *__restrict__ counters,
                                                     const double *__restrict__ src_part, int src_slabs, int src_pad,
                                                     const int64_t *__restrict__ src_count, int defer_combine) {
    // src_part   (COLS): the rows pass left its slab partials uncombined (defer_combine); this kernel's prologue adds them
    //            for the 512 classes of its slab -- w_c = n_c / sum of the partials, in slab order -- instead of reading w
    // defer_combine (ROWS): store the slab partials and stop: no ticket, no last-workgroup tail (~4 us of the launch)
    extern __shared__ double lds[];
    double *T = lds;                        // [LUT_G][256]
    double *xs = lds + LUT_G * 256;         // [LUT_SLAB]
    __shared__ double sh[NWAVE];
    __shared__ int is_last;
    const int tid = threadIdx.x;
    const int slab = blockIdx.x, chunk = blockIdx.y, n_slabs = gridDim.x;
    const int n = chunk * BLOCK + tid;
    const double st_done = scal[S_DONE], st_flag = scal[S_FLAG];
    double tot = 1.0;
    if (MODE == MODE_COLS) tot = scal[S_TOT_A];
    // this slab's matrix words of my row: requested before anything else
    uint64_t w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = n < Npad ? M[(size_t)(slab * 8 + i) * Npad + n] : 0ull;
    double xv = 0.0;
    if (tid < LUT_SLAB) {
        const int e = slab * LUT_SLAB + tid;
        if (e < n_k) {
            if (MODE == MODE_ROWS) xv = (x_mode == 2) ? 1.0 : (vec_pres[e] ? vec[e] : 0.0);
            else if (src_part) {
                double sp = 0.0;
                for (int k0 = 0; k0 < src_slabs; k0 += 16) {      // sixteen loads in flight, added in slab order
                    double v[16];
#pragma unroll
                    for (int k = 0; k < 16; ++k) v[k] = k0 + k < src_slabs ? src_part[(size_t)(k0 + k) * src_pad + e] : 0.0;
#pragma unroll
                    for (int k = 0; k < 16; ++k) sp += v[k];
                }
                xv = sp > 0.0 ? (double)src_count[e] / sp : 0.0;
            } else xv = vec[e];
        }
    }
    // epilogue operands of my row (only the chunk's last workgroup uses them; asking now takes them off its critical path)
    double e_count = 0.0, e_q = 0.0, e_len = 1.0;
    bool e_pres = true;
    if (n < N) {
        if (MODE == MODE_ROWS) e_count = (double)count[n];
        else {
            if (x_mode != 2) { e_q = q_in[n]; e_pres = pres_in[n] != 0; }
            if (len) e_len = len[n];
        }
    }
    if (st_done != 0.0) return;
    if (gate && st_flag == 0.0) return;
    if (MODE == MODE_COLS && x_mode != 2) {
        // a chunk without a present allele produces zeros whatever the matrix says (every workgroup of the chunk agrees)
        if (!__syncthreads_or(n < N && e_pres)) {
            if (slab == 0 && n < N) { y[n] = 0.0; pres_out[n] = 0; }
            if (slab == 0 && chunk == 0 && tid == 0) scal[S_NCOLS] += 1.0;
            return;
        }
    }
    if (MODE == MODE_ROWS && x_mode == 1) {
        double s = 0.0;
        for (int e = tid; e < n_k; e += BLOCK) if (vec_pres[e]) s += vec[e];
        tot = block_sum(s, sh);
        xv = xv / tot;
    }
    if (slab == 0 && chunk == 0 && tid == 0) {
        if (MODE == MODE_ROWS) { scal[S_TOT_A] = tot; scal[S_NROWS] += 1.0; }
        else scal[S_NCOLS] += 1.0;
    }
    if (tid < LUT_SLAB) xs[tid] = xv;
    __syncthreads();
    lut_build(xs, T, tid);              // subset sums: thread = (group, low nibble); the 16 high nibbles are unrolled
    __syncthreads();
    const double acc = lut_row(T, w);
    if (MODE == MODE_ROWS && defer_combine) {
        if (n < Npad) part[(size_t)slab * Npad + n] = acc;       // the cols pass adds the slabs (kernel boundary = visibility)
        return;
    }
    // ---- the last workgroup of this row chunk to finish adds the slabs ------------------------------------------
    // No device-wide fence (a release fence at agent scope writes the whole L2 back: ~100 us measured).  Instead the
    // partials are device-scope atomic stores (written through to memory, past this XCD's L2), the workgroup waits for
    // its own stores to complete, then one device-scope atomic bumps the chunk's counter; the workgroup that draws
    // the last ticket reads the partials with device-scope loads (never cached in its L2 during this launch).
    if (n < Npad)
        __hip_atomic_store((unsigned long long *)&part[(size_t)slab * Npad + n], (unsigned long long)__double_as_longlong(acc),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(&counters[chunk], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = old == (unsigned)(n_slabs - 1);
        if (is_last) __hip_atomic_store(&counters[chunk], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!is_last) return;
    if (n < N) {
        double t = 0.0;
        for (int s0 = 0; s0 < n_slabs; s0 += 32) {          // up to 32 loads in flight, added in slab order
            double v[32];
#pragma unroll
            for (int k = 0; k < 32; ++k)
                v[k] = s0 + k < n_slabs
                           ? __longlong_as_double((long long)__hip_atomic_load((unsigned long long *)&part[(size_t)(s0 + k) * Npad + n],
                                                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                           : 0.0;
#pragma unroll
            for (int k = 0; k < 32; ++k) t += v[k];
        }
        if (MODE == MODE_ROWS) {
            y[n] = t > 0.0 ? e_count / t : 0.0;
        } else {
            const bool init = x_mode == 2;
            const bool in = init || e_pres;
            double v = 0.0;
            if (in && t > 0.0) {
                v = init ? t : (e_q / tot) * t;
                if (len) v = v / e_len;
            }
            y[n] = v;
            pres_out[n] = (in && t > 0.0) ? 1 : 0;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Rows pass with the preceding VECTOR step folded into its prologue (table-lookup EM, vectors <= 8192 elements).
// SQUAREM (k_em_squarem) and the advance step (k_em_advance) are single-workgroup kernels of ~10 us each -- launch floor
// plus four dependent round trips for 4 608 elements -- sitting between the mat-vec launches of every iteration.
// Here EVERY workgroup of the following rows pass recomputes the step's reductions itself (same element-to-thread
// map and summation order as the standalone kernels, so the same bits) while its matrix words are in flight:
//   FM = 0: SQUAREM on (p, q1, q2) -> x = extrapolated q2' (raw); no extrapolation / key error -> the map is skipped
//   FM = 1: prob_diff + pruning on (p, q1 | q3) -> x = new p (raw); converged -> nothing else runs
// The chunk-0 workgroup of each slab writes its 512 elements of the step's output vector (a SEPARATE buffer: the
// other workgroups are still reading the inputs), workgroup (0,0) writes the next state words to scal_out -- the
// launch itself only reads scal_in, so no workgroup can observe a half-updated state.
// ------------------------------------------------------------------------------------------------------------
struct FuseArgs {
    const double *q1, *qb;            // q1 and (FM 0: q2 | FM 1: q3)
    const uint8_t *pr1, *prb;
    double *out;                      // FM 0: q2' | FM 1: new p
    uint8_t *out_pres;
    const double *scal_in;
    double *scal_out;
    int remove_low;
};

template <int FM>
__global__ __launch_bounds__(BLOCK) void k_lut_rows_fused(const uint64_t *__restrict__ M, int Npad, int n_k,
                                                          const double *__restrict__ p, const uint8_t *__restrict__ pr,
                                                          FuseArgs fz, double *__restrict__ part) {
    extern __shared__ double lds[];
    double *T = lds;
    double *xs = lds + LUT_G * 256;
    __shared__ double shf[3][NWAVE];
    __shared__ double shm[NWAVE];
    const int tid = threadIdx.x;
    const int slab = blockIdx.x, chunk = blockIdx.y;
    const int n = chunk * BLOCK + tid;
    const double st_done = fz.scal_in[S_DONE], st_flag = fz.scal_in[S_FLAG], it_d = fz.scal_in[S_ITER];
    uint64_t w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = n < Npad ? M[(size_t)(slab * 8 + i) * Npad + n] : 0ull;
    // the whole vectors, element tid + 1024 k in register k (as in k_em_squarem / k_em_advance)
    double vp[EPT], v1[EPT], vb[EPT];
    uint8_t f0[EPT], f1[EPT], fb[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int a = tid + BLOCK * k;
        const bool in = a < n_k;
        f0[k] = in ? pr[a] : 0; f1[k] = in ? fz.pr1[a] : 0; fb[k] = in ? fz.prb[a] : 0;
        vp[k] = in ? p[a] : 0.0; v1[k] = in ? fz.q1[a] : 0.0; vb[k] = in ? fz.qb[a] : 0.0;
    }
    // my element of this slab
    const int e = slab * LUT_SLAB + tid;
    const bool e_in = tid < LUT_SLAB && e < n_k;
    const double e0 = e_in ? p[e] : 0.0, e1 = e_in ? fz.q1[e] : 0.0, eb = e_in ? fz.qb[e] : 0.0;
    const bool g0 = e_in && pr[e], g1 = e_in && fz.pr1[e], gb = e_in && fz.prb[e];
    const bool lead = slab == 0 && chunk == 0 && tid == 0;
    auto carry_state = [&]() {
        for (int i = 0; i < S_N; ++i) fz.scal_out[i] = fz.scal_in[i];
    };
    if (st_done != 0.0) {
        // already finished: carry the state words -- and, for the advance form, the estimate -- into the buffers the host
        // switches to after every fused launch
        if (lead) carry_state();
        if (FM == 1 && chunk == 0 && e_in) { fz.out[e] = e0; fz.out_pres[e] = g0 ? 1 : 0; }
        return;
    }
    double xv = 0.0;
    if (FM == 0) {
        // ---- SQUAREM (common:1361-1380), arithmetic of k_em_squarem ----
        double red[2] = {0.0, 0.0};
#pragma unroll
        for (int k = 0; k < EPT; ++k) { if (f1[k]) red[0] += v1[k]; if (fb[k]) red[1] += vb[k]; }
        block_sum_n<2>(red, shf);
        const double tot1 = red[0], tot2 = red[1];
        double acc[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            if (!f0[k]) continue;
            if (!f1[k] || !fb[k]) { acc[2] = 1.0; continue; }
            const double p1 = v1[k] / tot1, p2 = vb[k] / tot2;
            const double r = p1 - vp[k];
            const double v = p2 - p1 - r;
            acc[0] += r * r;
            acc[1] += v * v;
        }
        block_sum_n<3>(acc, shf);
        const double tsr = acc[0], tsv = acc[1], tkey = acc[2];
        const bool ext = tsv > 0.0 && tkey == 0.0;
        if (lead) {
            carry_state();
            fz.scal_out[S_FLAG] = ext ? 1.0 : 0.0;
            if (tkey != 0.0) { fz.scal_out[S_KEYERR] = 1.0; fz.scal_out[S_DONE] = 1.0; }
            fz.scal_out[S_TOT_A] = 1.0;
        }
        if (!ext) return;                                   // the third map is skipped (or the EM stops on the key error)
        const double g = -sqrt(tsr / tsv);
        double val = eb;
        bool present = gb;
        if (g0) {
            const double p1 = e1 / tot1, p2 = eb / tot2;
            const double r = p1 - e0;
            const double v = p2 - p1 - r;
            val = fmax(0.0, e0 - 2 * g * r + g * g * v);
            present = true;
        }
        xv = present ? val : 0.0;
        if (chunk == 0 && e_in) { fz.out[e] = val; fz.out_pres[e] = present ? 1 : 0; }
    } else {
        // ---- prob_diff (common:1272-1279), pruning (common:1338-1346), stopping rule (common:1351): k_em_advance ----
        const bool ext = st_flag != 0.0;
        double s1[1] = {0.0};
#pragma unroll
        for (int k = 0; k < EPT; ++k) { const bool pr_ = ext ? fb[k] : f1[k]; if (pr_) s1[0] += ext ? vb[k] : v1[k]; }
        block_sum_n<1>(s1, shf);
        const double tot = s1[0];
        double d[1] = {0.0}, mx = 0.0;
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const bool an = ext ? fb[k] : f1[k];
            const double pn = an ? (ext ? vb[k] : v1[k]) / tot : 0.0;
            if (f0[k]) d[0] += an ? fabs(vp[k] - pn) : vp[k];
            if (an) mx = fmax(mx, pn);
        }
        const double tm = block_max(mx, shm);
        block_sum_n<1>(d, shf);
        const double td = d[0];
        const int iter = (int)it_d;
        const bool prune = fz.remove_low && iter >= 10;
        double kept[1] = {0.0};
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const bool an = ext ? fb[k] : f1[k];
            const double pn = an ? (ext ? vb[k] : v1[k]) / tot : 0.0;
            bool keep = an;
            if (prune && keep) keep = pn >= tm / 10.0;
            if (keep) kept[0] += 1.0;
        }
        block_sum_n<1>(kept, shf);
        const bool done = !(td > 0.0001) || iter + 1 >= 1000;
        if (lead) {
            carry_state();
            fz.scal_out[S_NPRES] = kept[0];
            fz.scal_out[S_DIFF] = td;
            fz.scal_out[S_ITER] = (double)(iter + 1);
            if (done) fz.scal_out[S_DONE] = 1.0;
            fz.scal_out[S_TOT_A] = 1.0;
        }
        const bool an = ext ? gb : g1;
        const double pn = an ? (ext ? eb : e1) / tot : 0.0;
        bool keep = an;
        if (prune && keep) keep = pn >= tm / 10.0;
        xv = keep ? pn : 0.0;
        if (chunk == 0 && e_in) { fz.out[e] = keep ? pn : 0.0; fz.out_pres[e] = keep ? 1 : 0; }
        if (done) return;                                   // the new estimate is written; nothing else runs
    }
    if (tid < LUT_SLAB) xs[tid] = xv;
    __syncthreads();
    lut_build(xs, T, tid);
    __syncthreads();
    const double acc2 = lut_row(T, w);
    if (n < Npad) part[(size_t)slab * Npad + n] = acc2;      // the cols pass adds the slabs
}



// arguments of the resident-block EM (k_em_grid, lab build): the host code that prepares them is shared
struct GkArgs {
    const uint64_t *Mr, *Mc;          // word-transposed matrices [w64c][Cp], [c64][A]
    int C, Cp, A;                     // classes, padded classes (multiple of 512), compact padded alleles (multiple of 512)
    int R, K;                         // class chunks of 1024, allele slabs of 512
    const int64_t *count;
    const double *len;
    double *p;                        // estimate in / out
    uint8_t *pr;
    double *part_r, *part_c, *Y;      // [2][K][Cp], [2][Cp/512][A], [2][R][A]
    unsigned *flags;                  // [3][R*K] words, 128 bytes apart, zero at launch
    double *scal;
    int *abort_flag;
    int remove_low, n_iters, do_init;
    unsigned long long *stamps;       // HGX_GRID_STAMPS=1: wall_clock64() of thread 0 at the phase boundaries, [R*K][GK_STAMPS]
};
constexpr int GK_STAMPS = 128;
[[maybe_unused]] constexpr long GK_SPIN_LIMIT = 1500000;     // bounded spin (~0.5 s): an error code, never a hung GPU
constexpr int GK_FLAG_STRIDE = 32;
[[maybe_unused]] constexpr size_t GK_LDS = (size_t)(LUT_G * 256 + BLOCK) * 8;

// u64-element transpose: out[c][r] = in[r][c]  (in [n_rows][n_cols])
__global__ __launch_bounds__(256) void k_word_transpose(const uint64_t *__restrict__ in, int n_rows, int n_cols,
                                                        uint64_t *__restrict__ out) {
    __shared__ uint64_t tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + tx;
        tile[k][tx] = (r < n_rows && c < n_cols) ? in[(size_t)r * n_cols + c] : 0ull;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, r = r0 + tx;
        if (c < n_cols && r < n_rows) out[(size_t)c * n_rows + r] = tile[tx][k];
    }
}

struct MatVec {
    const uint64_t *B;
    int n_rows, n_words, n_k;
    const uint64_t *P = nullptr;    // MFMA operand order of B (k_permute_mfma), or nullptr -> VALU kernel
    int n_super = 0;
    const uint64_t *M = nullptr;    // word-transposed B [n_words][n_pad] for the table-lookup kernel, or nullptr
    int n_pad = 0;
    double *part = nullptr;         // [n_words / 8][n_pad] slab partials
    unsigned *counters = nullptr;   // [ceil(n_rows / 1024)], zero between launches
    // table-lookup EM only: the rows pass leaves its partials for the cols pass to add (see k_lutmatvec)
    int narrow = 0;                 // the narrow-table form (k_lut4): a workgroup owns its rows for the whole of K
    int defer_combine = 0;
    const double *src_part = nullptr;
    int src_slabs = 0, src_pad = 0;
    const int64_t *src_count = nullptr;
};

int g_backend = 0;                  // 0 auto, 1 VALU (EXEC-masked FP64), 2 MFMA (int8 fixed point), 3 table lookup

inline bool use_mfma(const MatVec &m) {
    if (!m.P) return false;
    if (g_backend == 1) return false;
    if (g_backend == 2) return true;
    return false;   // auto: at one-sample problem sizes the VALU kernel is faster; MFMA is opt-in
}

template <int MODE>
int launch_mfma(const MatVec &m, hipStream_t st, const double *vec, const uint8_t *vec_pres, int x_mode, const int64_t *count,
                const double *q_in, const uint8_t *pres_in, const double *len, double *y, uint8_t *pres_out, double *scal,
                int gate) {
    hgx_set_error("this EM back-end is lab code: build libhgx_lab.so (hisat-genotype_amd/build.py build_lab) -- libhgx.so ships the table-lookup and reference-order paths only");
    return HGX_EINVAL;
}

// events that the next table-lookup launch attaches to its own dispatch (hipExtLaunchKernelGGL: the runtime timestamps the
// kernel's begin and end, like rocprofv3 does, instead of bracketing it with separately queued event records)
thread_local hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr;

template <int MODE>
int launch_matvec(const MatVec &m, hipStream_t st, const double *vec, const uint8_t *vec_pres, int x_mode,
                  const int64_t *count, const double *q_in, const uint8_t *pres_in, const double *len, double *y,
                  uint8_t *pres_out, double *scal, int gate) {
    if constexpr (MODE == MODE_ROWS || MODE == MODE_COLS) {
        if (use_mfma(m))
            return launch_mfma<MODE>(m, st, vec, vec_pres, x_mode, count, q_in, pres_in, len, y, pres_out, scal, gate);
        if (m.M) {
            HGX_ONCE_PER_DEVICE({
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lutmatvec<MODE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)LUT_LDS));
            });
            if (g_ev_start) {
                hipExtLaunchKernelGGL((k_lutmatvec<MODE>), dim3(m.n_words / 8, (m.n_rows + BLOCK - 1) / BLOCK), dim3(BLOCK), LUT_LDS, st,
                                      g_ev_start, g_ev_stop, 0, m.M, m.n_rows, m.n_pad, m.n_k, vec, vec_pres, x_mode, count, q_in, pres_in,
                                      len, y, pres_out, scal, gate, m.part, m.counters, m.src_part, m.src_slabs, m.src_pad, m.src_count,
                                      m.defer_combine);
                g_ev_start = g_ev_stop = nullptr;
            } else {
                hipLaunchKernelGGL((k_lutmatvec<MODE>), dim3(m.n_words / 8, (m.n_rows + BLOCK - 1) / BLOCK), dim3(BLOCK), LUT_LDS, st, m.M,
                                   m.n_rows, m.n_pad, m.n_k, vec, vec_pres, x_mode, count, q_in, pres_in, len, y, pres_out, scal, gate,
                                   m.part, m.counters, m.src_part, m.src_slabs, m.src_pad, m.src_count, m.defer_combine);
            }
            return HGX_OK;
        }
    }
#ifdef HGX_LAB
    const int rpb = rows_per_block(m.n_rows);
    const int grid = (m.n_rows + rpb - 1) / rpb;
    if (m.n_k <= 8 * BLOCK)
        hipLaunchKernelGGL((k_bitmatvec<8, MODE>), dim3(grid), dim3(BLOCK), 0, st, m.B, m.n_rows, m.n_words, m.n_k, rpb, vec,
                           vec_pres, x_mode, count, q_in, pres_in, len, y, pres_out, scal, gate);
    else
        hipLaunchKernelGGL((k_bitmatvec<16, MODE>), dim3(grid), dim3(BLOCK), 0, st, m.B, m.n_rows, m.n_words, m.n_k, rpb, vec,
                           vec_pres, x_mode, count, q_in, pres_in, len, y, pres_out, scal, gate);
    return HGX_OK;
#else
    hgx_set_error("bit mat-vec without table-lookup operands: the VALU back-end is lab code (libhgx_lab.so)");
    return HGX_EINVAL;
#endif
}

// Per-kernel timing for bench.py's roofline object.  When enabled (hgx_em_set_timing) every bit-mat-vec launch of
// hgx_em is bracketed by HIP events on its stream; totals accumulate per kernel instantiation until reset.
struct PassStats { double ms = 0; int64_t launches = 0, executed = 0, bytes = 0; };
thread_local PassStats g_stats[5];   // [0] <8,ROWS> [1] <16,ROWS> [2] <8,COLS> [3] <16,COLS> [4] k_em_grid (executed = applications)
thread_local int g_timing = 0;
struct Timed { hipEvent_t a, b; int slot; };
thread_local std::vector<hipEvent_t> g_event_pool;     // recycled HIP events (creating one per launch is not free)
inline hipEvent_t pool_event() {
    if (!g_event_pool.empty()) { hipEvent_t e = g_event_pool.back(); g_event_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreateWithFlags(&e, hipEventDefault);
    return e;
}

}   // namespace

// bit-matrix transpose kernel of hgx_dedup.hip: [n_rows][w_in] -> [w_in * 64][w_out]
__global__ void k_transpose(const uint64_t *bits, int n_classes, int w64, int c64, uint64_t *bitsT);

// OR of all class rows: which alleles occur at all.  A workgroup takes a band of rows, thread w ORs word w of every row of the
// band in a register (consecutive threads read consecutive words; eight rows in flight) and issues one global atomic at the end.
// (An LDS atomicOr per matrix word -- 1.8 M of them on w64 addresses -- took 20 us.)
__global__ __launch_bounds__(256) void k_col_or(const uint64_t *__restrict__ B, int n_rows, int w64,
                                                unsigned long long *__restrict__ mask) {
    const int per = (n_rows + gridDim.x - 1) / gridDim.x;
    const int r0 = blockIdx.x * per, r1 = min(n_rows, r0 + per);
    for (int w = threadIdx.x; w < w64; w += blockDim.x) {
        uint64_t acc = 0ull;
        int r = r0;
        for (; r + 8 <= r1; r += 8) {
            uint64_t v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = B[(size_t)(r + k) * w64 + w];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc |= v[k];
        }
        for (; r < r1; ++r) acc |= B[(size_t)r * w64 + w];
        if (acc) atomicOr(&mask[w], (unsigned long long)acc);
    }
}
// base[w] = number of active alleles in the words before w; act[] = the active alleles in ascending order (w64 <= 512)
__global__ __launch_bounds__(512) void k_act_from_mask(const unsigned long long *__restrict__ mask, int w64, int32_t *__restrict__ base,
                                                       int32_t *__restrict__ act) {
    __shared__ int pc[512];
    const int w = threadIdx.x;
    const unsigned long long m = w < w64 ? mask[w] : 0ull;
    pc[w] = __popcll(m);
    __syncthreads();
    for (int d = 1; d < 512; d <<= 1) {                  // inclusive scan
        const int v = w >= d ? pc[w - d] : 0;
        __syncthreads();
        pc[w] += v;
        __syncthreads();
    }
    if (w >= w64) return;
    int j = pc[w] - __popcll(m);
    base[w] = j;
    for (unsigned long long mm = m; mm; mm &= mm - 1) act[j++] = 64 * w + __builtin_ctzll(mm);
}
// rows [n, a1p) of bitsTC[a1p][c64] and the same entries of its word-transposed copy wcol[c64][a1p]
__global__ void k_zero_padding(uint64_t *__restrict__ bitsTC, uint64_t *__restrict__ wcol, int n, int a1p, int c64) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)(a1p - n) * c64) return;
    const int j = n + (int)(i / c64), cw = (int)(i % c64);
    bitsTC[(size_t)j * c64 + cw] = 0ull;
    wcol[(size_t)cw * a1p + j] = 0ull;
}
// bit transpose [C][w64] -> [a1p][c64] that keeps only the active alleles: row of allele (w, b) = base[w] + rank of b
// among the active bits of word w.  One wavefront per 64 x 64 tile (cf. k_transpose).
__global__ __launch_bounds__(256) void k_transpose_compact(const uint64_t *__restrict__ bits, int n_classes, int w64, int c64,
                                                           const unsigned long long *__restrict__ mask,
                                                           const int32_t *__restrict__ base, uint64_t *__restrict__ bitsTC,
                                                           uint64_t *__restrict__ wcol, int a1p) {
    const int lane = threadIdx.x & 63;
    const long tile = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long n_tiles = (long)c64 * w64;
    if (tile >= n_tiles) return;
    const int cw = (int)(tile / w64), aw = (int)(tile % w64);
    const uint64_t m = mask[aw];
    if (m == 0ull) return;
    const int c = cw * 64 + lane;
    const uint64_t x = (c < n_classes) ? bits[(size_t)c * w64 + aw] : 0ull;
    const uint64_t mine = wave_transpose64(x);
    if ((m >> lane) & 1ull) {
        const int j = base[aw] + __popcll(m & ((1ull << lane) - 1ull));
        bitsTC[(size_t)j * c64 + cw] = mine;
        wcol[(size_t)cw * a1p + j] = mine;              // the same word in the word-transposed copy (k_lutmatvec's cols pass)
    }
}

// second transpose of the set-up: compact [a1p][c64] -> class-major [c64 * 64][a1p / 64], written twice: row-major (tail
// kernel, MFMA order) and word-transposed (k_lutmatvec's rows pass)
__global__ __launch_bounds__(256) void k_transpose_dual(const uint64_t *__restrict__ bitsTC, int a1p, int c64,
                                                        uint64_t *__restrict__ bitsC, uint64_t *__restrict__ wrow) {
    const int lane = threadIdx.x & 63;
    const long tile = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int w64c = a1p / 64;
    const long n_tiles = (long)w64c * c64;
    if (tile >= n_tiles) return;
    const int aw = (int)(tile / c64), cw = (int)(tile % c64);          // 64 alleles x 64 classes
    const uint64_t x = bitsTC[(size_t)(aw * 64 + lane) * c64 + cw];
    const uint64_t mine = wave_transpose64(x);
    const size_t cls = (size_t)cw * 64 + lane;
    bitsC[cls * w64c + aw] = mine;
    wrow[(size_t)aw * ((size_t)c64 * 64) + cls] = mine;
}

static int hgx_ensure_compact(hgx_classes *c, hipStream_t st) {
    if (c->n_act >= 0) return HGX_OK;
    const int A = c->a_pad, w64 = c->w64, C = c->n_classes;
    c->c64 = ((C + 63) / 64 + 7) / 8 * 8;            // row stride of the transposed matrices: multiple of 8 words, zero padded
    // the mask / base tables are read by kernels queued below: they stay with the class set instead of forcing a sync here
    DevBuf b_mask, b_base;
    ALLOC(b_mask, (size_t)w64 * 8); ALLOC(b_base, (size_t)w64 * 4);
    c->d_setup0 = b_mask.p; c->d_setup1 = b_base.p;
    struct Release { DevBuf &a, &b; ~Release() { a.p = nullptr; b.p = nullptr; } } release{b_mask, b_base};   // owned by *c from here on
    HIPCHK(hipMemsetAsync(b_mask.p, 0, (size_t)w64 * 8, st));
    hipLaunchKernelGGL(k_col_or, dim3(std::min(512, std::max(1, C / 16))), dim3(w64 <= 128 ? 128 : 256), 0, st, c->d_bits, C, w64,
                       b_mask.as<unsigned long long>());
    // the device-side tables that follow from the mask (rank base per word, list of active alleles) are built on the device,
    // queued behind the OR: the host derives its own copy from the mask it fetches below, and no upload sits between the
    // round trip and the transposes (an 18 KB host-to-device copy took 29 us of device time plus the gaps around it)
    c->d_act = (int32_t *)hgx_pool_alloc((size_t)A * 4);
    if (!c->d_act) { hgx_set_error("device allocation failed"); return HGX_ENOMEM; }
    const bool dev_tables = w64 <= 512;               // (wider allele sets -- beyond the scoring path's limit -- upload the host's copy)
    if (dev_tables)
        hipLaunchKernelGGL(k_act_from_mask, dim3(1), dim3(512), 0, st, b_mask.as<unsigned long long>(), w64, b_base.as<int32_t>(), c->d_act);
    HIPCHK(hipGetLastError());
    std::vector<uint64_t> h_mask(w64);
    { int rc_ = hgx_d2h(h_mask.data(), b_mask.p, (size_t)w64 * 8, st); if (rc_) return rc_; }
    { int rc_ = hgx_sync(st); if (rc_) return rc_; }
    std::vector<int32_t> h_base(w64);
    c->h_act = new int32_t[A];
    int32_t n = 0;
    for (int w = 0; w < w64; ++w) {
        h_base[w] = n;
        for (uint64_t m = h_mask[w]; m; m &= m - 1) c->h_act[n++] = 64 * w + __builtin_ctzll(m);
    }
    if (!dev_tables) {
        { int rc_ = hgx_h2d(b_base.p, h_base.data(), (size_t)w64 * 4, st); if (rc_) return rc_; }
        { int rc_ = hgx_h2d(c->d_act, c->h_act, (size_t)std::max(n, 1) * 4, st); if (rc_) return rc_; }
    }
    const int a1p = std::max(512, (n + 511) / 512 * 512);
    c->d_bitsTC = (uint64_t *)hgx_pool_alloc((size_t)a1p * c->c64 * 8);
    c->d_bitsC = (uint64_t *)hgx_pool_alloc((size_t)c->c64 * 64 * (a1p / 64) * 8);
    c->d_wrow = (uint64_t *)hgx_pool_alloc((size_t)(a1p / 64) * c->c64 * 64 * 8);
    c->d_wcol = (uint64_t *)hgx_pool_alloc((size_t)c->c64 * a1p * 8);
    if (!c->d_act || !c->d_bitsTC || !c->d_bitsC || !c->d_wrow || !c->d_wcol) { hgx_set_error("device allocation failed"); return HGX_ENOMEM; }
    // padding rows [n, a1p) stay zero (in the word-transposed copy they are columns [n, a1p) of every row): one small launch
    if (a1p > n)
        hipLaunchKernelGGL(k_zero_padding, dim3(nblk((long)(a1p - n) * c->c64, 256)), dim3(256), 0, st, c->d_bitsTC, c->d_wcol, n, a1p,
                           c->c64);
    const long tiles_in = (long)c->c64 * w64;
    hipLaunchKernelGGL(k_transpose_compact, dim3(nblk(tiles_in, 4)), dim3(256), 0, st, c->d_bits, C, w64, c->c64,
                       b_mask.as<unsigned long long>(), b_base.as<int32_t>(), c->d_bitsTC, c->d_wcol, a1p);
    const long tiles = (long)(a1p / 64) * c->c64;
    hipLaunchKernelGGL(k_transpose_dual, dim3(nblk(tiles, 4)), dim3(256), 0, st, c->d_bitsTC, a1p, c->c64, c->d_bitsC, c->d_wrow);
    HIPCHK(hipGetLastError());
    c->a1p = a1p;
    c->n_act = n;
    return HGX_OK;
}

// 1 if the calling thread's last EM (hgx_em / hgx_em_ordered / hgx_em_masked) ran on the single-wavefront path in the reference's
// own order of floating-point operations (its abundances are then the reference's, bit for bit), 0 otherwise
static thread_local int g_last_exact = 0;
static thread_local std::vector<int32_t> g_last_order;     // per allele: position in the returned dict's insertion order (hgx_em_last_order)
static thread_local int g_em_fast = 0;           // hgx_em_set_fast: 1 = table-lookup arithmetic on the one-workgroup path (k_emx), -1 = the
                                                 // reference's order at every size (k_emx also beyond 4096 classes)
// Insertion order of the dict the calling thread's last hgx_em / hgx_em_ordered returned, when that EM ran in the reference's own
// order on the one-workgroup kernel (hgx_em_last_exact() == 1 and the problem had more than 64 classes or alleles): order_host[a] =
// position of allele a (-1 if not in the dict).  Returns 1 if available, else 0 (callers then use (first class, name order)).
extern "C" int hgx_em_last_order(int32_t *order_host, int32_t n) {
    if (g_last_order.empty() || !order_host || n > (int32_t)g_last_order.size()) return 0;
    memcpy(order_host, g_last_order.data(), (size_t)n * 4);
    return 1;
}
extern "C" int hgx_em_set_fast(int on) { const int old = g_em_fast; g_em_fast = on > 0 ? 1 : (on < 0 ? -1 : 0); return old; }
static thread_local bool g_no_grid = false;      // set while an EM is re-run after a resident-block launch was abandoned
extern "C" int hgx_em_last_exact(void) { return g_last_exact; }

extern "C" int hgx_em_set_backend(int backend) {
    ARGCHK(backend >= 0 && backend <= 3);
#ifndef HGX_LAB
    if (backend == 1 || backend == 2) { hgx_set_error("this EM back-end is lab code: build libhgx_lab.so (hisat-genotype_amd/build.py build_lab) -- libhgx.so ships the table-lookup and reference-order paths only"); return HGX_EINVAL; }
#endif
    g_backend = backend;
    return HGX_OK;
}

extern "C" int hgx_em_set_timing(int on) {
    ARGCHK(on >= 0 && on <= 2);
    if (on && !g_timing) for (auto &s : g_stats) s = PassStats();      // totals restart when timing is switched on
    g_timing = on;
    return HGX_OK;
}
// slot: 0 <8,ROWS>, 1 <16,ROWS>, 2 <8,COLS>, 3 <16,COLS>, 4 k_em_grid (whole launches; executed = applications of the EM map)
extern "C" int hgx_em_get_timing(int slot, double *ms_total, int64_t *launches, int64_t *executed, int64_t *bytes_total) {
    ARGCHK(slot >= 0 && slot < 5);
    if (ms_total) *ms_total = g_stats[slot].ms;
    if (launches) *launches = g_stats[slot].launches;
    if (executed) *executed = g_stats[slot].executed;
    if (bytes_total) *bytes_total = g_stats[slot].bytes;
    return HGX_OK;
}

static int em_impl(const hgx_classes *cc, int32_t n_alleles, int32_t remove_low, const int32_t *allele_len, double *prob_host,
                   int32_t *first_host, int32_t *n_iter_host, void *stream);
static int em_impl_inner(const hgx_classes *cc, int32_t n_alleles, int32_t remove_low, const int32_t *allele_len, double *prob_host,
                         int32_t *first_host, int32_t *n_iter_host, void *stream);
__global__ void k_first_set_rows(const uint64_t *BT, int n_rows, int c64, int32_t *first);

extern "C" int hgx_em(const hgx_classes *cc, int32_t n_alleles, int32_t remove_low, const int32_t *allele_len,
                      double *prob_host, int32_t *n_iter_host, void *stream) {
    return em_impl(cc, n_alleles, remove_low, allele_len, prob_host, nullptr, n_iter_host, stream);
}
extern "C" int hgx_em_ordered(const hgx_classes *cc, int32_t n_alleles, int32_t remove_low, const int32_t *allele_len,
                              double *prob_host, int32_t *first_class_host, int32_t *n_iter_host, void *stream) {
    ARGCHK(first_class_host);
    const int rc = em_impl(cc, n_alleles, remove_low, allele_len, prob_host, first_class_host, n_iter_host, stream);
    if (rc == HGX_OK)
        for (int a = 0; a < n_alleles; ++a) if (prob_host[a] < 0.0) first_class_host[a] = -1;     // only the returned dict's alleles
    return rc;
}

// ---- near-ties of the table-lookup path -------------------------------------------------------------------------
// The reference ranks with a plain stable sort on its own doubles: two alleles whose abundances differ in the last bit are ordered by
// that bit.  The large-problem path is good to ~1e-11 (bound 1e-9): where two alleles of the result with DIFFERENT class membership
// come out closer than EM_NEAR_REL, their order in the reference cannot be told from these values -- the EM is run again in the
// reference's own order of operations (k_emx at any size, a lone big problem in cluster mode: bit-identical abundances, ~15x the
// time; rare: alleles the data cannot tell apart have identical columns and are exact ties on both paths, and distinct columns
// agreeing to eight digits takes a symmetric construction), and that result is returned as exact.
constexpr double EM_NEAR_REL = 1e-8;
static std::atomic<long long> g_tie_reruns{0};
extern "C" long long hgx_em_tie_reruns(void) { return g_tie_reruns.load(); }

// rows of the transposed class matrix (allele-major) gathered into one contiguous block: the near-tie check compares them on the host
__global__ void k_gather_rows(const uint64_t *__restrict__ bitsT, int c64, const int32_t *__restrict__ ids, int n, uint64_t *__restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)n * c64) return;
    out[i] = bitsT[(size_t)ids[i / c64] * c64 + (i % c64)];
}

static int em_uncertain_near_tie(hgx_classes *c, int32_t n_alleles, const double *prob, hipStream_t st, bool *uncertain) {
    *uncertain = false;
    std::vector<int32_t> al;
    for (int32_t a = 0; a < n_alleles; ++a) if (prob[a] > 0.0) al.push_back(a);
    std::sort(al.begin(), al.end(), [&](int32_t x, int32_t y) { return prob[x] > prob[y]; });
    // Only an order somebody reads is worth a ~15x slower exact run: the report prints abundances >= 0.01 and at most 20
    // alleles (typing_core.py:2081, 2118-2121), the exon -> gene hand-off walks the ranking until `rank >= 10 and p < 0.03`
    // (core:1739-1749).  Neighbours below rank 32 AND below 0.005 are vanishing residues whose relative order reaches no output.
    std::vector<std::pair<int32_t, int32_t>> pairs;
    for (size_t i = 0; i + 1 < al.size(); ++i) {
        const double hi = prob[al[i]], lo = prob[al[i + 1]];
        if (i >= 32 && hi < 0.005) break;
        if (hi != lo && hi - lo <= EM_NEAR_REL * hi) pairs.emplace_back(al[i], al[i + 1]);
    }
    if (pairs.empty()) return HGX_OK;
    int rc = hgx_ensure_transposed(c, st);
    if (rc) return rc;
    // the candidates' membership rows in ONE gather, one copy and one wait (it was two blocking row copies and a stream sync per pair)
    std::vector<int32_t> ids;
    for (auto &pr : pairs) { ids.push_back(pr.first); ids.push_back(pr.second); }
    const size_t c64 = (size_t)c->c64;
    DevBuf b_ids, b_rows;
    ALLOC(b_ids, ids.size() * 4);
    ALLOC(b_rows, ids.size() * c64 * 8);
    { int rc_ = hgx_h2d(b_ids.p, ids.data(), ids.size() * 4, st); if (rc_) return rc_; }
    const long n_words = (long)ids.size() * (long)c64;
    hipLaunchKernelGGL(k_gather_rows, dim3(nblk(n_words, 256)), dim3(256), 0, st, (const uint64_t *)c->d_bitsT, (int)c64, b_ids.as<int32_t>(), (int)ids.size(),
                       b_rows.as<uint64_t>());
    HIPCHK(hipGetLastError());
    std::vector<uint64_t> rows(ids.size() * c64);
    { int rc_ = hgx_d2h(rows.data(), b_rows.p, rows.size() * 8, st); if (rc_) return rc_; }
    { int rc_ = hgx_sync(st); if (rc_) return rc_; }
    for (size_t k = 0; k < pairs.size(); ++k)
        if (memcmp(&rows[(2 * k) * c64], &rows[(2 * k + 1) * c64], c64 * 8) != 0) { *uncertain = true; return HGX_OK; }     // different membership, too close to call
    return HGX_OK;
}

static int em_impl(const hgx_classes *cc, int32_t n_alleles, int32_t remove_low, const int32_t *allele_len, double *prob_host,
                   int32_t *first_host, int32_t *n_iter_host, void *stream) {
    int rc = em_impl_inner(cc, n_alleles, remove_low, allele_len, prob_host, first_host, n_iter_host, stream);
    if (rc || g_last_exact || g_em_fast != 0) return rc;                    // (exact already, or the caller chose the arithmetic)
    hgx_classes *c = const_cast<hgx_classes *>(cc);
    if (!c->h_rank || c->n_classes > HGX_EMX_HARD_MAX_CLASSES || c->w64 > 128 || hgx_switch_has("em_skip", "exact") ||
        hgx_switch_has("em_skip", "emx") || hgx_switch_has("em_skip", "tie_rerun"))
        return rc;                                                          // (the reference's order is not available for this problem)
    bool uncertain = false;
    rc = em_uncertain_near_tie(c, n_alleles, prob_host, (hipStream_t)stream, &uncertain);
    if (rc || !uncertain) return rc;
    g_tie_reruns.fetch_add(1);
    const int old = g_em_fast;
    g_em_fast = -1;
    rc = em_impl_inner(cc, n_alleles, remove_low, allele_len, prob_host, first_host, n_iter_host, stream);
    g_em_fast = old;
    return rc;
}

static int em_impl_inner(const hgx_classes *cc, int32_t n_alleles, int32_t remove_low, const int32_t *allele_len, double *prob_host,
                         int32_t *first_host, int32_t *n_iter_host, void *stream) {
    ARGCHK(cc && prob_host && n_alleles > 0 && n_alleles <= cc->a_pad);
    g_last_exact = 0;
    g_last_order.clear();
    hgx_classes_order_after(cc, (hipStream_t)stream);
    if (first_host) for (int a = 0; a < n_alleles; ++a) first_host[a] = -1;
    hgx_classes *c = const_cast<hgx_classes *>(cc);
    hipStream_t st = (hipStream_t)stream;
    int A = c->a_pad;
    const int C = c->n_classes;
    if (n_iter_host) *n_iter_host = 0;
    if (C == 0) {
        // single_abundance({}): `diff` starts at 1.0, so the loop makes ONE pass over the empty dict before it stops
        // (typing_common.py:1351-1404) -- an empty result after 1 iteration (tools/fuzz_many.py: a hand-off that leaves no class)
        for (int a = 0; a < n_alleles; ++a) prob_host[a] = -1.0;
        if (n_iter_host) *n_iter_host = 1;
        return HGX_OK;
    }
    if (C <= 64 && c->w64 <= 128 && !hgx_switch_has("em_skip", "wave")) {
        // single-wavefront path (<= 64 classes over <= 64 distinct alleles); falls through if more alleles occur
        DevBuf b_len, b_scal, b_out;
        ALLOC(b_scal, S_N * 8); ALLOC(b_out, A * 8);
        double *d_len = nullptr;
        if (allele_len) {
            std::vector<double> l(A, 1.0);
            for (int a = 0; a < n_alleles; ++a) l[a] = (double)allele_len[a];
            ALLOC(b_len, A * 8);
            { int rc_ = hgx_h2d(b_len.p, l.data(), A * 8, st); if (rc_) return rc_; }
            d_len = b_len.as<double>();
        }
        HIPCHK(hipMemsetAsync(b_scal.p, 0, S_N * 8, st));
        DevBuf b_first;
        std::vector<int32_t> h_first;
        if (first_host) {
            ALLOC(b_first, (size_t)A * 4);
            HIPCHK(hipMemsetAsync(b_first.p, 0xFF, (size_t)A * 4, st));
            h_first.resize(A);
        }
        DevBuf b_rank;
        if (c->h_rank && !hgx_switch_has("em_skip", "exact")) {
            ALLOC(b_rank, (size_t)A * 4);
            { int rc_ = hgx_h2d(b_rank.p, c->h_rank, (size_t)A * 4, st); if (rc_) return rc_; }
        }
        hipLaunchKernelGGL(k_em_wave, dim3(1), dim3(64), 0, st, c->d_bits, C, c->w64, A, c->d_count, d_len, remove_low ? 1 : 0,
                           b_out.as<double>(), b_scal.as<double>(), first_host ? b_first.as<int32_t>() : nullptr,
                           b_rank.as<int32_t>());
        HIPCHK(hipGetLastError());
        std::vector<double> out(A);
        double h_scal[S_N];
        if (first_host) { int rc_ = hgx_d2h(h_first.data(), b_first.p, (size_t)A * 4, st); if (rc_) return rc_; }
        { int rc_ = hgx_d2h(out.data(), b_out.p, A * 8, st); if (rc_) return rc_; }
        { int rc_ = hgx_d2h(h_scal, b_scal.p, S_N * 8, st); if (rc_) return rc_; }
        { int rc_ = hgx_sync(st); if (rc_) return rc_; }
        if (h_scal[S_FALLBACK] == 0.0) {
            if (h_scal[S_KEYERR] != 0.0) {
                hgx_set_error("EM: allele missing from the next estimate (the reference raises KeyError here, common:1365-1369)");
                return HGX_EKEY;
            }
            for (int a = 0; a < n_alleles; ++a) prob_host[a] = out[a];
            if (first_host) for (int a = 0; a < n_alleles; ++a) first_host[a] = h_first[a];
            if (n_iter_host) *n_iter_host = (int)h_scal[S_ITER];
            g_last_exact = b_rank.p != nullptr;
            return HGX_OK;
        }
    }
    if (C <= (g_em_fast < 0 ? HGX_EMX_HARD_MAX_CLASSES : HGX_EMX_MAX_CLASSES) && c->w64 <= 128 && c->h_rank && !hgx_switch_has("em_skip", "exact") &&
        !hgx_switch_has("em_skip", "emx")) {
        // problems of up to 4096 classes over up to 8192 distinct alleles in the reference's own order of operations
        // (k_emx, hgx_emx.hip: one workgroup, one launch, bit-identical abundances); falls through if it does not take the problem
        DevBuf b_rank, b_len;
        ALLOC(b_rank, (size_t)A * 4);
        { int rc_ = hgx_h2d(b_rank.p, c->h_rank, (size_t)A * 4, st); if (rc_) return rc_; }
        if (allele_len) {
            std::vector<double> l(A, 1.0);
            for (int a = 0; a < n_alleles; ++a) l[a] = (double)allele_len[a];
            ALLOC(b_len, (size_t)A * 8);
            { int rc_ = hgx_h2d(b_len.p, l.data(), (size_t)A * 8, st); if (rc_) return rc_; }
        }
        hgx_emx_job job{};
        job.bits = c->d_bits; job.count = c->d_count; job.rank = b_rank.as<int32_t>(); job.len = allele_len ? b_len.as<double>() : nullptr;
        job.C = C; job.w64 = c->w64; job.a_pad = A; job.remove_low = remove_low ? 1 : 0;
        job.fast = g_em_fast > 0 ? 1 : 0;
        job.any_size = g_em_fast < 0 ? 1 : 0;
        std::vector<int32_t> order((size_t)n_alleles, -1);
        job.prob = prob_host; job.first = first_host; job.order = order.data(); job.n_out = n_alleles;
        { int rc_ = hgx_emx_run(&job, 1, st); if (rc_) return rc_; }
        if (job.status == 2) {
            hgx_set_error("EM: allele missing from the next estimate (the reference raises KeyError here, common:1365-1369)");
            return HGX_EKEY;
        }
        if (job.status == 0) {
            if (n_iter_host) *n_iter_host = job.n_iter;
            g_last_exact = job.fast ? 0 : 1;
            if (!job.fast) g_last_order.swap(order);
            return HGX_OK;
        }
        for (int a = 0; a < n_alleles; ++a) prob_host[a] = 0.0;
        if (first_host) for (int a = 0; a < n_alleles; ++a) first_host[a] = -1;
    }
#ifdef HGX_LAB
#include "lab/hgx_em_impl_mid.inc"        // round 2's mid-size reference-order EM (k_em_ref)
#endif
    // helper for the paths that do not carry the first-class information: look it up for the survivors afterwards
    [[maybe_unused]] auto first_for_present = [&]() -> int {
        if (!first_host) return HGX_OK;
        std::vector<int32_t> al;
        for (int a = 0; a < n_alleles; ++a) if (prob_host[a] >= 0.0) al.push_back(a);
        if (al.empty()) return HGX_OK;
        std::vector<int32_t> f(al.size());
        int rc_ = hgx_first_classes(cc, al.data(), (int32_t)al.size(), f.data(), stream);
        if (rc_) return rc_;
        for (size_t i = 0; i < al.size(); ++i) first_host[al[i]] = f[i];
        return HGX_OK;
    };
#ifdef HGX_LAB
#include "lab/hgx_em_impl_small.inc"        // round 1's one-workgroup EM (k_em_small)
#endif
    int rc = hgx_ensure_compact(c, st);
    if (rc) return rc;
    // from here on every vector lives in the compact allele space: element j is allele c->h_act[j]
    const int A_full = A;
    A = c->a1p;
    const int w64c = A / 64;
    DevBuf b_p, b_q1, b_q2, b_q3, b_wc, b_pr, b_pr1, b_pr2, b_pr3, b_len, b_scal, b_out;
    ALLOC(b_p, A * 8); ALLOC(b_q1, A * 8); ALLOC(b_q2, A * 8); ALLOC(b_q3, A * 8); ALLOC(b_out, A * 8);
    const size_t n_cnt_all = (size_t)std::max((C + BLOCK - 1) / BLOCK, (A + BLOCK - 1) / BLOCK);
    ALLOC(b_wc, (size_t)C * 8); ALLOC(b_pr, A); ALLOC(b_pr1, A); ALLOC(b_pr2, A); ALLOC(b_pr3, A); ALLOC(b_scal, S_N * 8 + n_cnt_all * 4);       // (scalars + the passes' slab counters: one block, one memset)
    double *d_len = nullptr;
    if (allele_len) {
        std::vector<double> l(A, 1.0);
        for (int j = 0; j < c->n_act; ++j) if (c->h_act[j] < n_alleles) l[j] = (double)allele_len[c->h_act[j]];
        ALLOC(b_len, A * 8);
        { int rc_ = hgx_h2d(b_len.p, l.data(), A * 8, st); if (rc_) return rc_; }
        { int rc_ = hgx_sync(st); if (rc_) return rc_; }
        d_len = b_len.as<double>();
    }
    double *p = b_p.as<double>(), *q1 = b_q1.as<double>(), *q2 = b_q2.as<double>(), *q3 = b_q3.as<double>();
    uint8_t *pr = b_pr.as<uint8_t>(), *pr1 = b_pr1.as<uint8_t>(), *pr2 = b_pr2.as<uint8_t>(), *pr3 = b_pr3.as<uint8_t>();
    double *wc = b_wc.as<double>(), *scal = b_scal.as<double>();
    HIPCHK(hipMemsetAsync(scal, 0, S_N * 8 + n_cnt_all * 4, st));
    // tie order of the result (first class containing each allele): independent of the EM, queued before it
    DevBuf b_fc;
    std::vector<int32_t> h_fc;
    if (first_host) {
        ALLOC(b_fc, (size_t)A * 4);
        h_fc.resize(A);
        hipLaunchKernelGGL(k_first_set_rows, dim3(nblk((long)A * 64, 256)), dim3(256), 0, st, c->d_bitsTC, A, c->c64, b_fc.as<int32_t>());
    }
    MatVec rows{c->d_bitsC, C, w64c, A};
    MatVec cols{c->d_bitsTC, A, c->c64, C};

    DevBuf b_part, b_part_c;
    if (g_backend == 0 || g_backend == 3) {
        // table-lookup kernels: word-transposed copies of both matrices, built once per class set
        const int Cp = c->c64 * 64;
        if (!c->d_wrow) {
            c->d_wrow = (uint64_t *)hgx_pool_alloc((size_t)w64c * Cp * 8);
            c->d_wcol = (uint64_t *)hgx_pool_alloc((size_t)c->c64 * A * 8);
            if (!c->d_wrow || !c->d_wcol) { hgx_set_error("device allocation failed"); return HGX_ENOMEM; }
            hipLaunchKernelGGL(k_word_transpose, dim3((w64c + 31) / 32, (Cp + 31) / 32), dim3(256), 0, st, c->d_bitsC, Cp, w64c, c->d_wrow);
            hipLaunchKernelGGL(k_word_transpose, dim3((c->c64 + 31) / 32, (A + 31) / 32), dim3(256), 0, st, c->d_bitsTC, A, c->c64, c->d_wcol);
        }
        const size_t n_part_r = (size_t)(w64c / 8) * Cp, n_part_c = (size_t)(c->c64 / 8) * A;
        ALLOC(b_part, std::max(n_part_r, n_part_c) * 8); ALLOC(b_part_c, n_part_c * 8);
        unsigned *const slab_counters = (unsigned *)(scal + S_N);                   // zeroed with the scalars above
        rows.M = c->d_wrow; rows.n_pad = Cp; rows.part = b_part.as<double>(); rows.counters = slab_counters;
        cols.M = c->d_wcol; cols.n_pad = A; cols.part = b_part_c.as<double>(); cols.counters = slab_counters;
        if (!rows.narrow && !HGX_LAB_SWITCH("em_persist") && !hgx_switch_has("em_skip", "defer")) {
            // the rows pass stops at its slab partials; the cols pass turns them into w_c in its prologue
            rows.defer_combine = 1;
            cols.src_part = rows.part; cols.src_slabs = w64c / 8; cols.src_pad = Cp; cols.src_count = c->d_count;
        }
    }
    // ---- resident-block path (k_em_grid): the block grid must be co-resident, one workgroup per CU ----------------
    // two resident grids that together need more CUs than the chip has could each hold a part and wait forever: grids reserve
    // their CUs from a process-wide budget and one that does not fit runs per pass instead
    static std::atomic<int> grid_cus{0};
    struct GridHold {
        int g = 0;
        void release() { if (g) grid_cus.fetch_sub(g); g = 0; }
        ~GridHold() { release(); }
    } grid_hold;
    GkArgs ga{};
    DevBuf b_gpr, b_gpc, b_gy, b_gfl, b_gst;
    bool grid = false;
    // opt-in (HGX_EM_GRID=1), or for small block grids only (HGX_EM_GRID_MAX workgroups: many small tasks in flight are bound by
    // the launch rate, and one launch replaces ~66)
    int grid_launches = 0;
    std::vector<Timed> timed;
    const int slot_rows = A <= 8 * BLOCK ? 0 : 1, slot_cols = C <= 8 * BLOCK ? 2 : 3;
    // timing: g_timing == 1 samples every 4th ungated rows pass of a call and the cols pass that follows it; g_timing == 2
    // times every plain rows / cols pass (bench.py turns it on for the last steps of the timed region only)
    int n_rows_seen = 0;
    bool stamping = false;
    auto stamp = [&](int slot, bool begin, int gate) {
        if (!g_timing) return;
        if (begin) {
            if (g_timing == 2) stamping = true;
            else if (slot == slot_rows) stamping = gate == 0 && (n_rows_seen++ % 4) == 0;
            else stamping = gate == 0;                       // called with gate 0 only right after a timed rows pass
            if (!stamping) return;
            Timed t;
            t.a = pool_event();
            t.b = pool_event();
            t.slot = slot;
            if (rows.M) { g_ev_start = t.a; g_ev_stop = t.b; }          // attached to the dispatch itself
            else (void)hipEventRecord(t.a, st);
            timed.push_back(t);
        } else if (stamping && !rows.M) (void)hipEventRecord(timed.back().b, st);
    };
    auto cols_pass = [&](const double *vec, const uint8_t *pres_v, int x_mode, double *q_out, uint8_t *pres_out, int gate,
                         bool after_timed_rows) -> int {
        stamp(slot_cols, true, (g_timing == 2 || after_timed_rows) ? 0 : 1);
        const int r = launch_matvec<MODE_COLS>(cols, st, wc, nullptr, x_mode == 2 ? 2 : 0, nullptr, vec, pres_v, d_len, q_out, pres_out, scal, gate);
        stamp(slot_cols, false, gate);
        return r;
    };
    // one application of the EM map: (vec, pres_v) -> (q_out, pres_out)
    auto next_prob = [&](const double *vec, const uint8_t *pres_v, int x_mode, double *q_out, uint8_t *pres_out, int gate) -> int {
        stamp(slot_rows, true, gate);
        int r = launch_matvec<MODE_ROWS>(rows, st, vec, pres_v, x_mode, c->d_count, nullptr, nullptr, nullptr, wc, nullptr, scal, gate);
        stamp(slot_rows, false, gate);
        if (r) return r;
        return cols_pass(vec, pres_v, x_mode, q_out, pres_out, gate, stamping);
    };
    // initial mass sum_c n_c / |S_c|, normalised (common:1299-1309); the resident-block kernel does it in its first launch
    if (!grid) {
        rc = next_prob(p, pr, 2, p, pr, 0);
        if (rc) return rc;
        hipLaunchKernelGGL(k_em_init_norm, dim3(1), dim3(BLOCK), 0, st, p, pr, A);
    }
    double h_scal[S_N];
    int h_abort = 0;
    std::vector<double> out(A);
    bool results_fetched = false;         // the speculative tail's results came back with its status word
    const int batch = 4;
    int launched_iters = 0;
    double tail_failed_at = 1e300;
    bool tail_done = false;
    const bool use_tail = !hgx_switch_has("em_skip", "tail");
    // ---- fused vector steps (k_lut_rows_fused): ping-pong estimate, extrapolated vector and state words -----------
    const bool fuse = (rows.defer_combine || rows.narrow) && A <= EPT * BLOCK && !hgx_switch_has("em_skip", "fuse");
    DevBuf b_palt, b_pralt, b_q2x, b_prx, b_scal2;
    double *p_alt = nullptr, *q2x = nullptr, *scal_alt = nullptr;
    uint8_t *pr_alt = nullptr, *prx = nullptr;
    if (fuse) {
        ALLOC(b_palt, A * 8); ALLOC(b_pralt, A); ALLOC(b_q2x, A * 8); ALLOC(b_prx, A); ALLOC(b_scal2, S_N * 8);
        p_alt = b_palt.as<double>(); pr_alt = b_pralt.as<uint8_t>(); q2x = b_q2x.as<double>(); prx = b_prx.as<uint8_t>();
        scal_alt = b_scal2.as<double>();
        HGX_ONCE_PER_DEVICE({
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lut_rows_fused<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)LUT_LDS));
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lut_rows_fused<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)LUT_LDS));
        });
    }
    auto rows_fused = [&](int fm, const double *qb, const uint8_t *prb, double *out_v, uint8_t *out_p) -> int {
        FuseArgs fz{q1, qb, pr1, prb, out_v, out_p, scal, scal_alt, remove_low ? 1 : 0};
        const dim3 grid(rows.n_words / 8, (rows.n_rows + BLOCK - 1) / BLOCK);
        if (fm == 0) hipLaunchKernelGGL(k_lut_rows_fused<0>, grid, dim3(BLOCK), LUT_LDS, st, rows.M, rows.n_pad, A, p, pr, fz, rows.part);
        else hipLaunchKernelGGL(k_lut_rows_fused<1>, grid, dim3(BLOCK), LUT_LDS, st, rows.M, rows.n_pad, A, p, pr, fz, rows.part);
        std::swap(scal, scal_alt);           // every later launch reads the state this one wrote
        return HGX_OK;
    };
    for (;;) {
        // with pruning, look at the survivor count right after the first pruning iteration (iteration index 10).
        // A first batch of 4 tells whether the EM is still far from converged (diff 10x above the stopping rule); if so the
        // remaining 7 iterations up to that point go out in one batch: every host sync is a ~40 us bubble in the chain.
        int nb = batch;
        if (use_tail && remove_low && launched_iters < 11) {
            nb = std::min(batch, 11 - launched_iters);
            if (launched_iters == batch && h_scal[S_DIFF] > 0.001) nb = 11 - launched_iters;
        }
        if (grid && use_tail && remove_low && launched_iters < 11) nb = 11 - launched_iters;    // no host sync inside: up to the first pruning
        launched_iters += nb;
        if (grid) {
            const int G = ga.R * ga.K;
            ga.p = p; ga.pr = pr; ga.scal = scal;
            ga.n_iters = nb;
            ga.do_init = grid_launches == 0 ? 1 : 0;
            ++grid_launches;
            HIPCHK(hipMemsetAsync(b_gfl.p, 0, ((size_t)3 * G * GK_FLAG_STRIDE + 32) * 4, st));
            Timed t{nullptr, nullptr, 4};
            if (g_timing) { t.a = pool_event(); t.b = pool_event(); timed.push_back(t); }
            HIPCHK(hipGetLastError());
        }
        // test switch em_graph: the batch's launches captured into a hipGraph and launched as one (measurement of what a graph
        // buys a chain of dependent 15 us kernels; NOTEBOOK.md section 10) -- capture, instantiation and launch are all inside the call
        const bool as_graph = !grid && !g_timing && HGX_LAB_SWITCH("em_graph") != nullptr;
        if (as_graph) HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int b = 0; b < (grid ? 0 : nb); ++b) {
            if (fuse) {
                // six launches per iteration: SQUAREM and the advance step ride in the prologue of the rows pass that follows
                if (b == 0) {
                    if ((rc = next_prob(p, pr, 0, q1, pr1, 0))) return rc;
                } else {
                    if ((rc = rows_fused(1, q3, pr3, p_alt, pr_alt))) return rc;      // advance of the previous iteration + rows(p)
                    std::swap(p, p_alt);
                    std::swap(pr, pr_alt);
                    if ((rc = cols_pass(p, pr, 0, q1, pr1, 0, false))) return rc;
                }
                if ((rc = next_prob(q1, pr1, 1, q2, pr2, 0))) return rc;
                if ((rc = rows_fused(0, q2, pr2, q2x, prx))) return rc;               // SQUAREM + rows(q2')
                if ((rc = cols_pass(q2x, prx, 0, q3, pr3, 1, false))) return rc;
                if (b == nb - 1)   // the host looks at the state after the batch: close it with the standalone step
                    hipLaunchKernelGGL(k_em_advance, dim3(1), dim3(BLOCK), 0, st, p, pr, q1, pr1, q3, pr3, A, remove_low ? 1 : 0, scal);
                continue;
            }
            if ((rc = next_prob(p, pr, 0, q1, pr1, 0))) return rc;        // Gene_prob_next  (p is used raw)
            if ((rc = next_prob(q1, pr1, 1, q2, pr2, 0))) return rc;      // Gene_prob_next2 (normalised on the fly)
            hipLaunchKernelGGL(k_em_squarem, dim3(1), dim3(BLOCK), 0, st, p, pr, q1, pr1, q2, pr2, A, scal);
            if ((rc = next_prob(q2, pr2, 0, q3, pr3, 1))) return rc;      // only if extrapolated
            hipLaunchKernelGGL(k_em_advance, dim3(1), dim3(BLOCK), 0, st, p, pr, q1, pr1, q3, pr3, A, remove_low ? 1 : 0, scal);
        }
        hipGraphExec_t gexec = nullptr;
        hipGraph_t gcap = nullptr;
        if (as_graph) {
            HIPCHK(hipStreamEndCapture(st, &gcap));
            HIPCHK(hipGraphInstantiate(&gexec, gcap, nullptr, nullptr, 0));
            HIPCHK(hipGraphLaunch(gexec, st));
        }
        struct GraphDrop { hipGraphExec_t &e; hipGraph_t &g; hipStream_t s; ~GraphDrop() { if (e) { (void)hipStreamSynchronize(s); (void)hipGraphExecDestroy(e); } if (g) (void)hipGraphDestroy(g); } } graph_drop{gexec, gcap, st};
        // Pruning has started (iteration index >= 10): the compact tail kernel goes out speculatively right behind the batch --
        // it checks the survivor count itself -- so that batch result and tail result come back in ONE host round trip.
        const bool spec_tail = use_tail && remove_low && launched_iters >= 11 && tail_failed_at == 1e300;
        if (spec_tail)
            hipLaunchKernelGGL(k_em_tail, dim3(1), dim3(BLOCK), 0, st, c->d_bitsTC, C, c->c64, A, c->d_count, p, pr, d_len,
                               remove_low ? 1 : 0, b_out.as<double>(), scal);
        if (spec_tail) {      // if the tail takes over, these ARE the results: fetch them in the same round trip
            if (first_host) { int rc_ = hgx_d2h(h_fc.data(), b_fc.p, (size_t)A * 4, st); if (rc_) return rc_; }
            { int rc_ = hgx_d2h(out.data(), b_out.p, A * 8, st); if (rc_) return rc_; }
        }
        { int rc_ = hgx_d2h(h_scal, scal, S_N * 8, st); if (rc_) return rc_; }
        if (grid) { int rc_ = hgx_d2h(&h_abort, ga.abort_flag, 4, st); if (rc_) return rc_; }
        { int rc_ = hgx_sync(st); if (rc_) return rc_; }
        if (grid && ga.stamps && grid_launches == 1) {
            // phase profile of the first launch: per workgroup, 11 stamps per application (see apply); printed as the mean over
            // applications 2.. of the time between consecutive stamps, for the first, a middle and the last workgroup, and the
            // maximum over all workgroups
            const int G = ga.R * ga.K;
            std::vector<unsigned long long> hs((size_t)G * GK_STAMPS);
            (void)hipMemcpy(hs.data(), ga.stamps, hs.size() * 8, hipMemcpyDeviceToHost);
            static const char *nm[11] = {"lut+rows", "publish A", "wait A", "sum A -> w", "cols (2 tables)", "publish B", "wait B",
                                         "reduce B -> q", "publish C", "wait C", "gather + vector step"};
            const int apps = (GK_STAMPS - 1) / 11;
            fprintf(stderr, "[k_em_grid] %d x %d workgroups; us per phase (wg 0 | wg %d | wg %d | max over wgs), applications 2..%d\n",
                    ga.R, ga.K, G / 2, G - 1, apps);
            double tot[4] = {0, 0, 0, 0};
            for (int ph = 0; ph < 11; ++ph) {
                double v[4] = {0, 0, 0, 0};
                for (int g = 0; g < G; ++g) {
                    double sum = 0;
                    int cnt = 0;
                    for (int ap = 1; ap < apps; ++ap) {
                        const unsigned long long t0 = hs[(size_t)g * GK_STAMPS + 1 + ap * 11 + ph];
                        const unsigned long long t1 = ph < 10 ? hs[(size_t)g * GK_STAMPS + 1 + ap * 11 + ph + 1]
                                                              : (ap + 1 < apps ? hs[(size_t)g * GK_STAMPS + 1 + (ap + 1) * 11] : 0);
                        if (!t0 || !t1) continue;
                        sum += (double)(t1 - t0) * 0.01;
                        ++cnt;
                    }
                    const double m = cnt ? sum / cnt : 0.0;
                    if (g == 0) v[0] = m;
                    if (g == G / 2) v[1] = m;
                    if (g == G - 1) v[2] = m;
                    v[3] = std::max(v[3], m);
                }
                for (int k = 0; k < 4; ++k) tot[k] += v[k];
                fprintf(stderr, "  %-22s %6.2f %6.2f %6.2f %6.2f\n", nm[ph], v[0], v[1], v[2], v[3]);
            }
            fprintf(stderr, "  %-22s %6.2f %6.2f %6.2f %6.2f\n", "application", tot[0], tot[1], tot[2], tot[3]);
            unsigned long long first = ~0ull, last_start = 0;
            for (int g = 0; g < G; ++g) { first = std::min(first, hs[(size_t)g * GK_STAMPS]); last_start = std::max(last_start, hs[(size_t)g * GK_STAMPS]); }
            fprintf(stderr, "  first workgroup -> last workgroup started: %.2f us\n", (double)(last_start - first) * 0.01);
        }
        if (grid && h_abort) {
            // a workgroup of the block grid never became resident (bounded spin): run this EM again, one launch per pass
            grid_hold.release();
            g_no_grid = true;
            const int r = em_impl(cc, n_alleles, remove_low, allele_len, prob_host, first_host, n_iter_host, stream);
            g_no_grid = false;
            return r;
        }
        if (spec_tail) {
            if (h_scal[S_TAIL] == 1.0) { tail_done = true; results_fetched = true; break; }
            if (h_scal[S_TAIL] == -1.0) tail_failed_at = h_scal[S_NPRES];
        }
        if (h_scal[S_DONE] != 0.0) break;
        if (!spec_tail && use_tail && h_scal[S_NPRES] <= 64.0 && h_scal[S_NPRES] < tail_failed_at) {
            // few survivors: finish on one wavefront (k_em_tail) unless too many distinct class masks remain
            hipLaunchKernelGGL(k_em_tail, dim3(1), dim3(BLOCK), 0, st, c->d_bitsTC, C, c->c64, A, c->d_count, p, pr, d_len,
                               remove_low ? 1 : 0, b_out.as<double>(), scal);
            const double rows_ran = h_scal[S_NROWS], cols_ran = h_scal[S_NCOLS];
            { int rc_ = hgx_d2h(h_scal, scal, S_N * 8, st); if (rc_) return rc_; }
            { int rc_ = hgx_sync(st); if (rc_) return rc_; }
            h_scal[S_NROWS] = rows_ran; h_scal[S_NCOLS] = cols_ran;
            if (h_scal[S_TAIL] == 1.0) { tail_done = true; break; }
            tail_failed_at = h_scal[S_NPRES];
        }
    }
    if (g_timing) {
        const int64_t rows_bytes = (int64_t)C * w64c * 8 + (int64_t)A * 9 + (int64_t)C * 16;
        const int64_t cols_bytes = (int64_t)A * c->c64 * 8 + (int64_t)C * 8 + (int64_t)A * 26;
        for (auto &t : timed) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, t.a, t.b);
            g_stats[t.slot].ms += ms;
            g_stats[t.slot].launches += 1;
            g_event_pool.push_back(t.a);
            g_event_pool.push_back(t.b);
        }
        // bytes describe the TIMED launches (all ungated: they did the full pass); executed = every pass of the call
        if (grid) {       // SURVEY 8d's bytes of one application (both mat-vecs) x the applications the launches ran
            g_stats[4].executed += (int64_t)h_scal[S_NROWS];
            g_stats[4].bytes += (int64_t)h_scal[S_NROWS] * (rows_bytes + cols_bytes);
        } else {
            for (auto &t : timed) g_stats[t.slot].bytes += t.slot == slot_rows ? rows_bytes : cols_bytes;
            g_stats[slot_rows].executed += (int64_t)h_scal[S_NROWS];
            g_stats[slot_cols].executed += (int64_t)h_scal[S_NCOLS];
        }
    }
    if (h_scal[S_KEYERR] != 0.0) {
        hgx_set_error("EM: allele missing from the next estimate (the reference raises KeyError here, common:1365-1369)");
        return HGX_EKEY;
    }
    if (!tail_done) hipLaunchKernelGGL(k_em_finish, dim3(1), dim3(BLOCK), 0, st, p, pr, d_len, A, remove_low ? 1 : 0, b_out.as<double>());
    HIPCHK(hipGetLastError());
    if (!results_fetched) {
        if (first_host) { int rc_ = hgx_d2h(h_fc.data(), b_fc.p, (size_t)A * 4, st); if (rc_) return rc_; }
        { int rc_ = hgx_d2h(out.data(), b_out.p, A * 8, st); if (rc_) return rc_; }
        { int rc_ = hgx_sync(st); if (rc_) return rc_; }
    }
    for (int a = 0; a < n_alleles; ++a) prob_host[a] = -1.0;
    for (int j = 0; j < c->n_act; ++j) if (c->h_act[j] < n_alleles) prob_host[c->h_act[j]] = out[j];
    if (first_host) for (int j = 0; j < c->n_act; ++j) if (c->h_act[j] < n_alleles) first_host[c->h_act[j]] = h_fc[j];
    (void)A_full;
    if (n_iter_host) *n_iter_host = (int)h_scal[S_ITER];
    return HGX_OK;
}

// Gene_cmpt2 + EM #2 (core:1752-1782) without materialising the filtered class set when <= 64 alleles pass the filter
extern "C" int hgx_em_masked(const hgx_classes *cc, const uint64_t *mask_host, int32_t n_alleles, int32_t remove_low,
                             const int32_t *allele_len, double *prob_host, int32_t *first_class_host, int32_t *n_iter_host,
                             int32_t *n_classes_host, void *stream) {
    ARGCHK(cc && mask_host && prob_host && first_class_host && n_alleles > 0 && n_alleles <= cc->a_pad);
    g_last_exact = 0;
    hipStream_t st = (hipStream_t)stream;
    hgx_classes_order_after(cc, st);
    for (int a = 0; a < n_alleles; ++a) { prob_host[a] = -1.0; first_class_host[a] = -1; }
    if (n_iter_host) *n_iter_host = 1;          // (an empty Gene_cmpt2 still costs the reference one pass of its loop: see em_impl)
    if (n_classes_host) *n_classes_host = 0;
    const int C = cc->n_classes;
    if (C == 0) return HGX_OK;
    std::vector<int32_t> al;
    for (int w = 0; w < cc->w64; ++w)
        for (uint64_t m = mask_host[w]; m; m &= m - 1) {
            const int a = 64 * w + __builtin_ctzll(m);
            if (a < n_alleles) al.push_back(a);
        }
    if (al.empty()) return HGX_OK;
    if (al.size() <= 64 && !hgx_switch_has("em_skip", "masked")) {
        const int A1 = (int)al.size();
        // with the alleles' name order at hand the kept alleles go up in that order (= their order inside a class key), and the
        // single-wavefront EM follows the reference's summation order exactly
        const bool exact = cc->h_rank != nullptr && !hgx_switch_has("em_skip", "exact");
        if (exact) std::sort(al.begin(), al.end(), [&](int32_t x, int32_t y) { return cc->h_rank[x] < cc->h_rank[y]; });
        // one staging struct each way: [al | len] up, [scal | ticket | out | first] down (one copy + one memset + one copy)
        struct Up { int32_t al[64]; double len[64]; } up;
        struct Down { double scal[S_N]; unsigned ticket[2]; double out[64]; int32_t first[64]; } down;
        for (int j = 0; j < 64; ++j) { up.al[j] = j < A1 ? al[j] : 0; up.len[j] = (allele_len && j < A1) ? (double)allele_len[al[j]] : 1.0; }
        DevBuf b_up, b_down, b_pub;
        ALLOC(b_up, sizeof(Up)); ALLOC(b_down, sizeof(Down));
        { int rc_ = hgx_h2d(b_up.p, &up, sizeof(Up), st); if (rc_) return rc_; }
        HIPCHK(hipMemsetAsync(b_down.p, 0, offsetof(Down, out), st));
        const int nb = std::max(1, std::min(64, (C + 511) / 512));
        ALLOC(b_pub, (size_t)nb * 64 * sizeof(MaskedEntry));
        Up *d_up = (Up *)b_up.p;
        Down *d_down = (Down *)b_down.p;
        hipLaunchKernelGGL(k_em_masked, dim3(nb), dim3(BLOCK), 0, st, cc->d_bits, C, cc->w64, cc->d_count, d_up->al, A1,
                           allele_len ? d_up->len : nullptr, remove_low ? 1 : 0, d_down->out, d_down->first, d_down->scal,
                           (MaskedEntry *)b_pub.p, d_down->ticket, exact ? 1 : 0);
        HIPCHK(hipGetLastError());
        { int rc_ = hgx_d2h(&down, b_down.p, sizeof(Down), st); if (rc_) return rc_; }
        { int rc_ = hgx_sync(st); if (rc_) return rc_; }
        const double *h_scal = down.scal, *out = down.out;
        const int32_t *first = down.first;
        if (h_scal[S_FALLBACK] == 0.0) {
            if (h_scal[S_KEYERR] != 0.0) {
                hgx_set_error("EM: allele missing from the next estimate (the reference raises KeyError here, common:1365-1369)");
                return HGX_EKEY;
            }
            for (int j = 0; j < A1; ++j) {
                prob_host[al[j]] = out[j];
                first_class_host[al[j]] = out[j] >= 0.0 ? first[j] : -1;
            }
            if (n_iter_host) *n_iter_host = (int)h_scal[S_ITER];
            if (n_classes_host) *n_classes_host = (int)h_scal[S_NCLS];
            g_last_exact = exact ? 1 : 0;
            return HGX_OK;
        }
    }
    // general path: materialise the filtered, merged class set and run the ordinary EM on it
    DevBuf b_mask;
    ALLOC(b_mask, (size_t)cc->w64 * 8);
    { int rc_ = hgx_h2d(b_mask.p, mask_host, (size_t)cc->w64 * 8, st); if (rc_) return rc_; }
    hgx_classes *sub = nullptr;
    int rc = hgx_dedup_classes(&sub, cc->d_bits, nullptr, cc->d_count, C, cc->a_pad, b_mask.as<uint64_t>(), stream);
    if (rc == HGX_OK) {
        if (n_classes_host) *n_classes_host = sub->n_classes;
        // the filtered set inherits the alleles' name order: a hand-off with more than 64 alleles then still runs in the
        // reference's order when it fits k_em_ref (the fuzz found an iteration count off by one exactly here)
        if (cc->h_rank && !sub->h_rank) {
            sub->h_rank = new int32_t[sub->a_pad];
            for (int a = 0; a < sub->a_pad; ++a) sub->h_rank[a] = a < cc->a_pad ? cc->h_rank[a] : 0x7fffffff;
        }
        rc = hgx_em_ordered(sub, n_alleles, remove_low, allele_len, prob_host, first_class_host, n_iter_host, stream);
    }
    hgx_classes_destroy(sub);
    return rc;
}

// Name order of the alleles (rank_host[a] = position of allele a's name in sorted order = its place inside a class key,
// typing_core.py:1229): lets the small EMs sum in the reference's order.  Copied; NULL clears it.
extern "C" int hgx_classes_set_allele_rank(hgx_classes *c, const int32_t *rank_host, int32_t n) {
    ARGCHK(c && n >= 0 && n <= c->a_pad);
    delete[] c->h_rank;
    c->h_rank = nullptr;
    if (!rank_host) return HGX_OK;
    c->h_rank = new int32_t[c->a_pad];
    for (int a = 0; a < c->a_pad; ++a) c->h_rank[a] = a < n ? rank_host[a] : 0x7fffffff;
    return HGX_OK;
}

// Test aid: one rows pass (s_c -> w_c = n_c / s_c) or cols pass (t_a) of the EM map with a chosen backend.
//   which = 0: y[c] = count[c] / sum_a B[c][a] x[a]   (x: a_pad doubles)      -> n_classes doubles
//   which = 1: y[a] = sum_c B[c][a] x[c]              (x: n_classes doubles)  -> a_pad doubles
extern "C" int hgx_debug_matvec(const hgx_classes *cc, int which, int backend, const double *x_host, double *y_host) {
#ifndef HGX_LAB
    (void)cc; (void)which; (void)backend; (void)x_host; (void)y_host;
    hgx_set_error("hgx_debug_matvec drives the lab back-ends of the bit mat-vec: build libhgx_lab.so (hisat-genotype_amd/build.py build_lab)");
    return HGX_EINVAL;
#else
#include "lab/hgx_em_debug_matvec.inc"
#endif
}
