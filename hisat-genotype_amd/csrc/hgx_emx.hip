// hgx_emx.hip -- 8a-8 for MANY problems at once: single_abundance (typing_common.py:1282-1410) in the REFERENCE'S OWN ORDER of
// floating-point operations, one workgroup per problem, any number of problems per launch (gfx950).
//
// Contract (as ref_em_run / k_em_ref in hgx_em.hip, which this generalises to thousands of classes over thousands of alleles
// and to a task dimension): every sum is sequential in the order typing_common.py walks its dicts -- alleles of a class in key
// (name) order, classes in dict order, dict values in insertion order --, every term is formed as the reference forms it
// (`float(count) * Gene_prob[allele] / alleles_prob`), nothing is contracted.  The abundances, the pruning decisions
// (`prob >= max / 10`) and the stopping decision (`diff > 0.0001`) are therefore the reference's, bit for bit; follows
// oracle/hgx_oracle.c orc_single_abundance line by line.
//
// How the order is kept while 1024 threads work:
//   rows   alleles_prob of 64 classes at a time: lane = class, the walk over the alleles j (name order) is the loop; the 64-bit
//          word "which of my 64 classes contain allele j" and p_j are wave-uniform and arrive through the scalar cache
//          (s_load_dwordx16: eight j per load); the add is acc = fma(b, p_j, acc) with b = 1.0 / 0.0 picked per lane by the mask
//          word (ONE v_cndmask for the high half of b; fma(1, x, acc) == acc + x and fma(0, x, acc) == acc exactly, x finite)
//   cols   next[a] of 64 alleles at a time: lane = allele, the walk over the classes c (dict order) is the loop; count_c,
//          alleles_prob_c and a refined reciprocal are wave-uniform scalars; the quotient (count * prob) / alleles_prob is formed
//          for all lanes with the last three instructions of the compiler's own correctly rounded division (q0 = x * r,
//          e = fma(-s, q0, x), q = fma(e, r, q0): the part that depends on the numerator; the reciprocal refinement, which only
//          depends on alleles_prob, is done once per class) whenever neither operand is so large or small that the division
//          would rescale -- otherwise by the plain `/` --, and added under the mask as above
//   sums over a dict (normalisation totals, the two SQUAREM sums, prob_diff): the operands are first stored in insertion
//          order, then ONE wavefront adds them eight per scalar load: that is what `sum(d.values())` does, no faster order
//          gives the same bits
//   the insertion order of a dict = (first WALKED class containing the allele, key order): re-derived (bitonic sort in LDS)
//          only when the membership or the set of walked classes changed.
// The class matrix is re-laid once per problem in the compact name-ordered allele space, in the two word orders the passes
// stream: Mk[allele tile][class] and Mr[class tile][allele].
//
// Three instantiations: k_emx<false> (this contract), k_emx<true> (hgx_type_opts.em_fast: the same skeleton with table-lookup
// mat-vecs and tree reductions, within ~1e-11 of the reference) and k_emx<false, true> (cluster mode: ONE large problem on several
// workgroups, same contract, same bits -- see the comment at the kernel).
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "hgx_common.hpp"
#include "hgx_emx.hpp"

namespace {

constexpr int XB = 1024, XNW = XB / 64;
constexpr int XA = HGX_EMX_MAX_ALLELES, XC = HGX_EMX_HARD_MAX_CLASSES;
constexpr int XAW = XA / 64, XCW = XC / 64;

enum { XS_ITER = 0, XS_STATUS = 1, XS_A1 = 2, XS_ORDERS = 3, XS_RES_OFF = 4, XS_RES_N = 5, XS_NCLS = 6, XS_APPS = 7, XS_N = 8 };

typedef hgx_emx_rec EmxRes;                                  // one allele of a returned dict

struct EmxTask {
    const uint64_t *B;
    const int64_t *count;
    const int32_t *rank;
    const double *len;
    const uint64_t *mask;   // [w64] or NULL: the hand-off (core:1752-1766) -- rows AND mask, empty ones dropped, equal ones merged
    int32_t C, w64, a_pad, remove_low;
    int32_t c_alloc;        // classes the class-indexed scratch is sized for (= C; the hand-off: the merged classes it may produce)
    int32_t fast;           // 1: table-lookup arithmetic (any summation order; within rounding of the reference), see fast_* below
    // scratch
    uint64_t *Rm;       // [Cp][w64]   set-up only: the class rows in the compact name-ordered allele space, class-major
    uint64_t *Mk;       // [A1w][Cp]   word (aw, c): which alleles of tile aw are in class c
    uint64_t *Mr;       // [Cw][A1s]   word (cw, j): which classes of tile cw contain allele j
    double *dv;         // [3][A1s]    dict values
    uint16_t *pos;      // [4][A1s]    insertion position of compact allele j in order buffer k (0xFFFF: not in it)
    double *tmpv;       // [A1s]
    double *vlen;       // [A1s]
    double *cls;        // [5][Cp]     count, count / |class|, and per application: count (0 if skipped), alleles_prob, reciprocal
    uint8_t *din;       // [3][A1s]    dict membership
    int32_t *sorted;    // [A1s]       allele index of compact allele j
    int32_t *first_c;   // [A1s]       first class (dict order) containing compact allele j
    double *scal;       // [XS_N]
    EmxRes *res;        // result records of ALL jobs of the launch: a job reserves its run with one atomic add on *cursor
    unsigned long long *cursor;
    unsigned long long *stamps;   // [16] or NULL (HGX_EMX_STAMPS=1)
    // cluster mode (k_emx<false, true>: ONE problem on gridDim.x workgroups): [0] barrier count, [1] abort flag, [2] command,
    // [3] "some class needs the plain division" -- and the words the tile loops hand to the leader
    int32_t cluster;              // > 0: this job runs as k_emx<false, true> on that many workgroups (the ordinary launch skips it)
    int32_t cl_first;             // ... blocks [cl_first, cl_first + cluster) of the cluster launch (several clusters side by side)
    int32_t cl_spins;             // polls a cluster barrier waits before the cluster gives up (test switch emx_cluster_spins)
    unsigned int *cl_ctl;
    unsigned long long *gvalid;   // [XCW]  as XLds::validw, written by the workgroup that walked the class tile
    unsigned long long *gin;      // [XAW]  as XLds::in_now
};

typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));

// Wave-uniform 64-byte loads through the scalar cache.  Issue and wait sit in ONE asm statement: the compiler never sees a
// register that is still being written (with the pressure of this kernel it spills scalar registers, and a spill between an
// issued load and its wait would save the old contents).  The latency is covered by the other wavefronts of the SIMD.
__device__ __forceinline__ void sload2(const void *pa, const void *pb, u32x16 &a, u32x16 &b) {
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(a), "=&s"(b) : "s"(pa), "s"(pb) : "memory");
}
__device__ __forceinline__ void sload4(const void *pa, const void *pb, const void *pc, const void *pd, u32x16 &a, u32x16 &b,
                                       u32x16 &c, u32x16 &d) {
    asm volatile("s_load_dwordx16 %0, %4, 0x0\n\ts_load_dwordx16 %1, %5, 0x0\n\ts_load_dwordx16 %2, %6, 0x0\n\t"
                 "s_load_dwordx16 %3, %7, 0x0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(a), "=&s"(b), "=&s"(c), "=&s"(d) : "s"(pa), "s"(pb), "s"(pc), "s"(pd) : "memory");
}
__device__ __forceinline__ double dbl_of(const u32x16 &v, int k) {
    return __hiloint2double((int)v[2 * k + 1], (int)v[2 * k]);
}
__device__ __forceinline__ uint64_t u64_of(const u32x16 &v, int k) { return ((uint64_t)v[2 * k + 1] << 32) | v[2 * k]; }

// Everything this workgroup stored to global memory becomes readable through the scalar cache (and by its other waves):
// stores drained, workgroup barrier, scalar cache invalidated (it may hold the previous contents of the same lines).
__device__ __forceinline__ void phase_sync() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __builtin_amdgcn_s_dcache_inv();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// agent-scope 8- / 4-byte accesses (global_load / global_store ... sc1): what a table-lookup cluster hands over goes through these on
// both sides, so its hand-overs need no cache write-back or invalidate (MI355X_MICROARCH.md, "8-B agent atomics both sides")
typedef __attribute__((address_space(1))) double gdouble_t;
typedef __attribute__((address_space(1))) unsigned int guint_t;
__device__ __forceinline__ double ld_agent(const double *p) { return __hip_atomic_load((const gdouble_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(double *p, double v) { __hip_atomic_store((gdouble_t *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned int ld_agent(const unsigned int *p) { return __hip_atomic_load((const guint_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(unsigned int *p, unsigned int v) { __hip_atomic_store((guint_t *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct XLds {
    double tmpo[2][XA];                        // operands of a sequential sum in insertion order ([1] doubles as sort scratch and,
                                               // during the set-up, as the name-ordered allele list; [0] as row buffers there)
    unsigned long long validw[XCW], sig_valid[XCW];
    unsigned long long in_now[XAW], sig_in[XAW];
    unsigned long long orw[128], rbm[128];
    int rpre[128];
    double red[XNW];
    double bc[4];
    int npos[4];
    int cache_ord, need_slow, A1, res_base, cl_ok;
    double xs[512];                            // (fast mode) the slab of the vector a lookup table is built from
};

static_assert(sizeof(XLds) + 256 <= 160 * 1024, "k_emx: one workgroup per CU, all of its LDS");

#pragma clang fp contract(off)

// acc + (bit of my lane in `mask` ? x : 0), as ONE select and ONE fma: b = 1.0 or 0.0 (their low halves are both zero),
// fma(b, x, acc) rounds x + acc once (b = 1) or returns acc (b = 0, x finite).
__device__ __forceinline__ double sel_add(double acc, uint64_t mask, double x) {
    const int hi = __builtin_amdgcn_inverse_ballot_w64(mask) ? 0x3FF00000 : 0;
    return __builtin_fma(__hiloint2double(hi, 0), x, acc);
}

// sum of arr[0..n8) (LDS; n8 a multiple of 8, the tail padded with +0.0) added one by one in index order by the calling
// wavefront (every lane reads the same address: a broadcast, and every lane returns the sum)
// (round 6: the next group of eight is requested -- unconditionally, so that the loads stay wave-uniform scalar loads -- before the eight
// dependent adds of the current one: the sum of 4 549 values 44 -> 33 us; same adds in the same order)
__device__ __forceinline__ double seq_sum(const double *arr, int n8) {
    double t = 0.0;
    if (n8 <= 0) return t;
    double x[8], y[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = arr[k];
    int r0 = 8;
    for (; r0 + 8 <= n8; r0 += 16) {                    // two groups per trip: the next group's (unconditional) loads before this group's adds
#pragma unroll
        for (int k = 0; k < 8; ++k) y[k] = arr[r0 + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) t = ((t) + (x[k]));
        if (r0 + 16 <= n8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = arr[r0 + 8 + k];
#pragma unroll
            for (int k = 0; k < 8; ++k) t = ((t) + (y[k]));
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) t = ((t) + (y[k]));
            return t;
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) t = ((t) + (x[k]));
    return t;
}
__device__ __forceinline__ void seq_sum2(const double *a, const double *b, int n8, double &ta, double &tb) {
    ta = 0.0; tb = 0.0;
    if (n8 <= 0) return;
    double x[8], y[8], u[8], v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { x[k] = a[k]; y[k] = b[k]; }
    int r0 = 8;
    for (; r0 + 8 <= n8; r0 += 16) {                    // as seq_sum: the next group's loads before this group's adds
#pragma unroll
        for (int k = 0; k < 8; ++k) { u[k] = a[r0 + k]; v[k] = b[r0 + k]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) { ta = ((ta) + (x[k])); tb = ((tb) + (y[k])); }
        if (r0 + 16 <= n8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { x[k] = a[r0 + 8 + k]; y[k] = b[r0 + 8 + k]; }
#pragma unroll
            for (int k = 0; k < 8; ++k) { ta = ((ta) + (u[k])); tb = ((tb) + (v[k])); }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) { ta = ((ta) + (u[k])); tb = ((tb) + (v[k])); }
            return;
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { ta = ((ta) + (x[k])); tb = ((tb) + (y[k])); }
}

#pragma clang fp contract(fast)
// ---- fast mode: "four Russians" over a 0/1 matrix (as k_lutmatvec, hgx_em.hip, on ONE workgroup) ---------------------------------
// 256 subset sums of each of the 64 groups of 8 consecutive elements of xs[512]: one lookup then stands for 8 matrix bits
__device__ __forceinline__ void xlut_build(const double *xs, double *Tb, int tid) {
    const int g = tid >> 4, lo = tid & 15;
    const double x0 = xs[8 * g], x1 = xs[8 * g + 1], x2 = xs[8 * g + 2], x3 = xs[8 * g + 3];
    const double x4 = xs[8 * g + 4], x5 = xs[8 * g + 5], x6 = xs[8 * g + 6], x7 = xs[8 * g + 7];
    const double L = (((lo & 1 ? x0 : 0.0) + (lo & 2 ? x1 : 0.0)) + (lo & 4 ? x2 : 0.0)) + (lo & 8 ? x3 : 0.0);
    double *Tg = Tb + g * 256 + lo;
#pragma unroll
    for (int hi = 0; hi < 16; ++hi) {
        const double H = (((hi & 1 ? x4 : 0.0) + (hi & 2 ? x5 : 0.0)) + (hi & 4 ? x6 : 0.0)) + (hi & 8 ? x7 : 0.0);
        Tg[hi * 16] = L + H;
    }
}
__device__ __forceinline__ double xlut_row(const double *Tb, const uint64_t (&w)[8]) {
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (__ballot(w[i] != 0ull) == 0ull) continue;
        const uint32_t wl = (uint32_t)w[i], wh = (uint32_t)(w[i] >> 32);
        const double *Ti = Tb + i * 8 * 256;
        acc += Ti[0 * 256 + (wl & 255u)];
        acc += Ti[1 * 256 + ((wl >> 8) & 255u)];
        acc += Ti[2 * 256 + ((wl >> 16) & 255u)];
        acc += Ti[3 * 256 + (wl >> 24)];
        acc += Ti[4 * 256 + (wh & 255u)];
        acc += Ti[5 * 256 + ((wh >> 8) & 255u)];
        acc += Ti[6 * 256 + ((wh >> 16) & 255u)];
        acc += Ti[7 * 256 + (wh >> 24)];
        asm volatile("" : "+v"(acc));                      // (eight lookups in flight at a time, not sixty-four)
    }
    return acc;
}

#pragma clang fp contract(off)
// CL (cluster mode, reference-order arithmetic only): ONE problem on gridDim.x workgroups -- for a problem far beyond the default gate
// (16 000 classes: hgx_emx_job::any_size) the tile loops (class rows, Mk / Mr, initial estimate, rows, cols) are shared out over
// the workgroups, everything per allele or per dict stays with workgroup 0 (the leader); the others serve commands.  Hand-overs
// are agent-scope release / acquire fences around a counting barrier in global memory (bounded spins: a cluster that is not
// co-resident in time gives up and the caller runs the problem on one workgroup).  Same sums in the same orders: same bits.
template <bool FAST, bool CL = false>
__global__ __launch_bounds__(XB) void k_emx(const EmxTask *__restrict__ tasks, int n_tasks) {
    extern __shared__ double xlds_raw[];
    XLds &S = *reinterpret_cast<XLds *>(xlds_raw);
    int t_idx = CL ? 0 : (int)blockIdx.x;
    if constexpr (CL && FAST) {                           // every task of the launch has a block range, in task order: bisection
        int lo = 0, hi = n_tasks - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tasks[mid].cl_first <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
        t_idx = lo;
    } else if constexpr (CL) {                            // several clusters side by side: mine is the one whose block range holds me
        for (int k = 0; k < n_tasks; ++k)
            if (tasks[k].cluster > 0 && (int)blockIdx.x >= tasks[k].cl_first && (int)blockIdx.x < tasks[k].cl_first + tasks[k].cluster) t_idx = k;
    }
    const EmxTask T = tasks[t_idx];
    if ((T.fast != 0) != FAST) return;                    // (the launch of the other arithmetic takes this job)
    if ((T.cluster > 0) != CL) return;                    // (... and the cluster launch a cluster job)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = CL ? (int)blockIdx.x - T.cl_first : 0, n_wg = CL ? T.cluster : 1;       // tile loops: start wave * n_wg + wg, stride XNW * n_wg
    int cl_phase = 0;
    bool cl_dead = false;
    auto cluster_sync = [&]() {
        if constexpr (CL) {
            if (n_wg == 1) return;                         // (a table-lookup launch seats most problems on one workgroup)
            cl_phase += 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                bool ok = !cl_dead;
                if (ok) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    __hip_atomic_fetch_add(&T.cl_ctl[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned target = (unsigned)n_wg * (unsigned)cl_phase;
                    long spins = 0;
                    while (__hip_atomic_load(&T.cl_ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                        if (__hip_atomic_load(&T.cl_ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = false; break; }
                        if (++spins > (long)T.cl_spins) { __hip_atomic_store(&T.cl_ctl[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = false; break; }
                        __builtin_amdgcn_s_sleep(8);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
                S.cl_ok = ok ? 1 : 0;
            }
            __syncthreads();
            if (!S.cl_ok) cl_dead = true;
            __builtin_amdgcn_s_dcache_inv();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
        }
    };
    // hand-over of a table-lookup cluster: the bytes travel as agent-scope stores and loads (st_agent / ld_agent), so the barrier is
    // only the counter -- stores drained, one lane adds and polls, the workgroup's barrier releases the other wavefronts
    auto cluster_sync_light = [&]() {
        if constexpr (CL) {
            if (n_wg == 1) return;
            cl_phase += 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                bool ok = !cl_dead;
                if (ok) {
                    __hip_atomic_fetch_add(&T.cl_ctl[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned target = (unsigned)n_wg * (unsigned)cl_phase;
                    long spins = 0;
                    while (__hip_atomic_load(&T.cl_ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                        if (__hip_atomic_load(&T.cl_ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = false; break; }
                        if (++spins > (long)T.cl_spins) { __hip_atomic_store(&T.cl_ctl[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = false; break; }
                        __builtin_amdgcn_s_sleep(2);
                    }
                }
                S.cl_ok = ok ? 1 : 0;
            }
            __syncthreads();
            if (!S.cl_ok) cl_dead = true;
        }
    };
    const int w64 = T.w64, A1s = T.a_pad;
    const int CpA = (T.c_alloc + 63) & ~63;             // stride of the class-indexed scratch arrays
    int C = T.C;                                        // (the hand-off mode continues with the merged classes)
    unsigned long long t_mark = T.stamps ? wall_clock64() : 0, acc_t[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto lap = [&](int k) { if (T.stamps) { const unsigned long long t = wall_clock64(); acc_t[k] += t - t_mark; t_mark = t; } };
    auto give_up = [&](double status) { if (tid == 0) { T.scal[XS_STATUS] = status; T.scal[XS_ITER] = 0.0; T.scal[XS_RES_N] = 0.0; } };
    if (C <= 0 || (!T.mask && C > XC) || w64 > 128 || A1s > XA) { give_up(1.0); return; }
    // ---- which alleles occur at all, and their place in name order ----------------------------------------------------
    for (int w = tid; w < 128; w += XB) { S.orw[w] = 0ull; S.rbm[w] = 0ull; }
    __syncthreads();
    {
        const int w = tid & 127, slice = tid >> 7;
        unsigned long long acc = 0ull;
        if (w < w64)
            for (int c = slice; c < C; c += 8) acc |= T.B[(size_t)c * w64 + w];
        if (acc && T.mask) acc &= T.mask[w];
        if (acc) atomicOr(&S.orw[w], acc);
    }
    __syncthreads();
    for (int a = tid; a < 64 * w64; a += XB)
        if ((S.orw[a >> 6] >> (a & 63)) & 1ull) {
            const int r = T.rank[a];
            atomicOr(&S.rbm[(r >> 6) & 127], 1ull << (r & 63));
        }
    __syncthreads();
    if (tid == 0) {
        int t = 0;
        for (int w = 0; w < 128; ++w) { S.rpre[w] = t; t += __popcll(S.rbm[w]); }
        S.A1 = t;
    }
    __syncthreads();
    const int A1 = S.A1;
    lap(7);
    if (A1 > XA || (T.mask && A1 > 64)) { give_up(1.0); return; }
    if (A1 <= 0) {                                      // (hand-off: no class left) an empty result -- after ONE pass of the reference's
        // loop: `diff` starts at 1.0, the pass over an empty dict leaves it at 0 (typing_common.py:1351-1404; found by tools/fuzz_many.py)
        if (tid == 0) { T.scal[XS_STATUS] = 0.0; T.scal[XS_ITER] = 1.0; T.scal[XS_RES_N] = 0.0; T.scal[XS_NCLS] = 0.0; T.scal[XS_A1] = 0.0; }
        return;
    }
    const int A1w = (A1 + 63) >> 6;
    int *srt = reinterpret_cast<int *>(&S.tmpo[1][0]);  // (free until the first order derivation)
    for (int a = tid; a < 64 * w64; a += XB)
        if ((S.orw[a >> 6] >> (a & 63)) & 1ull) {
            const int r = T.rank[a], w = (r >> 6) & 127;
            const int j = S.rpre[w] + __popcll(S.rbm[w] & ((1ull << (r & 63)) - 1ull));
            srt[j] = a;
            T.sorted[j] = a;
        }
    __syncthreads();
    // ---- Mk: one wavefront per class picks, for every compact allele, its bit out of the row (held in LDS) ------------------
    double *cnt_c = T.cls, *t0_c = T.cls + CpA, *n_c = T.cls + 2 * (size_t)CpA, *s_c = T.cls + 3 * (size_t)CpA, *r_c = T.cls + 4 * (size_t)CpA;
    unsigned long long *rowbuf = reinterpret_cast<unsigned long long *>(&S.tmpo[0][0]) + (size_t)wave * 512;      // 4 rows of 128 words
    if (!T.mask) {
        // four classes per pass: the compact allele list (srt, 4 bytes per allele) is read once for the four, and a row is
        // read as 32-bit words -- the phase is bound by these LDS reads (12 bytes per class and allele before, 5 now)
        const uint32_t *row32 = reinterpret_cast<const uint32_t *>(rowbuf);
        for (int c0 = 4 * (wave * n_wg + wg); c0 < CpA; c0 += 4 * XNW * n_wg) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + u;
                rowbuf[128 * u + lane] = (c < C && lane < w64) ? T.B[(size_t)c * w64 + lane] : 0ull;
                rowbuf[128 * u + 64 + lane] = (c < C && lane + 64 < w64) ? T.B[(size_t)c * w64 + 64 + lane] : 0ull;
            }
            __builtin_amdgcn_wave_barrier();
            int size[4] = {0, 0, 0, 0};
            unsigned long long keep0[4] = {0ull, 0ull, 0ull, 0ull}, keep1[4] = {0ull, 0ull, 0ull, 0ull};
            for (int aw0 = 0; aw0 < A1w; aw0 += 2) {
                int g[2];
#pragma unroll
                for (int v = 0; v < 2; ++v) { const int j = 64 * (aw0 + v) + lane; g[v] = j < A1 ? srt[j] : -1; }
                uint32_t word[4][2];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < 2; ++v) word[u][v] = row32[256 * u + ((g[v] < 0 ? 0 : g[v]) >> 5)];
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    const int aw = aw0 + v;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const bool bit = g[v] >= 0 && ((word[u][v] >> (g[v] & 31)) & 1u);
                        const unsigned long long m = __ballot(bit);
                        size[u] += __popcll(m);
                        if (lane == (aw & 63)) { if (aw < 64) keep0[u] = m; else keep1[u] = m; }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            // (class-major first: one contiguous run of words per class; a store per word straight into Mk[aw][c] would be
            // A1w scattered 8-byte writes per class)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + u;
                if (lane < A1w) T.Rm[(size_t)c * w64 + lane] = keep0[u];
                if (lane + 64 < A1w) T.Rm[(size_t)c * w64 + 64 + lane] = keep1[u];
                if (lane == 0) {
                    const double n = c < C ? (double)T.count[c] : 0.0;
                    cnt_c[c] = n;
                    t0_c[c] = size[u] > 0 ? ((n) / ((double)size[u])) : 0.0;      // float(count) / len(alleles), common:1304
                }
            }
        }
    } else {
        // Gene_cmpt2 (core:1752-1766): every class filtered to the kept alleles (<= 64: one word over the compact alleles), the
        // empty ones dropped, equal ones merged with their counts added -- an LDS table keyed by the word -- in the order of
        // their first class (= the insertion order of the reference's dict)
        constexpr int MT = 4096;
        unsigned long long *mkey = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(&S.tmpo[0][0]) + (16 << 10));
        unsigned int *mfirst = reinterpret_cast<unsigned int *>(reinterpret_cast<char *>(&S.tmpo[0][0]) + (48 << 10));
        unsigned long long *mcnt = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(&S.tmpo[0][0]) + (66 << 10));
        unsigned int *mlist = reinterpret_cast<unsigned int *>(reinterpret_cast<char *>(&S.tmpo[0][0]) + (100 << 10));
        for (int i = tid; i < MT; i += XB) { mkey[i] = 0ull; mcnt[i] = 0ull; mfirst[i] = 0xFFFFFFFFu; }
        if (tid == 0) { S.npos[0] = 0; S.npos[1] = 0; }
        __syncthreads();
        unsigned long long *rowm = reinterpret_cast<unsigned long long *>(&S.tmpo[0][0]) + (size_t)wave * 128;    // (below the tables: 16 KB)
        for (int c = wave; c < C; c += XNW) {
            rowm[lane] = lane < w64 ? T.B[(size_t)c * w64 + lane] & T.mask[lane] : 0ull;
            rowm[64 + lane] = lane + 64 < w64 ? T.B[(size_t)c * w64 + 64 + lane] & T.mask[64 + lane] : 0ull;
            __builtin_amdgcn_wave_barrier();
            const int g = lane < A1 ? srt[lane] : 0;
            const unsigned long long m = __ballot(lane < A1 && ((rowm[g >> 6] >> (g & 63)) & 1ull));
            __builtin_amdgcn_wave_barrier();
            if (m != 0ull && lane == 0) {
                unsigned int h = (unsigned int)(mix64(m) >> 40) & (MT - 1);
                for (;;) {
                    const unsigned long long old = atomicCAS(&mkey[h], 0ull, m);
                    if (old == 0ull) atomicAdd(&S.npos[0], 1);
                    if (old == 0ull || old == m) break;
                    h = (h + 1) & (MT - 1);
                    if (S.npos[0] > MT / 2) break;
                }
                if (S.npos[0] <= MT / 2) { atomicAdd(&mcnt[h], (unsigned long long)T.count[c]); atomicMin(&mfirst[h], (unsigned int)c); }
            }
        }
        __syncthreads();
        const int C1 = S.npos[0];
        if (C1 > MT / 2 || C1 > T.c_alloc) { give_up(1.0); return; }
        for (int i = tid; i < MT; i += XB)
            if (mkey[i] != 0ull) mlist[atomicAdd(&S.npos[1], 1)] = (unsigned int)i;
        __syncthreads();
        const int C1p = (C1 + 63) & ~63;
        for (int e = tid; e < C1p; e += XB) {
            if (e < C1) {
                const unsigned int slot = mlist[e], f = mfirst[slot];
                int rk = 0;
                for (int o = 0; o < C1; ++o) rk += mfirst[mlist[o]] < f;       // first classes are distinct: a permutation
                const unsigned long long m = mkey[slot];
                const double n = (double)(long long)mcnt[slot];
                T.Mk[rk] = m;
                cnt_c[rk] = n;
                t0_c[rk] = ((n) / ((double)__popcll(m)));
            } else { T.Mk[e] = 0ull; cnt_c[e] = 0.0; t0_c[e] = 0.0; }
        }
        C = C1;
        __syncthreads();
    }
    const int Cp = (C + 63) & ~63, Cw = Cp >> 6;
    lap(8);
    if (wg == 0) for (int j = tid; j < A1s; j += XB) T.vlen[j] = (T.len && j < A1) ? T.len[srt[j]] : 1.0;
    phase_sync();
    cluster_sync();                                       // every class row is in place
    // ---- Mk (word (aw, c), coalesced over c) and Mr (64 x 64 bit transposes, coalesced over the alleles) from the class-major rows
    if (!T.mask) {
        // (class tile, eight allele tiles) per item.  The class-major rows are read the way they lie in memory -- 8 classes x 8
        // words = eight 64-byte runs per load instruction; lane = class would be 64 cache lines per load -- and change hands through
        // a 4 KB stage per wavefront in LDS (slot of word u of class k: 8 k + (u ^ (k & 7)))
        unsigned long long *stage = reinterpret_cast<unsigned long long *>(&S.tmpo[0][0]) + (size_t)wave * 512;
        for (int item = wave * n_wg + wg; item < Cw * ((A1w + 7) / 8); item += XNW * n_wg) {
            const int cw = item % Cw, a8 = item / Cw;
            const int c = 64 * cw + lane;
            {
                const int u = lane & 7, aw = 8 * a8 + u;
                uint64_t y[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int k = 8 * r + (lane >> 3);
                    y[r] = aw < A1w ? T.Rm[(size_t)(64 * cw + k) * w64 + aw] : 0ull;
                }
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int k = 8 * r + (lane >> 3);
                    stage[8 * k + (u ^ (k & 7))] = y[r];
                }
            }
            __builtin_amdgcn_wave_barrier();
            uint64_t x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = stage[8 * lane + (u ^ (lane & 7))];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int aw = 8 * a8 + u;
                if (aw < A1w) {
                    T.Mk[(size_t)aw * CpA + c] = x[u];
                    T.Mr[(size_t)cw * A1s + 64 * aw + lane] = wave_transpose64(x[u]);
                }
            }
        }
    } else {
        for (int item = wave; item < Cw * A1w; item += XNW) {
            const int cw = item / A1w, aw = item - cw * A1w;
            const uint64_t x = T.Mk[(size_t)aw * CpA + 64 * cw + lane];
            T.Mr[(size_t)cw * A1s + 64 * aw + lane] = wave_transpose64(x);
        }
    }
    lap(9);
    const bool use_len = T.len != nullptr;
    const int remove_low = T.remove_low;
    if (tid == 0) { S.cache_ord = -1; S.need_slow = 0; }
    uint16_t *posb[4] = {T.pos, T.pos + A1s, T.pos + 2 * (size_t)A1s, T.pos + 3 * (size_t)A1s};
    uint32_t *skeys = reinterpret_cast<uint32_t *>(&S.tmpo[1][0]);
    phase_sync();
    cluster_sync();                                       // Mk and Mr are complete
    if (wg == 0) for (int j = tid; j < A1; j += XB) {
        int fc = -1;
        for (int cw = 0; cw < Cw && fc < 0; ++cw) { const uint64_t m = T.Mr[(size_t)cw * A1s + j]; if (m) fc = 64 * cw + __builtin_ctzll(m); }
        T.first_c[j] = fc;
    }
    lap(0);

    // (table-lookup cluster: every workgroup keeps the three dicts for itself -- the vector steps are done by all of them alike, only
    // the two matrix passes are shared out -- so nothing but n_c and the raw result of a cols pass ever changes hands)
    const size_t dict_of_wg = (FAST && CL) ? (size_t)wg * 3 * (size_t)A1s : 0;
    double *dv[3] = {T.dv + dict_of_wg, T.dv + dict_of_wg + A1s, T.dv + dict_of_wg + 2 * (size_t)A1s};
    uint8_t *din[3] = {T.din + dict_of_wg, T.din + dict_of_wg + A1s, T.din + dict_of_wg + 2 * (size_t)A1s};
    double *tmpo0 = &S.tmpo[0][0], *tmpo1 = &S.tmpo[1][0];
    const int A1p8 = (A1 + 7) & ~7;
    int n_orders = 0, n_apps = 0;                       // (applications of the EM map: what the byte model of a launch is counted in)

    // ---- helpers ---------------------------------------------------------------------------------------------------
    auto block_max_exact = [&](double v) -> double {
        v = wave_max_nonneg_f64(v);
        __syncthreads();
        if (lane == 0) S.red[wave] = v;
        __syncthreads();
        double t = S.red[0];
#pragma unroll
        for (int i = 1; i < XNW; ++i) t = fmax(t, S.red[i]);
        return t;
    };
    // insertion order of dict d when it is filled class by class over the classes of `walk` (NULL = all): (first class, key order)
    auto derive_order = [&](int d, const unsigned long long *walk, int buf) {
        int N = 1024;
        while (N < A1) N <<= 1;
        for (int j = tid; j < N; j += XB) {
            uint32_t key = 0xFFFFFFFFu;
            if (j < A1 && din[d][j]) {
                int fc = -1;
                for (int cw = 0; cw < Cw && fc < 0; ++cw) {
                    const uint64_t m = T.Mr[(size_t)cw * A1s + j] & (walk ? walk[cw] : ~0ull);
                    if (m) fc = 64 * cw + __builtin_ctzll(m);
                }
                if (fc >= 0) key = (uint32_t)fc << 13 | (uint32_t)j;
            }
            skeys[j] = key;
        }
        __syncthreads();
        for (int k = 2; k <= N; k <<= 1)
            for (int j2 = k >> 1; j2 > 0; j2 >>= 1) {
                for (int idx = tid; idx < N / 2; idx += XB) {
                    const int i = ((idx & ~(j2 - 1)) << 1) | (idx & (j2 - 1)), p = i | j2;
                    const uint32_t a = skeys[i], b = skeys[p];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { skeys[i] = b; skeys[p] = a; }
                }
                __syncthreads();
            }
        for (int j = tid; j < A1s; j += XB) posb[buf][j] = 0xFFFFu;
        if (tid == 0) S.npos[buf] = 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int r = tid; r < N; r += XB) {
            const uint32_t kk = skeys[r];
            if (kk != 0xFFFFFFFFu) {
                posb[buf][kk & 8191u] = (uint16_t)r;
                if (r + 1 == N || skeys[r + 1] == 0xFFFFFFFFu) S.npos[buf] = r + 1;      // the keys are sorted: members first
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        n_orders += 1;
    };
    // operands of a sequential sum over dict d in the order of buffer `ord`: dst[pos] = member ? val : +0.0, tail padded
    auto store_ordered = [&](double *dst, int ord, int j, bool member, double val) {
        const uint32_t p = posb[ord][j];
        if (p != 0xFFFFu) dst[p] = member ? val : 0.0;
    };
    auto pad_ordered = [&](double *dst, int ord) {
        const int np = S.npos[ord];
        if (tid < 8 && np + tid < ((np + 7) & ~7)) dst[np + tid] = 0.0;
    };
    auto normalize = [&](int d, int ord) {                 // common:1285-1297
        for (int j = tid; j < A1; j += XB) {
            const double mine = use_len ? ((dv[d][j]) / (T.vlen[j])) : dv[d][j];
            T.tmpv[j] = mine;
            store_ordered(tmpo0, ord, j, din[d][j] != 0, mine);
        }
        pad_ordered(tmpo0, ord);
        __syncthreads();
        if (wave == 0) { const double t = seq_sum(tmpo0, (S.npos[ord] + 7) & ~7); if (lane == 0) S.bc[0] = t; }
        __syncthreads();
        const double total = S.bc[0];
        for (int j = tid; j < A1; j += XB)
            if (din[d][j]) dv[d][j] = ((T.tmpv[j]) / (total));
        __syncthreads();
    };
    int ord_of[3] = {0, 0, 0};
    // Gene_prob_next (common:1311-1336): dict P -> dict N (N != P)
    enum : unsigned { CMD_EXIT = 0u, CMD_NEXT = 0x100u, CMD_INIT = 0x200u };
    // Cluster mode walks its tiles with ONE or two wavefronts per SIMD, so nothing hides the latency of a scalar load per eight
    // steps (it was 2/3 of a walk).  There the wave-uniform operands of 64 steps are fetched by the lanes (coalesced), parked in a
    // per-wavefront LDS block and read back as broadcasts -- plain loads that the compiler keeps in flight across the steps -- while
    // the next block's global loads are under way.  Same operations in the same order as the scalar-cache walks.
    double *stg = reinterpret_cast<double *>(&S.tmpo[0][0]) + (size_t)wave * 512;          // two buffers of four 64-entry arrays
    auto uni64 = [&](double v) -> uint64_t {
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
    };
    // rows: alleles_prob of the classes of tile cw, alleles in key order.  A non-member's value is +0.0 in every dict
    // (next_prob, select_alleles) and x + 0.0 == x: no membership test.
    auto rows_share = [&](int P) {
        for (int cw = wave * n_wg + wg; cw < Cw; cw += XNW * n_wg) {
            double acc = 0.0;
            const uint64_t *mrow = T.Mr + (size_t)cw * A1s;
            if constexpr (CL) {
                double m_nx = __longlong_as_double((long long)mrow[lane]), p_nx = dv[P][lane];
                for (int blk = 0; blk < A1w; ++blk) {
                    double *b = stg + (blk & 1) * 256;
                    b[lane] = m_nx; b[64 + lane] = p_nx;
                    __builtin_amdgcn_wave_barrier();
                    if (blk + 1 < A1w) { m_nx = __longlong_as_double((long long)mrow[64 * (blk + 1) + lane]); p_nx = dv[P][64 * (blk + 1) + lane]; }
                    // eight steps' operands at a time, the NEXT eight requested before the current eight dependent adds (round 6: with
                    // sixteen read and then sixteen added, every group waited ~200 cycles for its own LDS reads first)
                    double mv[2][8], xv[2][8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { mv[0][u] = b[u]; xv[0][u] = b[64 + u]; }
#pragma unroll
                    for (int g = 0; g < 8; ++g) {
                        if (g + 1 < 8) {
#pragma unroll
                            for (int u = 0; u < 8; ++u) { mv[(g + 1) & 1][u] = b[8 * (g + 1) + u]; xv[(g + 1) & 1][u] = b[64 + 8 * (g + 1) + u]; }
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) acc = sel_add(acc, uni64(mv[g & 1][u]), xv[g & 1][u]);
                    }
                }
            } else {
                for (int j0 = 0; j0 < A1p8; j0 += 8) {
                    u32x16 mw, pv;
                    sload2(mrow + j0, dv[P] + j0, mw, pv);
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc = sel_add(acc, u64_of(mw, k), dbl_of(pv, k));
                }
            }
            const int c = 64 * cw + lane;
            const bool valid = c < C && acc > 0.0;        // classes with alleles_prob <= 0 are skipped (common:1321)
            // what the compiler's division computes from the denominator alone (no rescaling inside [2^-200, 2^200])
            double r2 = 1.0;
            bool slow = false;
            if (valid) {
                slow = !(acc >= 0x1p-200 && acc <= 0x1p200);
                const double r0 = __builtin_amdgcn_rcp(acc);
                const double e0 = __builtin_fma(-acc, r0, 1.0);
                const double r1 = __builtin_fma(r0, e0, r0);
                const double e1 = __builtin_fma(-acc, r1, 1.0);
                r2 = __builtin_fma(r1, e1, r1);
            }
            n_c[c] = valid ? cnt_c[c] : 0.0;
            s_c[c] = valid ? acc : 1.0;
            r_c[c] = r2;
            const unsigned long long bm = __ballot(valid);
            if constexpr (CL) {
                if (lane == 0) T.gvalid[cw] = bm;
                if (__any(slow) && lane == 0) atomicOr(&T.cl_ctl[3], 1u);
            } else {
                if (lane == 0) S.validw[cw] = bm;
                if (__any(slow) && lane == 0) atomicOr(&S.need_slow, 1);
            }
        }
    };
    // cols: next[a] += count * prob / alleles_prob over the walked classes in dict order (a skipped class adds +0.0)
    auto cols_share = [&](int P, int N) {
        const unsigned long long *validw = CL ? T.gvalid : S.validw;
        const bool block_slow = CL ? (T.cl_ctl[3] != 0u) : (S.need_slow != 0);
        for (int aw = wave * n_wg + wg; aw < A1w; aw += XNW * n_wg) {
            const int j = 64 * aw + lane;
            const bool alive = j < A1;
            const bool pin = alive && din[P][j] != 0;
            const double p = pin ? dv[P][j] : 0.0;
            const bool lane_fast = p == 0.0 || (p >= 0x1p-600 && p <= 0x1p600);
            const bool tile_slow = block_slow || __any(!lane_fast);
            double acc = 0.0;
            const uint64_t *mcol = T.Mk + (size_t)aw * CpA;
            if (CL && !tile_slow) {
                double m_nx = __longlong_as_double((long long)mcol[lane]), n_nx = n_c[lane], s_nx = s_c[lane], r_nx = r_c[lane];
                for (int blk = 0; blk < Cw; ++blk) {
                    double *b = stg + (blk & 1) * 256;
                    b[lane] = m_nx; b[64 + lane] = n_nx; b[128 + lane] = s_nx; b[192 + lane] = r_nx;
                    __builtin_amdgcn_wave_barrier();
                    if (blk + 1 < Cw) {
                        const int c = 64 * (blk + 1) + lane;
                        m_nx = __longlong_as_double((long long)mcol[c]); n_nx = n_c[c]; s_nx = s_c[c]; r_nx = r_c[c];
                    }
                    for (int k0 = 0; k0 < 64; k0 += 8) {       // eight classes' operands in flight, the quotients, then the dependent adds
                        double mv[8], nv[8], sv[8], rv[8], q[8];    // (four at a time: 36.2 ms of cols per EM instead of 33.6; a two-set
                                                                    // software pipeline does not fit the 128 registers of a 1024-thread block)
#pragma unroll
                        for (int u = 0; u < 8; ++u) { mv[u] = b[k0 + u]; nv[u] = b[64 + k0 + u]; sv[u] = b[128 + k0 + u]; rv[u] = b[192 + k0 + u]; }
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const double x = ((nv[u]) * (p));
                            const double q0 = ((x) * (rv[u]));
                            const double e = __builtin_fma(-sv[u], q0, x);
                            q[u] = __builtin_fma(e, rv[u], q0);
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) acc = sel_add(acc, uni64(mv[u]), q[u]);
                    }
                }
            } else if (!tile_slow) {
                for (int c0 = 0; c0 < Cp; c0 += 8) {
                    u32x16 mw, vn, vs, vr;
                    sload4(mcol + c0, n_c + c0, s_c + c0, r_c + c0, mw, vn, vs, vr);
                    double q[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const double s = dbl_of(vs, k), r = dbl_of(vr, k);
                        const double x = ((dbl_of(vn, k)) * (p));
                        const double q0 = ((x) * (r));
                        const double e = __builtin_fma(-s, q0, x);
                        q[k] = __builtin_fma(e, r, q0);
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc = sel_add(acc, u64_of(mw, k), q[k]);
                }
            } else {
                for (int c0 = 0; c0 < Cp; c0 += 8) {
                    u32x16 mw, vn, vs, vr;
                    sload4(mcol + c0, n_c + c0, s_c + c0, r_c + c0, mw, vn, vs, vr);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const double q = ((((dbl_of(vn, k)) * (p))) / (dbl_of(vs, k)));
                        acc = sel_add(acc, u64_of(mw, k), q);
                    }
                }
            }
            bool seen = false;                             // some walked class contains the allele: it enters the next dict
            if (alive)
                for (int cw = 0; cw < Cw; ++cw) seen = seen || (T.Mr[(size_t)cw * A1s + j] & validw[cw]) != 0ull;
            const bool nin = pin && seen;
            if (alive) { dv[N][j] = nin ? acc : 0.0; din[N][j] = nin ? 1 : 0; }
            const unsigned long long inb = __ballot(nin);
            if constexpr (CL) { if (lane == 0) T.gin[aw] = inb; }
            else { if (lane == 0) S.in_now[aw] = inb; }
        }
    };
    auto next_prob = [&](int P, int N, int live_a, int live_b) {
        n_apps += 1;
        if (tid == 0) {
            S.need_slow = 0;
            if constexpr (CL) { T.cl_ctl[3] = 0u; T.cl_ctl[2] = CMD_NEXT | (unsigned)P | ((unsigned)N << 2); }
        }
        phase_sync();                                      // dv[P] as written by this workgroup is what the scalar loads see
        cluster_sync();                                    // (cluster: the helpers take the command; dv[P] is complete for them)
        rows_share(P);
        phase_sync();
        cluster_sync();
        lap(1);
        cols_share(P, N);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cluster_sync();
        if constexpr (CL) {                                // what the tile loops of all workgroups found, into the leader's LDS
            for (int w = tid; w < Cw; w += XB) S.validw[w] = T.gvalid[w];
            for (int w = tid; w < A1w; w += XB) S.in_now[w] = T.gin[w];
            __syncthreads();
        }
        lap(2);
        // insertion order: unchanged unless the membership or the walked classes changed since it was last derived
        bool differ = S.cache_ord < 0;
        if (tid < A1w) differ = differ || S.in_now[tid] != S.sig_in[tid];
        if (tid < Cw) differ = differ || S.validw[tid] != S.sig_valid[tid];
        const int changed = __syncthreads_or(differ);
        int ord;
        if (!changed) ord = S.cache_ord;
        else {
            ord = 0;
            while (ord == live_a || ord == live_b) ++ord;  // a buffer no live dict refers to (4 buffers, <= 2 live besides N)
            derive_order(N, S.validw, ord);
            if (tid < A1w) S.sig_in[tid] = S.in_now[tid];
            if (tid < Cw) S.sig_valid[tid] = S.validw[tid];
            if (tid == 0) S.cache_ord = ord;
            __syncthreads();
        }
        lap(3);
        ord_of[N] = ord;
        normalize(N, ord);
        lap(4);
    };
    auto select_alleles = [&](int d) {                     // common:1338-1346
        double mx = 0.0;
        for (int j = tid; j < A1; j += XB) if (din[d][j]) mx = fmax(mx, dv[d][j]);
        mx = block_max_exact(mx);
        for (int j = tid; j < A1; j += XB)
            if (din[d][j] && !(dv[d][j] >= ((mx) / (10.0)))) { din[d][j] = 0; dv[d][j] = 0.0; }
        __syncthreads();
    };

    int prob = 0, next = 1, next2 = 2;
    if (wg == 0 || (FAST && CL))
        for (int j = tid; j < A1s; j += XB)
            for (int d = 0; d < 3; ++d) { dv[d][j] = 0.0; din[d][j] = 0; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    double diff = 1.0;
    int iter = 0;
    bool keyerr = false;
    if constexpr (FAST) {
        // ---- the same EM with table-lookup mat-vecs and tree reductions (any summation order): T(p)_a = p_a * sum_c n_c / s_c ----
        double *Tb = &S.tmpo[0][0];                        // [64][256]
        auto drain = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); };
        // cluster (k_emx<true, true>): the lookups of a pass are shared out by WAVEFRONT (wave w of every 1 024-row tile belongs to
        // workgroup w % n_wg: the passes are bound by VALU issue, so half the wavefronts take half the time); the tables are
        // built by every workgroup.  A rows pass hands over n_c, a cols pass its raw sums and membership flags (vraw / vin);
        // everything else every workgroup does for itself, on its own dicts, bit for bit what one workgroup does.
        // whole 1 024-row tiles where a pass has at least one per workgroup (a tile's time does not depend on how many of its
        // wavefronts look up), else the wavefronts of the tiles
        const int tiles_c = (Cp + XB - 1) / XB, tiles_a = (64 * A1w + XB - 1) / XB;
        auto mine_of = [&](int k, int n_tiles) -> bool {
            if constexpr (!CL) return true;
            return n_tiles >= n_wg ? (k % n_wg) == wg : (wave % n_wg) == wg;
        };
        double *vraw = T.tmpv;
        unsigned int *vin = reinterpret_cast<unsigned int *>(T.pos);
        auto block_sum = [&](double v) __attribute__((always_inline)) -> double {
            v = wave_sum_f64(v);
            __syncthreads();
            if (lane == 0) S.red[wave] = v;
            __syncthreads();
            double t = 0.0;
#pragma unroll
            for (int i = 0; i < XNW; ++i) t = t + S.red[i];
            return t;
        };
        // y_j = sum over the classes of x_c [class contains j], for every compact allele j; lane = allele, 512 classes per table
        // y_j = sum over the classes of x_c [class contains j], for every compact allele j; lane = allele, 512 classes per table.
        // (A (slab, tile) unit is 64 lookups per lane = 512 KB of LDS reads per workgroup: ~1.7 us at the CU's 128 B per clock, and
        // the units measure 2.0-2.3 us -- the passes sit on the LDS pipe; requesting the next unit's matrix words ahead of the
        // lookups changed nothing but the spill count: NOTEBOOK.md section 11.)
        auto cols_lut = [&](const double *x, bool handed, double (&acc)[8]) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = 0.0;
            for (int sl = 0; sl * 512 < Cp; ++sl) {
                __syncthreads();
                if (tid < 512) { const int c = 512 * sl + tid; S.xs[tid] = c < Cp ? ((CL && handed) ? ld_agent(&x[c]) : x[c]) : 0.0; }
                __syncthreads();
                xlut_build(S.xs, Tb, tid);
                __syncthreads();
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    int j = tid + XB * k;
                    asm volatile("" : "+v"(j));            // (addresses formed here, not kept in 128 registers across the slabs)
                    if (j < 64 * A1w && mine_of(k, tiles_a)) {
                        uint64_t w[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) { const int cw = 8 * sl + i; w[i] = cw < Cw ? T.Mr[(size_t)cw * A1s + j] : 0ull; }
                        acc[k] += xlut_row(Tb, w);
                    }
                    asm volatile("" ::: "memory");         // (one allele's eight words at a time: keeps the eight accumulators in registers)
                }
            }
        };
        // Gene_prob_next (common:1311-1336) from dict P into dict N, normalised
        auto next_fast = [&](int P, int N) __attribute__((always_inline)) {
            n_apps += 1;
            double sacc[4] = {0.0, 0.0, 0.0, 0.0};         // rows: alleles_prob of class tid + 1024 k; 512 alleles per table
            for (int sl = 0; sl * 512 < A1; ++sl) {
                __syncthreads();
                if (tid < 512) { const int j = 512 * sl + tid; S.xs[tid] = j < A1 ? dv[P][j] : 0.0; }    // (a non-member holds 0)
                __syncthreads();
                xlut_build(S.xs, Tb, tid);
                __syncthreads();
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    int c = tid + XB * k;
                    asm volatile("" : "+v"(c));
                    if (c < Cp && mine_of(k, tiles_c)) {
                        uint64_t w[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) { const int aw = 8 * sl + i; w[i] = aw < A1w ? T.Mk[(size_t)aw * CpA + c] : 0ull; }
                        sacc[k] += xlut_row(Tb, w);
                    }
                    asm volatile("" ::: "memory");
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = tid + XB * k;
                if (c < Cp && mine_of(k, tiles_c)) {          // classes with alleles_prob <= 0 are skipped
                    const double v = (c < C && sacc[k] > 0.0) ? cnt_c[c] / sacc[k] : 0.0;
                    if constexpr (CL) st_agent(&n_c[c], v); else n_c[c] = v;
                }
            }
            drain();
            cluster_sync_light();                          // (cluster: n_c is complete)
            lap(1);
            double acc[8];
            cols_lut(n_c, true, acc);
            double part = 0.0;
            if constexpr (CL) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int j = tid + XB * k;
                    if (j < A1 && mine_of(k, tiles_a)) {
                        const bool nin = din[P][j] != 0 && acc[k] > 0.0;
                        double v = nin ? dv[P][j] * acc[k] : 0.0;
                        if (use_len) v = v / T.vlen[j];
                        st_agent(&vraw[j], v);
                        st_agent(&vin[j], nin ? 1u : 0u);
                    }
                }
                drain();
                cluster_sync_light();                      // (every allele's raw value is there)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int j = tid + XB * k;
                    if (j < A1) { const double v = ld_agent(&vraw[j]); dv[N][j] = v; din[N][j] = (uint8_t)ld_agent(&vin[j]); part += v; }
                }
            } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int j = tid + XB * k;
                if (j < A1) {
                    const bool nin = din[P][j] != 0 && acc[k] > 0.0;
                    double v = nin ? dv[P][j] * acc[k] : 0.0;
                    if (use_len) v = v / T.vlen[j];
                    dv[N][j] = v;
                    din[N][j] = nin ? 1 : 0;
                    part += v;
                }
            }
            }
            const double total = block_sum(part);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int j = tid + XB * k;
                if (j < A1 && din[N][j]) dv[N][j] = dv[N][j] / total;
            }
            drain();
            lap(2);
        };
        auto select_fast = [&](int d) __attribute__((always_inline)) {
            double mx = 0.0;
            for (int j = tid; j < A1; j += XB) if (din[d][j]) mx = fmax(mx, dv[d][j]);
            mx = block_max_exact(mx);
            for (int j = tid; j < A1; j += XB)
                if (din[d][j] && !(dv[d][j] >= mx / 10.0)) { din[d][j] = 0; dv[d][j] = 0.0; }
            drain();
        };
        {   // initial estimate (common:1300-1309)
            double acc[8], part = 0.0;
            cols_lut(t0_c, false, acc);
            if constexpr (CL) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int j = tid + XB * k;
                    if (j < A1 && mine_of(k, tiles_a)) st_agent(&vraw[j], use_len ? acc[k] / T.vlen[j] : acc[k]);
                }
                drain();
                cluster_sync_light();
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int j = tid + XB * k;
                if (j < A1) { const double v = CL ? ld_agent(&vraw[j]) : (use_len ? acc[k] / T.vlen[j] : acc[k]); dv[prob][j] = v; din[prob][j] = 1; part += v; }
            }
            const double total = block_sum(part);
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int j = tid + XB * k; if (j < A1) dv[prob][j] = dv[prob][j] / total; }
            drain();
            lap(5);
        }
        while (diff > 0.0001 && iter < 1000 && !cl_dead) { // common:1351
            next_fast(prob, next);
            next_fast(next, next2);
            bool bad = false;
            double sr = 0.0, sv = 0.0;
            for (int j = tid; j < A1; j += XB) {
                if (!din[prob][j]) continue;
                bad = bad || !din[next][j] || !din[next2][j];
                const double p_r = dv[next][j] - dv[prob][j];
                const double p_v = (dv[next2][j] - dv[next][j]) - p_r;
                sr += p_r * p_r;
                sv += p_v * p_v;
            }
            if (__syncthreads_or(bad)) { keyerr = true; break; }      // the reference's KeyError (Q6)
            const double ssr = block_sum(sr), ssv = block_sum(sv);
            if (ssv > 0.0) {                               // common:1370-1383
                const double gamma = -sqrt(ssr / ssv);
                for (int j = tid; j < A1; j += XB)
                    if (din[prob][j]) {
                        const double pv0 = dv[prob][j];
                        const double p_r = dv[next][j] - pv0;
                        const double p_v = (dv[next2][j] - dv[next][j]) - p_r;
                        const double x = (pv0 - (2.0 * gamma) * p_r) + (gamma * gamma) * p_v;
                        dv[next2][j] = 0.0 > x ? 0.0 : x;
                    }
                drain();
                next_fast(next2, next);
            }
            double dd = 0.0;
            for (int j = tid; j < A1; j += XB)
                if (din[prob][j]) dd += din[next][j] ? fabs(dv[prob][j] - dv[next][j]) : dv[prob][j];
            diff = block_sum(dd);
            { const int t = prob; prob = next; next = t; }
            if (iter >= 10 && remove_low) select_fast(prob);
            iter += 1;
            lap(6);
        }
        if constexpr (CL) {
            if (cl_dead) { if (wg == 0) give_up(3.0); return; }     // (not co-resident in time: status 3, the caller re-runs the job on one workgroup)
            if (wg != 0) return;                           // (the result is the leader's to report)
        }
        if (!keyerr) {
            if (remove_low) select_fast(prob);
            double part = 0.0;
            for (int j = tid; j < A1; j += XB) if (din[prob][j]) part += use_len ? dv[prob][j] / T.vlen[j] : dv[prob][j];
            const double total = block_sum(part);
            for (int j = tid; j < A1; j += XB)
                if (din[prob][j]) dv[prob][j] = use_len ? dv[prob][j] / T.vlen[j] / total : dv[prob][j] / total;
            drain();
        }
    } else {
    // ---- initial estimate (common:1300-1309): prob[a] += count / |class| over the classes in dict order ---------------
    auto init_share = [&](int d) {
        for (int aw = wave * n_wg + wg; aw < A1w; aw += XNW * n_wg) {
            const int j = 64 * aw + lane;
            double acc = 0.0;
            const uint64_t *mcol = T.Mk + (size_t)aw * CpA;
            if constexpr (CL) {
                double m_nx = __longlong_as_double((long long)mcol[lane]), t_nx = t0_c[lane];
                for (int blk = 0; blk < Cw; ++blk) {
                    double *b = stg + (blk & 1) * 256;
                    b[lane] = m_nx; b[64 + lane] = t_nx;
                    __builtin_amdgcn_wave_barrier();
                    if (blk + 1 < Cw) { m_nx = __longlong_as_double((long long)mcol[64 * (blk + 1) + lane]); t_nx = t0_c[64 * (blk + 1) + lane]; }
                    for (int k0 = 0; k0 < 64; k0 += 16) {      // sixteen steps' operands in flight, then the sixteen dependent adds
                        double mv[16], xv[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) { mv[u] = b[k0 + u]; xv[u] = b[64 + k0 + u]; }
#pragma unroll
                        for (int u = 0; u < 16; ++u) acc = sel_add(acc, uni64(mv[u]), xv[u]);
                    }
                }
            } else {
                for (int c0 = 0; c0 < Cp; c0 += 8) {
                    u32x16 mw, vt;
                    sload2(mcol + c0, t0_c + c0, mw, vt);
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc = sel_add(acc, u64_of(mw, k), dbl_of(vt, k));
                }
            }
            if (j < A1) { dv[d][j] = acc; din[d][j] = 1; }
        }
    };
    if constexpr (CL) {
        if (wg != 0) {                                     // a helper: the tile loops the leader asks for, until it says stop
            for (;;) {
                cluster_sync();
                if (cl_dead) return;
                const unsigned cmd = (unsigned)__builtin_amdgcn_readfirstlane((int)T.cl_ctl[2]);
                if (cmd == CMD_EXIT) return;
                if (cmd & CMD_INIT) { init_share((int)(cmd & 3u)); cluster_sync(); continue; }
                rows_share((int)(cmd & 3u));
                phase_sync();
                cluster_sync();
                cols_share((int)(cmd & 3u), (int)((cmd >> 2) & 3u));
                cluster_sync();
            }
        }
        if (tid == 0) T.cl_ctl[2] = CMD_INIT | (unsigned)prob;
        phase_sync();
        cluster_sync();
    }
    init_share(prob);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cluster_sync();
    derive_order(prob, nullptr, 3);
    ord_of[prob] = 3;
    normalize(prob, 3);
    lap(5);
    while (diff > 0.0001 && iter < 1000 && !cl_dead) {     // common:1351
        next_prob(prob, next, ord_of[prob], -1);
        next_prob(next, next2, ord_of[prob], ord_of[next]);
        bool bad = false;
        for (int j = tid; j < A1; j += XB) bad = bad || (din[prob][j] && (!din[next][j] || !din[next2][j]));
        if (__syncthreads_or(bad)) { keyerr = true; break; }      // the reference's KeyError (Q6)
        const int op = ord_of[prob];
        for (int j = tid; j < A1; j += XB) {
            const bool pin = din[prob][j] != 0;
            const double pv0 = dv[prob][j];
            const double p_r = ((dv[next][j]) - (pv0));
            const double p_v = ((((dv[next2][j]) - (dv[next][j]))) - (p_r));
            store_ordered(tmpo0, op, j, pin, ((p_r) * (p_r)));
            store_ordered(tmpo1, op, j, pin, ((p_v) * (p_v)));
        }
        pad_ordered(tmpo0, op);
        pad_ordered(tmpo1, op);
        __syncthreads();
        if (wave == 0) {
            double ta, tb;
            seq_sum2(tmpo0, tmpo1, (S.npos[op] + 7) & ~7, ta, tb);
            if (lane == 0) { S.bc[0] = ta; S.bc[1] = tb; }
        }
        __syncthreads();
        const double ssr = S.bc[0], ssv = S.bc[1];
        __syncthreads();
        if (ssv > 0.0) {                                   // common:1370-1383
            const double gamma = -sqrt(((ssr) / (ssv)));
            for (int j = tid; j < A1; j += XB)
                if (din[prob][j]) {
                    const double pv0 = dv[prob][j];
                    const double p_r = ((dv[next][j]) - (pv0));
                    const double p_v = ((((dv[next2][j]) - (dv[next][j]))) - (p_r));
                    const double x = ((((pv0) - (((((2.0) * (gamma))) * (p_r))))) + (((((gamma) * (gamma))) * (p_v))));
                    dv[next2][j] = 0.0 > x ? 0.0 : x;
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            next_prob(next2, next, ord_of[prob], ord_of[next2]);
        }
        for (int j = tid; j < A1; j += XB) {               // prob_diff, common:1272-1279
            const bool pin = din[prob][j] != 0;
            const double pv0 = dv[prob][j];
            store_ordered(tmpo0, op, j, pin, din[next][j] ? fabs(((pv0) - (dv[next][j]))) : pv0);
        }
        pad_ordered(tmpo0, op);
        __syncthreads();
        if (wave == 0) { const double t = seq_sum(tmpo0, (S.npos[op] + 7) & ~7); if (lane == 0) S.bc[0] = t; }
        __syncthreads();
        diff = S.bc[0];
        __syncthreads();
        { const int t = prob; prob = next; next = t; }     // prob = next (common:1387)
        if (iter >= 10 && remove_low) select_alleles(prob);
        iter += 1;
        lap(6);
    }
    if constexpr (CL) {
        if (tid == 0) T.cl_ctl[2] = CMD_EXIT;
        cluster_sync();                                    // the helpers leave
        if (cl_dead) { give_up(3.0); return; }             // (a cluster that was not co-resident in time: status 3, the caller re-runs the job on one workgroup)
    }
    if (!keyerr) {
        if (remove_low) select_alleles(prob);              // common:1402-1407
        normalize(prob, ord_of[prob]);
    }
    }   // (exact mode)
    // the returned dict as records (allele, first class, abundance): members counted per tile, one atomic add reserves the run
    int res_n = 0;
    if (!keyerr) {
        for (int aw = wave; aw < A1w; aw += XNW) {
            const int j = 64 * aw + lane;
            const unsigned long long m = __ballot(j < A1 && din[prob][j] != 0);
            if (lane == 0) S.in_now[aw] = m;
        }
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int aw = 0; aw < A1w; ++aw) { S.rpre[aw] = t; t += __popcll(S.in_now[aw]); }
            S.npos[0] = t;
            S.res_base = (int)atomicAdd(T.cursor, (unsigned long long)t);
        }
        __syncthreads();
        res_n = S.npos[0];
        for (int j = tid; j < A1; j += XB)
            if (din[prob][j]) {
                const int aw = j >> 6;
                const int idx = S.res_base + S.rpre[aw] + __popcll(S.in_now[aw] & ((1ull << (j & 63)) - 1ull));
                EmxRes r;
                r.allele = T.sorted[j]; r.first = T.first_c[j]; r.prob = dv[prob][j];
                r.order = FAST ? -1 : (int)posb[ord_of[prob]][j];
                r.pad_ = 0;
                T.res[idx] = r;
            }
    }
    if (tid == 0) {
        T.scal[XS_ITER] = (double)iter;
        T.scal[XS_STATUS] = keyerr ? 2.0 : 0.0;
        T.scal[XS_A1] = (double)A1;
        T.scal[XS_ORDERS] = (double)n_orders;
        T.scal[XS_NCLS] = (double)C;
        T.scal[XS_APPS] = (double)n_apps;
        T.scal[XS_RES_OFF] = (double)S.res_base;
        T.scal[XS_RES_N] = (double)res_n;
    }
    if (T.stamps && tid == 0) for (int k = 0; k < 12; ++k) T.stamps[k] = acc_t[k];
}
#pragma clang fp contract(fast)

inline size_t up64(size_t n) { return (n + 63) & ~(size_t)63; }

// kernel timing for roofline reports (bench.py): HIP events around the k_emx launches of the calls made while it is on
struct EmxStats { double ms = 0.0; long long launches = 0, apps = 0, bytes = 0, jobs = 0; };
std::mutex g_emx_mu;
bool g_emx_timing = false;
EmxStats g_emx_stats[2];           // [0] reference order, [1] table lookups

}   // namespace

extern "C" int hgx_emx_set_timing(int on) {
    std::lock_guard<std::mutex> g(g_emx_mu);
    if (on && !g_emx_timing) { g_emx_stats[0] = EmxStats(); g_emx_stats[1] = EmxStats(); }
    g_emx_timing = on != 0;
    return HGX_OK;
}
// totals since timing was switched on, for the launches in the reference's order (fast = 0) or with table lookups (fast = 1):
// kernel milliseconds, launches, jobs, applications of the EM map, and the algorithmic bytes of those applications
// (per application C * A' / 8 + 16 A' + 16 C with A' = the distinct alleles of the job's classes)
extern "C" int hgx_emx_get_timing(int fast, double *ms, long long *launches, long long *jobs, long long *apps, long long *bytes) {
    std::lock_guard<std::mutex> g(g_emx_mu);
    const EmxStats &s = g_emx_stats[fast ? 1 : 0];
    if (ms) *ms = s.ms;
    if (launches) *launches = s.launches;
    if (jobs) *jobs = s.jobs;
    if (apps) *apps = s.apps;
    if (bytes) *bytes = s.bytes;
    return HGX_OK;
}

static int emx_run(hgx_emx_job *jobs, int n_jobs, hipStream_t st, std::vector<hgx_emx_rec> *recs_out, bool allow_cluster);
// cluster launches and give-ups since the library was loaded (hgx_emx_cluster_stats; printed under HGX_TYPE_PROFILE)
static std::atomic<long long> g_cluster_jobs{0}, g_cluster_fallbacks{0};
extern "C" int hgx_emx_cluster_stats(long long *cluster_jobs, long long *fallbacks) {
    if (cluster_jobs) *cluster_jobs = g_cluster_jobs.load();
    if (fallbacks) *fallbacks = g_cluster_fallbacks.load();
    return HGX_OK;
}
int hgx_emx_run(hgx_emx_job *jobs, int n_jobs, hipStream_t st, std::vector<hgx_emx_rec> *recs_out) {
    int rc = emx_run(jobs, n_jobs, st, recs_out, true);
    if (rc) return rc;
    // A cluster that was not co-resident in time gave up (status 3): THOSE jobs again, each on one workgroup -- the same sums in
    // the same orders, ~7x slower; the others keep their results.  Counted and, under HGX_TYPE_PROFILE, said.
    std::vector<int> again;
    for (int i = 0; i < n_jobs; ++i) if (jobs[i].status == 3) again.push_back(i);
    if (again.empty()) return HGX_OK;
    g_cluster_fallbacks.fetch_add((long long)again.size());
    if (getenv("HGX_TYPE_PROFILE"))
        fprintf(stderr, "[hgx_emx_run] %zu cluster problem(s) were not co-resident in time: re-run on one workgroup each\n", again.size());
    std::vector<hgx_emx_job> sub;
    for (int i : again) sub.push_back(jobs[i]);
    std::vector<hgx_emx_rec> recs2;
    rc = emx_run(sub.data(), (int)sub.size(), st, recs_out ? &recs2 : nullptr, false);
    if (rc) return rc;
    const size_t shift = recs_out ? recs_out->size() : 0;
    if (recs_out) recs_out->insert(recs_out->end(), recs2.begin(), recs2.end());
    for (size_t k = 0; k < again.size(); ++k) {
        hgx_emx_job &J = jobs[again[k]];
        J.status = sub[k].status; J.n_iter = sub[k].n_iter; J.n_classes = sub[k].n_classes;
        J.rec_off = sub[k].rec_off + shift; J.n_rec = sub[k].n_rec;
    }
    return HGX_OK;
}
static int emx_run(hgx_emx_job *jobs, int n_jobs, hipStream_t st, std::vector<hgx_emx_rec> *recs_out, bool allow_cluster) {
    ARGCHK(n_jobs >= 0);
    if (n_jobs == 0) return HGX_OK;
    ARGCHK(jobs);
    HGX_ONCE_PER_DEVICE({
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_emx<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(XLds)));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_emx<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(XLds)));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_emx<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(XLds)));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_emx<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(XLds)));
    });
    // cluster mode: a problem far beyond the default gate (the any-size mode of hgx_em / hgx_type_*) gets a launch of its own on several
    // workgroups; such launches follow each other on the stream (one problem's cluster fills a good part of the chip)
    const bool clusters = allow_cluster && !hgx_test_switch("emx_no_cluster");
    // ... and a sizeable reference-order problem that has the launch (nearly) to itself -- EM #1 of ONE sample typed by a one-task call:
    // 1 730 classes x 4 500 alleles of a 10 000-read sample take 12 ms on one workgroup (the dense class x allele walk is bound by one
    // CU's FP64 issue) -- is shared out the same way: same sums in the same orders, same bits (test switch emx_cluster_lone = 0: off)
    int n_plain_big = 0;
    for (int i = 0; i < n_jobs; ++i) n_plain_big += (!jobs[i].fast && !jobs[i].mask && (int64_t)jobs[i].C * jobs[i].a_pad >= (int64_t)512 * 4096) ? 1 : 0;
    const char *lone_sw = hgx_test_switch("emx_cluster_lone");
    // (up to eight of them side by side: the loci of ONE sample typed together by hgx_type_many_loci -- typing() over a locus_list)
    const bool lone_ok = clusters && n_jobs <= 16 && n_plain_big >= 1 && n_plain_big <= 8 && !(lone_sw && atoi(lone_sw) == 0);
    auto wants_cluster = [&](const hgx_emx_job &J) {
        if (!clusters || J.fast || J.mask) return false;
        if (J.any_size && J.C > HGX_EMX_MAX_CLASSES) return true;
        return lone_ok && (int64_t)J.C * J.a_pad >= (int64_t)512 * 4096;
    };
    // table-lookup problems: while a launch leaves CUs idle (a single sample, a few samples), its big problems run on two or four
    // workgroups each (k_emx<true, true>: the tiles of the two matrix passes shared out, hand-overs through agent-scope loads
    // and stores; 3.45 -> 2.48 -> 2.29 ms for a 1 456-class x 4 500-allele problem).  A full panel keeps one workgroup per
    // problem: its launch is as long as its longest problem either way (6.28 -> 6.07 ms with pairs), and an oversubscribed chip
    // is where a pair may have to wait for a seat.  test switch emx_fast_wg = N: N workgroups for EVERY table-lookup problem.
    const char *fwg_sw = hgx_test_switch("emx_fast_wg");
    static const int n_cu_f = [] {
        int dev = 0;
        hipDeviceProp_t prop;
        return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }();
    int n_big_fast = 0, n_fast = 0;
    for (int i = 0; i < n_jobs; ++i) {
        n_fast += jobs[i].fast ? 1 : 0;
        n_big_fast += (jobs[i].fast && !jobs[i].mask && (int64_t)jobs[i].C * jobs[i].a_pad >= (int64_t)768 * 4096) ? 1 : 0;
    }
    const int wg_for_big = (n_fast + 3 * n_big_fast <= n_cu_f / 2) ? 4 : (n_fast + n_big_fast <= n_cu_f / 2) ? 2 : 1;
    auto fast_wg = [&](const hgx_emx_job &J) -> int {
        if (!clusters || !J.fast || J.mask) return 1;
        if (fwg_sw) return std::max(1, std::min(8, atoi(fwg_sw)));
        return (int64_t)J.C * J.a_pad >= (int64_t)768 * 4096 ? wg_for_big : 1;
    };
    const bool stamps = HGX_LAB_SWITCH("emx_stamps") != nullptr;
    // scratch of every job out of ONE block; jobs beyond the kernel's limits get status 1 without a descriptor
    struct Lay { size_t Rm, Mk, Mr, dv, pos, tmpv, vlen, cls, din, sorted, first, stamps, cl, end; };
    std::vector<Lay> lays;
    std::vector<int> job_of;
    std::vector<size_t> base;
    size_t total = 0, res_cap = 0;
    for (int i = 0; i < n_jobs; ++i) {
        hgx_emx_job &J = jobs[i];
        J.n_iter = 0;
        J.n_classes = 0;
        J.status = 1;
        const int c_max = (J.any_size && !J.fast) ? HGX_EMX_HARD_MAX_CLASSES : HGX_EMX_MAX_CLASSES;      // (the fast arithmetic holds 4 classes per thread)
        if (J.C <= 0 || (!J.mask && J.C > c_max) || J.w64 > 128 || J.a_pad > HGX_EMX_MAX_ALLELES || J.a_pad != 64 * J.w64) continue;
        ARGCHK(J.bits && J.count && J.rank && (J.prob || recs_out) && J.n_out <= J.a_pad);
        const size_t c_alloc = J.mask ? std::min<size_t>((size_t)J.C, 2048) : (size_t)J.C;
        const size_t Cp = (c_alloc + 63) & ~(size_t)63, A1s = (size_t)J.a_pad, A1w = J.mask ? 1 : A1s / 64, Cw = Cp / 64;
        Lay L;
        size_t o = 0;
        L.Rm = o; o += up64(J.mask ? 64 : Cp * A1s / 64 * 8);
        L.Mk = o; o += up64(A1w * Cp * 8);
        L.Mr = o; o += up64(Cw * A1s * 8);
        const size_t n_dict = (size_t)fast_wg(J);             // (a table-lookup cluster: the dicts once per workgroup)
        L.dv = o; o += up64(n_dict * 3 * A1s * 8);
        L.pos = o; o += up64(4 * A1s * 2);
        L.tmpv = o; o += up64(A1s * 8);
        L.vlen = o; o += up64(A1s * 8);
        L.cls = o; o += up64(5 * Cp * 8);
        L.din = o; o += up64(n_dict * 3 * A1s);
        L.sorted = o; o += up64(A1s * 4);
        L.first = o; o += up64(A1s * 4);
        L.stamps = o; o += up64(16 * 8);
        L.cl = o; o += up64(64 + (size_t)(XCW + XAW) * 8);        // cluster control words, gvalid, gin
        L.end = o;
        lays.push_back(L);
        base.push_back(total);
        total += o;
        res_cap += A1s;
        job_of.push_back(i);
    }
    const int n = (int)job_of.size();
    if (n == 0) return HGX_OK;
    {
        // longest first: workgroups are dispatched in block order, and the launch ends when the last one does
        std::vector<int> ord((size_t)n);
        for (int t = 0; t < n; ++t) ord[t] = t;
        std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) {
            const hgx_emx_job &a = jobs[job_of[x]], &b = jobs[job_of[y]];
            return (int64_t)(a.mask ? 1 : a.C) * a.a_pad > (int64_t)(b.mask ? 1 : b.C) * b.a_pad;
        });
        std::vector<Lay> l2((size_t)n);
        std::vector<int> j2((size_t)n);
        std::vector<size_t> b2((size_t)n);
        for (int t = 0; t < n; ++t) { l2[t] = lays[ord[t]]; j2[t] = job_of[ord[t]]; b2[t] = base[ord[t]]; }
        lays.swap(l2); job_of.swap(j2); base.swap(b2);
    }
    // results: [cursor + padding | scal of every job | records], fetched together
    const size_t head = 64 + (size_t)n * XS_N * 8;
    DevBuf b_scr, b_tasks, b_res;
    ALLOC(b_scr, total);
    ALLOC(b_tasks, (size_t)n * sizeof(EmxTask));
    ALLOC(b_res, head + res_cap * sizeof(EmxRes));
    char *scr = b_scr.as<char>(), *resb = b_res.as<char>();
    HIPCHK(hipMemsetAsync(resb, 0, 64, st));
    std::vector<EmxTask> tasks((size_t)n);
    for (int t = 0; t < n; ++t) {
        const hgx_emx_job &J = jobs[job_of[t]];
        const Lay &L = lays[t];
        char *b = scr + base[t];
        EmxTask &T = tasks[t];
        T.B = J.bits; T.count = J.count; T.rank = J.rank; T.len = J.len; T.mask = J.mask;
        T.C = J.C; T.w64 = J.w64; T.a_pad = J.a_pad; T.remove_low = J.remove_low ? 1 : 0;
        T.c_alloc = J.mask ? std::min(J.C, 2048) : J.C;
        T.fast = J.fast ? 1 : 0;
        T.Rm = (uint64_t *)(b + L.Rm); T.Mk = (uint64_t *)(b + L.Mk); T.Mr = (uint64_t *)(b + L.Mr); T.dv = (double *)(b + L.dv); T.pos = (uint16_t *)(b + L.pos);
        T.tmpv = (double *)(b + L.tmpv); T.vlen = (double *)(b + L.vlen); T.cls = (double *)(b + L.cls); T.din = (uint8_t *)(b + L.din);
        T.sorted = (int32_t *)(b + L.sorted); T.first_c = (int32_t *)(b + L.first);
        T.scal = (double *)(resb + 64 + (size_t)t * XS_N * 8);
        T.res = (EmxRes *)(resb + head);
        T.cursor = (unsigned long long *)resb;
        T.stamps = stamps ? (unsigned long long *)(b + L.stamps) : nullptr;
        T.cluster = 0;
        T.cl_first = 0;
        T.cl_spins = 400000;
        T.cl_ctl = (unsigned int *)(b + L.cl);
        T.gvalid = (unsigned long long *)(b + L.cl + 64);
        T.gin = T.gvalid + XCW;
    }
    // Cluster problems run SIDE BY SIDE in one launch: a workgroup of k_emx needs a whole CU (150 KB of LDS), so the clusters of a
    // launch share the chip's CUs -- 64 workgroups for a lone problem (a class tile per SIMD), fewer each when there are several;
    // more cluster problems than a launch can seat go out in further launches.
    std::vector<std::vector<int>> cl_launches;
    {
        int n_cl = 0;
        for (int t = 0; t < n; ++t) n_cl += wants_cluster(jobs[job_of[t]]) ? 1 : 0;
        static const int n_cu = [] {
            int dev = 0;
            hipDeviceProp_t prop;
            return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        }();
        const int seats = std::max(8, n_cu - 16);                       // (leave room for whatever else is resident)
        const char *spin_sw = hgx_test_switch("emx_cluster_spins");
        int first = 0;
        for (int t = 0; t < n; ++t) {
            const hgx_emx_job &J = jobs[job_of[t]];
            if (!wants_cluster(J)) continue;
            const int Cw = (int)(((size_t)J.C + 63) / 64);
            const int per = std::max(2, std::min(64, seats / std::max(1, std::min(n_cl, seats / 2))));
            // (a class tile per SIMD for the huge problems; a mid-size one -- <= 4 096 classes -- gains up to two tiles per workgroup:
            // 1 730 classes x 4 500 alleles: 4 workgroups 9.9 ms per call, 7: 6.9, 14: 5.4, 28: 5.05, 56: 5.0, tools/dropin_profile.py)
            int want = std::max(2, std::min(per, Cw <= 64 ? (Cw + 1) / 2 : (Cw + 3) / 4));
            if (const char *wsw = hgx_test_switch("emx_cluster_wg")) want = std::max(2, std::min(std::min(per, 64), atoi(wsw)));   // (measurement: workgroups per cluster)
            if (cl_launches.empty() || first + want > seats) { cl_launches.push_back({}); first = 0; }
            tasks[t].cluster = want;
            tasks[t].cl_first = first;
            if (spin_sw) tasks[t].cl_spins = std::max(1, atoi(spin_sw));
            first += want;
            cl_launches.back().push_back(t);
            HIPCHK(hipMemsetAsync(scr + base[t] + lays[t].cl, 0, 64 + (size_t)(XCW + XAW) * 8, st));
            g_cluster_jobs.fetch_add(1);
        }
    }
    // the table-lookup launch with clusters: its tasks in launch order, every one with a block range (most of them one block)
    std::vector<EmxTask> ftasks;
    DevBuf b_ftasks, b_fctl;
    int fast_blocks = 0;
    {
        bool any_pair = false;
        for (int t = 0; t < n; ++t) any_pair = any_pair || fast_wg(jobs[job_of[t]]) > 1;
        if (any_pair) {
            const char *spin_sw = hgx_test_switch("emx_cluster_spins");
            for (int t = 0; t < n; ++t) {
                if (!tasks[t].fast) continue;
                EmxTask T = tasks[t];
                T.cluster = fast_wg(jobs[job_of[t]]);
                T.cl_first = fast_blocks;
                if (spin_sw) T.cl_spins = std::max(1, atoi(spin_sw));
                fast_blocks += T.cluster;
                if (T.cluster > 1) g_cluster_jobs.fetch_add(1);
                ftasks.push_back(T);
            }
            ALLOC(b_fctl, ftasks.size() * 64);                          // barrier count + abort flag per task, on a line of its own
            HIPCHK(hipMemsetAsync(b_fctl.p, 0, ftasks.size() * 64, st));
            for (size_t k = 0; k < ftasks.size(); ++k) ftasks[k].cl_ctl = (unsigned int *)(b_fctl.as<char>() + 64 * k);
            ALLOC(b_ftasks, ftasks.size() * sizeof(EmxTask));
            for (size_t off = 0; off < ftasks.size() * sizeof(EmxTask);) {
                const size_t chunk = std::min<size_t>(ftasks.size() * sizeof(EmxTask) - off, 128u << 10);
                { int rc_ = hgx_h2d(b_ftasks.as<char>() + off, (const char *)ftasks.data() + off, chunk, st); if (rc_) return rc_; }
                off += chunk;
            }
        }
    }
    for (size_t off = 0; off < tasks.size() * sizeof(EmxTask);) {      // descriptors through the pinned staging buffer
        const size_t chunk = std::min<size_t>(tasks.size() * sizeof(EmxTask) - off, 128u << 10);
        { int rc_ = hgx_h2d(b_tasks.as<char>() + off, (const char *)tasks.data() + off, chunk, st); if (rc_) return rc_; }
        off += chunk;
    }
    bool any_fast = false, any_exact = false;
    for (const EmxTask &T : tasks) { any_fast = any_fast || T.fast; any_exact = any_exact || !T.fast; }
    bool timing;
    { std::lock_guard<std::mutex> g(g_emx_mu); timing = g_emx_timing; }
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    if (timing) for (auto &e : ev) HIPCHK(hipEventCreate(&e));
    if (any_exact) {
        if (timing) HIPCHK(hipEventRecord(ev[0], st));
        bool any_plain = false;
        for (const EmxTask &T : tasks) any_plain = any_plain || (!T.fast && T.cluster == 0);
        if (any_plain) hipLaunchKernelGGL(k_emx<false>, dim3((unsigned)n), dim3(XB), sizeof(XLds), st, b_tasks.as<EmxTask>(), n);
        for (const std::vector<int> &grp : cl_launches) {
            // (a launch's clusters are consecutive in the task table: the kernel scans tasks[first .. last] for its block range)
            const int lo = grp.front(), hi = grp.back();
            const int blocks = tasks[hi].cl_first + tasks[hi].cluster;
            hipLaunchKernelGGL((k_emx<false, true>), dim3((unsigned)blocks), dim3(XB), sizeof(XLds), st, b_tasks.as<EmxTask>() + lo, hi - lo + 1);
        }
        if (timing) HIPCHK(hipEventRecord(ev[1], st));
    }
    if (any_fast) {
        if (timing) HIPCHK(hipEventRecord(ev[2], st));
        if (!ftasks.empty())
            hipLaunchKernelGGL((k_emx<true, true>), dim3((unsigned)fast_blocks), dim3(XB), sizeof(XLds), st, b_ftasks.as<EmxTask>(), (int)ftasks.size());
        else
            hipLaunchKernelGGL(k_emx<true>, dim3((unsigned)n), dim3(XB), sizeof(XLds), st, b_tasks.as<EmxTask>(), n);
        if (timing) HIPCHK(hipEventRecord(ev[3], st));
    }
    HIPCHK(hipGetLastError());
    // one round trip brings the state words and the first records; a second one the rest of a long result
    const size_t first_recs = std::min<size_t>(res_cap, ((200u << 10) - head % (200u << 10)) / sizeof(EmxRes) + 0);
    const size_t first_bytes = std::min<size_t>(head + first_recs * sizeof(EmxRes), head + res_cap * sizeof(EmxRes));
    // (the host copy is sized by what came back, never by res_cap: tens of MB of fresh pages per call -- above malloc's mmap
    // threshold -- cost more than the EM #2 launch itself)
    std::vector<char> h(first_bytes);
    for (size_t off = 0; off < first_bytes;) {
        const size_t chunk = std::min<size_t>(first_bytes - off, 128u << 10);
        { int rc_ = hgx_d2h(h.data() + off, resb + off, chunk, st); if (rc_) return rc_; }
        off += chunk;
    }
    { int rc_ = hgx_sync(st); if (rc_) return rc_; }
    const unsigned long long n_rec = *(const unsigned long long *)h.data();
    if (n_rec > res_cap) { hgx_set_error("EM result records overflow (%llu > %zu)", n_rec, res_cap); return HGX_EHIP; }
    const size_t all_bytes = head + (size_t)n_rec * sizeof(EmxRes);
    if (all_bytes > first_bytes) {
        h.resize(all_bytes);
        for (size_t off = first_bytes; off < all_bytes;) {
            const size_t chunk = std::min<size_t>(all_bytes - off, 128u << 10);
            { int rc_ = hgx_d2h(h.data() + off, resb + off, chunk, st); if (rc_) return rc_; }
            off += chunk;
        }
        { int rc_ = hgx_sync(st); if (rc_) return rc_; }
    }
    const EmxRes *recs = (const EmxRes *)(h.data() + head);
    if (recs_out) recs_out->assign(recs, recs + n_rec);
    if (timing) {
        std::lock_guard<std::mutex> g(g_emx_mu);
        for (int f = 0; f < 2; ++f) {
            if (!(f ? any_fast : any_exact)) continue;
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, ev[2 * f], ev[2 * f + 1]);
            g_emx_stats[f].ms += ms;
            g_emx_stats[f].launches += 1;
        }
        for (int t = 0; t < n; ++t) {
            const double *sc = (const double *)(h.data() + 64 + (size_t)t * XS_N * 8);
            if ((int)sc[XS_STATUS] == 1) continue;
            EmxStats &s = g_emx_stats[tasks[t].fast ? 1 : 0];
            const long long C = (long long)sc[XS_NCLS], A1 = (long long)sc[XS_A1], apps = (long long)sc[XS_APPS];
            s.jobs += 1;
            s.apps += apps;
            s.bytes += apps * (C * A1 / 8 + 16 * A1 + 16 * C);
        }
        for (auto &e : ev) (void)hipEventDestroy(e);
    }
    for (int t = 0; t < n; ++t) {
        hgx_emx_job &J = jobs[job_of[t]];
        const double *sc = (const double *)(h.data() + 64 + (size_t)t * XS_N * 8);
        J.status = (int32_t)sc[XS_STATUS];
        J.n_iter = (int32_t)sc[XS_ITER];
        J.n_classes = (int32_t)sc[XS_NCLS];
        if (J.status == 1) continue;
        if (recs_out) {
            J.rec_off = J.status == 0 ? (size_t)sc[XS_RES_OFF] : 0;
            J.n_rec = J.status == 0 ? (int32_t)sc[XS_RES_N] : 0;
            continue;
        }
        for (int a = 0; a < J.n_out; ++a) J.prob[a] = -1.0;
        if (J.first) for (int a = 0; a < J.n_out; ++a) J.first[a] = -1;
        if (J.order) for (int a = 0; a < J.n_out; ++a) J.order[a] = -1;
        if (J.status != 0) continue;
        const size_t off = (size_t)sc[XS_RES_OFF], cnt = (size_t)sc[XS_RES_N];
        for (size_t k = 0; k < cnt; ++k) {
            const EmxRes &r = recs[off + k];
            if (r.allele >= 0 && r.allele < J.n_out) {
                J.prob[r.allele] = r.prob;
                if (J.first) J.first[r.allele] = r.first;
                if (J.order) J.order[r.allele] = r.order;
            }
        }
    }
    if (stamps) {
        for (int t = 0; t < std::min(n, 8); ++t) {
            unsigned long long hs[12];
            (void)hipMemcpy(hs, scr + base[t] + lays[t].stamps, 96, hipMemcpyDeviceToHost);
            const double *sc = (const double *)(h.data() + 64 + (size_t)t * XS_N * 8);
            fprintf(stderr, "[k_emx] job %d C %d A1 %d iters %d orders %d: set-up %.1f us | rows %.1f | cols %.1f | order %.1f | normalise %.1f | "
                            "init %.1f | vector steps %.1f || set-up: active alleles %.1f, class rows %.1f, Mk + Mr %.1f, rest %.1f\n", job_of[t], jobs[job_of[t]].C, (int)sc[XS_A1], (int)sc[XS_ITER], (int)sc[XS_ORDERS],
                    (hs[0] + hs[7] + hs[8] + hs[9]) * 0.01, hs[1] * 0.01, hs[2] * 0.01, hs[3] * 0.01, hs[4] * 0.01, hs[5] * 0.01, hs[6] * 0.01, hs[7] * 0.01,
                    hs[8] * 0.01, hs[9] * 0.01, hs[0] * 0.01);
        }
    }
    return HGX_OK;
}
