// hgx_common.hpp -- shared by the device translation units of libhgx (gfx950).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

#include "hgx.h"

extern "C" void hgx_set_error(const char *fmt, ...);
// test hook (hgx.h: hgx_test_switch_set): value of a path-forcing switch, NULL when it is not set (one relaxed load then)
extern "C" const char *hgx_test_switch(const char *name);
// a switch whose value is a comma-separated list ("em_skip" = "emx,tail"): is `token` in it?
static inline bool hgx_switch_has(const char *name, const char *token) {
    const char *v = hgx_test_switch(name);
    if (!v) return false;
    const size_t n = strlen(token);
    for (const char *p = v;;) {
        const char *e = strchr(p, ',');
        const size_t len = e ? (size_t)(e - p) : strlen(p);
        if (len == n && memcmp(p, token, n) == 0) return true;
        if (!e) return false;
        p = e + 1;
    }
}
// switches of comparison / measurement forms that only the lab library (-DHGX_LAB) carries
#ifdef HGX_LAB
#define HGX_LAB_SWITCH(name) hgx_test_switch(name)
#else
#define HGX_LAB_SWITCH(name) ((const char *)nullptr)
#endif

#define HIPCHK(expr)                                                                   \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) {                                                        \
            hgx_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return HGX_EHIP;                                                           \
        }                                                                              \
    } while (0)
#define ARGCHK(cond)                                                                   \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            hgx_set_error("invalid argument: %s (%s:%d)", #cond, __FILE__, __LINE__);  \
            return HGX_EINVAL;                                                         \
        }                                                                              \
    } while (0)

// caching device allocator (hgx_device.hip): hipMalloc/hipFree are far too slow for per-step scratch
void *hgx_pool_alloc(size_t bytes);
// Small host<->device transfers go through a per-thread PINNED staging buffer: a D2H into pageable memory costs an extra
// staging hop + wait inside the runtime (measured 27 us vs 15 us per kernel+copy+sync round trip on MI355X).
// hgx_d2h queues the copy and remembers where the bytes finally belong; hgx_sync = hipStreamSynchronize + delivery.
int hgx_d2h(void *host_dst, const void *dev_src, size_t n, hipStream_t st);
int hgx_h2d(void *dev_dst, const void *host_src, size_t n, hipStream_t st);
int hgx_sync(hipStream_t st);
void hgx_pool_free(void *p);

struct hgx_index {
    int32_t n_alleles, a_pad, n_vars, n_words, w64;
    uint32_t *d_bits;
    uint64_t *d_exon_mask, *d_gene_mask;
    // per-LOCUS pattern tables of the piece x allele kernel (hgx_device.hip, k_piece_compat_pat), made on first use from d_bits:
    // the distinct 32-bit values of every variant word over the alleles (~100 of 7 000 at HLA-A) and every allele's value id
    int pat_state = 0;                   // 0 = not made yet, 1 = ready, 2 = this locus does not fit (a word with too many values)
    uint16_t *d_pid = nullptr;           // [n_words][a_pad]
    uint32_t *d_vals = nullptr;          // [n_words][HGX_PAT_D]
    int32_t *d_nval = nullptr;           // [n_words]
    int pat_dmax = 0;
    std::mutex *pat_mu = nullptr;
};
#define HGX_PAT_D 512

struct hgx_classes {
    int32_t n_classes, a_pad, w64, c64;
    uint64_t *d_bits;        // [n_classes][w64]
    int64_t *d_count;        // [n_classes]
    int64_t *d_first_row;    // [n_classes]
    uint64_t *d_bitsT;       // lazily built [a_pad][c64]
    uint64_t *d_prow = nullptr, *d_pcol = nullptr;   // lazily built MFMA-operand orders of the compact matrices (hgx_em.hip)
    // lazily built by the EM (hgx_em.hip): the matrices restricted to the alleles that occur in some class
    int32_t n_act = -1, a1p = 0;                     // active alleles, padded to a multiple of 512
    int32_t *d_act = nullptr;                        // [n_act] ascending allele ids
    uint64_t *d_bitsC = nullptr;                     // [c64 * 64][a1p / 64]  (rows >= n_classes are zero)
    uint64_t *d_bitsTC = nullptr;                    // [a1p][c64]
    int32_t *h_act = nullptr;                        // host copy of d_act (new[])
    int32_t *h_rank = nullptr;                       // name order of the alleles, if the caller supplied it (new[]; hgx_classes_set_allele_rank)
    uint64_t *d_wrow = nullptr, *d_wcol = nullptr;   // word-transposed compact matrices [a1p/64][c64*64], [c64][a1p]
    void *d_setup0 = nullptr, *d_setup1 = nullptr;   // small tables the set-up kernels read (kept so that no sync is needed)
    hipStream_t made_on = nullptr;                   // stream the kernels that fill this class set were queued on
    hipEvent_t ready = nullptr;                      // recorded behind them: consumers on OTHER streams wait for it (device side)
    void *d_keep[12] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                        nullptr, nullptr, nullptr, nullptr};   // dedup scratch still read by queued kernels
};

// a piece batch resident in HBM: uploaded from a host batch (hgx_dbatch_create) or built there by the device front end
// (hgx_front.hip), which also leaves the pileup tables behind
struct hgx_dbatch {
    int32_t n_pieces = 0, n_pairs = 0, n_reads = 0;
    int64_t n_refs = 0, n_mask_u32 = 0, sum_piece_words = 0, n_gene_refs = 0;
    hgx_piece *d_pieces = nullptr;
    uint32_t *d_masks = nullptr;
    int32_t *d_pair_off = nullptr;
    uint32_t *d_pair_ref = nullptr;
    int32_t n_ref = 0;                   // pileup tables (device front end only): counts[n_ref][6], nt_set[n_ref]
    uint32_t *d_counts = nullptr;
    uint8_t *d_nt_set = nullptr;
    std::vector<std::string> trace;      // hgx_parse_opts.keep_trace: one line per kept record (hgx_batch_trace_text after hgx_dbatch_to_host)
};

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) hgx_pool_free(p); }
    int alloc(size_t n) { p = hgx_pool_alloc(n); return p ? 0 : -1; }
    template <class T> T *as() { return (T *)p; }
};
#define ALLOC(buf, bytes)                                              \
    do {                                                               \
        if ((buf).alloc(bytes)) {                                      \
            hgx_set_error("device allocation of %zu bytes failed", (size_t)(bytes));   \
            return HGX_ENOMEM;                                         \
        }                                                              \
    } while (0)

// a consumer of class set `c` about to queue work on `st`: order it behind the kernels that are still filling the set
static inline void hgx_classes_order_after(const hgx_classes *c, hipStream_t st) {
    if (c && c->ready && st != c->made_on) (void)hipStreamWaitEvent(st, c->ready, 0);
}

// `body` (hipFuncSetAttribute calls: per-DEVICE state) runs once per device at this call site, under a lock: a second device, or a
// second host thread arriving during the first call, never launches without the attribute
#define HGX_ONCE_PER_DEVICE(body)                                            \
    do {                                                                     \
        static std::mutex once_mu_;                                          \
        static uint64_t once_done_ = 0;                                      \
        int once_dev_ = 0;                                                   \
        HIPCHK(hipGetDevice(&once_dev_));                                    \
        std::lock_guard<std::mutex> once_g_(once_mu_);                       \
        if (!((once_done_ >> (once_dev_ & 63)) & 1ull)) {                    \
            body;                                                            \
            once_done_ |= 1ull << (once_dev_ & 63);                          \
        }                                                                    \
    } while (0)

static inline unsigned nblk(long n, int per) { return (unsigned)((n + per - 1) / per); }

int hgx_ensure_transposed(hgx_classes *c, hipStream_t st);
int hgx_pair_classes_sel(const hgx_index *ix, const uint64_t *compat, const int32_t *pair_off, const uint32_t *refs,
                         const int64_t *sel, int32_t n_pairs, uint64_t *eb, uint64_t *gb, uint64_t *eh,
                         uint64_t *gh, hipStream_t st);

// segment-aware forms for many tasks in one launch chain (hgx_dedup.hip; used by hgx_type_many): pair_seg / row_seg = task of a pair / row
struct hgx_groups;
int hgx_group_pairs_seg(hgx_groups **out, const int32_t *pair_off, const uint32_t *refs, int32_t n_pairs, int32_t level,
                        const uint32_t *pair_seg, void *stream);
int hgx_level_classes_grouped_seg(hgx_classes **out, const hgx_index *ix, const uint64_t *compat, const int32_t *pair_off,
                                  const uint32_t *refs, hgx_groups *g, uint64_t *rows_scratch, uint64_t *hash_scratch,
                                  const uint32_t *pair_seg, void *stream);
int hgx_dedup_classes_seg(hgx_classes **out, const uint64_t *rows, const uint64_t *row_hash, int64_t n_rows, int32_t a_pad,
                          const uint32_t *row_seg, void *stream);

// fused gene-level form (hgx_device.hip): rows are claimed / verified against their class' representative by the wavefront
// that computed them; hgx_dedup.hip (hgx_pair_classes_dedup) owns the table and turns it into a class set
int hgx_pair_classes_fused_launch(const hgx_index *ix, const uint64_t *compat, const int32_t *pair_off, const uint32_t *refs,
                                  int32_t n_pairs, int32_t level, unsigned long long *keys, uint32_t tmask, uint32_t *rep,
                                  uint32_t *slot_of, int *bad, uint64_t *rows, hipStream_t st);

int hgx_pair_classes_dedup_ev(hgx_classes **out, const hgx_index *ix, const uint64_t *compat, const int32_t *pair_off, const uint32_t *refs,
                              int32_t n_pairs, int32_t level, uint64_t *rows_scratch, void *stream, void *ev_begin, void *ev_end);

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
#define HGX_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull

__device__ __forceinline__ uint64_t mix64(uint64_t x) {   // splitmix64 finaliser
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}
__device__ __forceinline__ uint64_t word_hash(uint64_t w, int idx) {
    return w ? mix64(w + 0x9e3779b97f4a7c15ull * (uint64_t)(idx + 1)) : 0ull;
}
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl_xor(lo, m, 64);
    hi = __shfl_xor(hi, m, 64);
    return ((uint64_t)hi << 32) | lo;
}
// Wave-wide butterfly reductions (every lane ends up with the result; fixed association: lane ^ 32, ^ 16, ^ 8, ^ 4, ^ 2, ^ 1 --
// bit-identical to the __shfl_xor form they replace, checked by tools/wave_reduce_test.hip).  __shfl_xor is a ds_bpermute per
// 32 bits (an LDS-crossbar round trip, ~100+ cycles each, twelve in a row for one FP64 sum); the same data movement with
// register-file operations: v_permlane32_swap / v_permlane16_swap (gfx950) for the two cross-row steps, DPP row_ror:8 and
// row_ror:4 (== lane ^ 4 once lanes i and i ^ 8 agree, which they do after the ^ 8 step of a commutative reduction) and quad_perm
// for the rest.  The single-wavefront EM kernels spend most of their time in these (nine reductions per iteration).
template <int CTRL> __device__ __forceinline__ uint64_t dpp_u64(uint64_t b) {
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, false);
    return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}
__device__ __forceinline__ uint64_t other_half_u64(uint64_t b) {       // value of lane ^ 32
    auto lo = __builtin_amdgcn_permlane32_swap((int)b, (int)b, false, false);
    auto hi = __builtin_amdgcn_permlane32_swap((int)(b >> 32), (int)(b >> 32), false, false);
    const bool up = (threadIdx.x & 32) != 0;
    return ((uint64_t)(uint32_t)(up ? hi[0] : hi[1]) << 32) | (uint32_t)(up ? lo[0] : lo[1]);
}
__device__ __forceinline__ uint64_t other_row_u64(uint64_t b) {        // value of lane ^ 16
    auto lo = __builtin_amdgcn_permlane16_swap((int)b, (int)b, false, false);
    auto hi = __builtin_amdgcn_permlane16_swap((int)(b >> 32), (int)(b >> 32), false, false);
    const bool up = (threadIdx.x & 16) != 0;
    return ((uint64_t)(uint32_t)(up ? hi[0] : hi[1]) << 32) | (uint32_t)(up ? lo[0] : lo[1]);
}
// op must be commutative bit for bit (a + b, max of non-negative values): see the row_ror:4 remark above
template <class T, class Op> __device__ __forceinline__ T wave_butterfly(T v, Op op) {
    static_assert(sizeof(T) == 8, "64-bit values");
    auto bits = [](T x) { uint64_t b; __builtin_memcpy(&b, &x, 8); return b; };
    auto val = [](uint64_t b) { T x; __builtin_memcpy(&x, &b, 8); return x; };
    v = op(v, val(other_half_u64(bits(v))));
    v = op(v, val(other_row_u64(bits(v))));
    v = op(v, val(dpp_u64<0x128>(bits(v))));       // row_ror:8
    v = op(v, val(dpp_u64<0x124>(bits(v))));       // row_ror:4
    v = op(v, val(dpp_u64<0x4E>(bits(v))));        // quad_perm [2,3,0,1]
    v = op(v, val(dpp_u64<0xB1>(bits(v))));        // quad_perm [1,0,3,2]
    return v;
}
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
    return wave_butterfly(v, [](uint64_t a, uint64_t b) { return a + b; });
}
__device__ __forceinline__ double wave_sum_f64(double v) {
    return wave_butterfly(v, [](double a, double b) { return a + b; });
}
__device__ __forceinline__ double wave_max_nonneg_f64(double v) {       // operands >= 0, no NaN: fmax is then commutative bit for bit
    return wave_butterfly(v, [](double a, double b) { return fmax(a, b); });
}
// 64 x 64 bit transpose across a wavefront: lane r holds row r; afterwards lane c holds column c (bit r = old bit c of lane r).
// Six block-swap stages (Hacker's Delight 7-3), the partner lane's word fetched through the register file: v_permlane32_swap,
// v_permlane16_swap, DPP row_ror:8, row_half_mirror + reversed quads (= lane ^ 4), quad_perm -- ~90 instructions instead of the
// 64 ballots + selects (~500) the set-up transposes of the EM used.
__device__ __forceinline__ uint64_t wave_transpose64(uint64_t x) {
    const int l = threadIdx.x & 63;
    auto stage = [&](uint64_t t, int j, uint64_t m) {
        return (l & j) ? ((x & ~m) | ((t >> j) & m)) : ((x & m) | ((t & m) << j));
    };
    x = stage(other_half_u64(x), 32, 0x00000000FFFFFFFFull);
    x = stage(other_row_u64(x), 16, 0x0000FFFF0000FFFFull);
    x = stage(dpp_u64<0x128>(x), 8, 0x00FF00FF00FF00FFull);                  // row_ror:8 == lane ^ 8
    x = stage(dpp_u64<0x1B>(dpp_u64<0x141>(x)), 4, 0x0F0F0F0F0F0F0F0Full);   // half-row mirror, then reversed quads == lane ^ 4
    x = stage(dpp_u64<0x4E>(x), 2, 0x3333333333333333ull);                   // quad_perm [2,3,0,1]
    x = stage(dpp_u64<0xB1>(x), 1, 0x5555555555555555ull);                   // quad_perm [1,0,3,2]
    return x;
}
__device__ __forceinline__ uint64_t finish_hash(uint64_t h, bool nonzero) {
    if (!nonzero) return HGX_EMPTY_KEY;
    return h == HGX_EMPTY_KEY ? HGX_EMPTY_KEY - 1 : h;
}

