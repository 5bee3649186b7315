// hgx_device.hip -- gfx950 (MI355X, CDNA4) kernels + C-ABI for the device side of libhgx.
//
// Data layout in HBM:
//   link bits     uint32 [n_words][a_pad]   word-major: one wavefront reads 256 contiguous bytes per
//                                           32-variant word (lane = allele)
//   piece compat  uint64 [n_pieces][w64]    allele bitset per DISTINCT piece        (w64 = a_pad/64)
//   class rows    uint64 [n_pairs][w64]     per pair and level
//   class matrix  uint64 [C][w64] + int64 count[C]; transposed copy uint64 [a_pad][c64] for the EM
// Everything is wave64; no MFMA (the work is bit/byte logic and FP64 mat-vec over a 0/1 matrix).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

#include "hgx.h"

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

extern "C" void hgx_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char *hgx_last_error(void) { return g_err; }
extern "C" int hgx_version(void) { return 100; }

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

extern "C" int hgx_device_count(int *n) { HIPCHK(hipGetDeviceCount(n)); return HGX_OK; }
extern "C" int hgx_set_device(int dev) { HIPCHK(hipSetDevice(dev)); return HGX_OK; }
extern "C" int hgx_dev_alloc(void **p, size_t bytes) {
    ARGCHK(p != nullptr);
    HIPCHK(hipMalloc(p, bytes ? bytes : 8));
    return HGX_OK;
}
extern "C" int hgx_dev_free(void *p) { if (p) HIPCHK(hipFree(p)); return HGX_OK; }
extern "C" int hgx_memcpy_h2d(void *d, const void *s, size_t n, void *st) {
    HIPCHK(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, (hipStream_t)st));
    HIPCHK(hipStreamSynchronize((hipStream_t)st));
    return HGX_OK;
}
extern "C" int hgx_memcpy_d2h(void *d, const void *s, size_t n, void *st) {
    HIPCHK(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, (hipStream_t)st));
    HIPCHK(hipStreamSynchronize((hipStream_t)st));
    return HGX_OK;
}
extern "C" int hgx_memset(void *d, int v, size_t n, void *st) {
    HIPCHK(hipMemsetAsync(d, v, n, (hipStream_t)st));
    return HGX_OK;
}
extern "C" int hgx_stream_sync(void *st) { HIPCHK(hipStreamSynchronize((hipStream_t)st)); return HGX_OK; }
extern "C" int hgx_event_create(void **ev) {
    ARGCHK(ev != nullptr);
    hipEvent_t e;
    HIPCHK(hipEventCreate(&e));
    *ev = (void *)e;
    return HGX_OK;
}
extern "C" int hgx_event_destroy(void *ev) { if (ev) HIPCHK(hipEventDestroy((hipEvent_t)ev)); return HGX_OK; }
extern "C" int hgx_event_record(void *ev, void *st) { HIPCHK(hipEventRecord((hipEvent_t)ev, (hipStream_t)st)); return HGX_OK; }
extern "C" int hgx_event_elapsed_ms(void *a, void *b, float *ms) {
    ARGCHK(a && b && ms);
    HIPCHK(hipEventSynchronize((hipEvent_t)b));
    HIPCHK(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
    return HGX_OK;
}

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
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_u64(v, m);
    return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ uint64_t finish_hash(uint64_t h, bool nonzero) {
    if (!nonzero) return HGX_EMPTY_KEY;
    return h == HGX_EMPTY_KEY ? HGX_EMPTY_KEY - 1 : h;
}

// ------------------------------------------------------------------------------------------------
// 8a-0 index
// ------------------------------------------------------------------------------------------------
struct hgx_index {
    int32_t n_alleles, a_pad, n_vars, n_words, w64;
    uint32_t *d_bits;
    uint64_t *d_exon_mask, *d_gene_mask;
};

extern "C" int32_t hgx_a_pad(int32_t n) { return (n + 255) / 256 * 256; }

extern "C" int hgx_index_create(hgx_index **out, int32_t n_alleles, int32_t n_vars, const uint32_t *bits,
                                const uint64_t *exon_mask, const uint64_t *gene_mask) {
    ARGCHK(out && n_alleles > 0 && n_vars >= 0 && bits && exon_mask && gene_mask);
    ARGCHK(n_vars <= 65535 * 32);
    hgx_index *ix = new hgx_index();
    ix->n_alleles = n_alleles;
    ix->a_pad = hgx_a_pad(n_alleles);
    ix->n_vars = n_vars;
    ix->n_words = std::max(1, (n_vars + 31) / 32);
    ix->w64 = ix->a_pad / 64;
    size_t nb = (size_t)ix->n_words * ix->a_pad * sizeof(uint32_t);
    HIPCHK(hipMalloc((void **)&ix->d_bits, nb));
    HIPCHK(hipMalloc((void **)&ix->d_exon_mask, ix->w64 * 8));
    HIPCHK(hipMalloc((void **)&ix->d_gene_mask, ix->w64 * 8));
    HIPCHK(hipMemcpy(ix->d_bits, bits, nb, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ix->d_exon_mask, exon_mask, ix->w64 * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ix->d_gene_mask, gene_mask, ix->w64 * 8, hipMemcpyHostToDevice));
    *out = ix;
    return HGX_OK;
}
extern "C" int hgx_index_destroy(hgx_index *ix) {
    if (!ix) return HGX_OK;
    (void)hipFree(ix->d_bits); (void)hipFree(ix->d_exon_mask); (void)hipFree(ix->d_gene_mask);
    delete ix;
    return HGX_OK;
}
extern "C" int hgx_index_dims(const hgx_index *ix, int32_t *na, int32_t *ap, int32_t *nv, int32_t *nw) {
    ARGCHK(ix);
    if (na) *na = ix->n_alleles;
    if (ap) *ap = ix->a_pad;
    if (nv) *nv = ix->n_vars;
    if (nw) *nw = ix->n_words;
    return HGX_OK;
}
extern "C" int hgx_index_device_bits(const hgx_index *ix, void **p, size_t *bytes) {
    ARGCHK(ix && p && bytes);
    *p = ix->d_bits;
    *bytes = (size_t)ix->n_words * ix->a_pad * sizeof(uint32_t);
    return HGX_OK;
}

// ------------------------------------------------------------------------------------------------
// 8a-5 stage 1: piece x allele compatibility.  compat(a) <=> AND_i ((bits[lo+i][a] & MP_i) == P_i)
// One wavefront per (piece, 16 x 64 alleles): lane = allele, masks are wave-uniform (SGPRs), each
// word row is a 256-byte coalesced load, the 64 verdicts leave as one __ballot word.
// ------------------------------------------------------------------------------------------------
#define PC_GROUPS 16
__global__ __launch_bounds__(256) void k_piece_compat(const uint32_t *__restrict__ bits, int a_pad,
                                                      const hgx_piece *__restrict__ pieces,
                                                      const uint32_t *__restrict__ masks, int n_pieces,
                                                      uint64_t *__restrict__ compat, int w64, int chunks) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int piece = (int)(wave / chunks);
    const int chunk = (int)(wave % chunks);
    if (piece >= n_pieces) return;
    const hgx_piece pc = pieces[piece];
    const int lo = __builtin_amdgcn_readfirstlane((int)pc.lo_word);
    const int nw = __builtin_amdgcn_readfirstlane((int)pc.n_words);
    const uint32_t *m = masks + __builtin_amdgcn_readfirstlane((int)pc.mask_off);
    uint64_t mine = 0;
    const int g_end = min(PC_GROUPS, w64 - chunk * PC_GROUPS);
    for (int g = 0; g < g_end; ++g) {
        const int a = (chunk * PC_GROUPS + g) * 64 + lane;
        const uint32_t *col = bits + (size_t)lo * a_pad + a;
        bool ok = true;
        for (int i = 0; i < nw; ++i) {
            const uint32_t r = col[(size_t)i * a_pad];
            ok = ok && ((r & m[2 * i]) == m[2 * i + 1]);
        }
        const uint64_t b = __ballot(ok);
        if (lane == g) mine = b;
    }
    if (lane < g_end) compat[(size_t)piece * w64 + chunk * PC_GROUPS + lane] = mine;
}

extern "C" int hgx_piece_compat(const hgx_index *ix, const hgx_piece *pieces, const uint32_t *masks, int32_t n_pieces,
                                uint64_t *compat, void *stream) {
    ARGCHK(ix && n_pieces >= 0);
    if (n_pieces == 0) return HGX_OK;
    ARGCHK(pieces && masks && compat);
    const int chunks = (ix->w64 + PC_GROUPS - 1) / PC_GROUPS;
    const long waves = (long)n_pieces * chunks;
    const long blocks = (waves + 3) / 4;
    hipLaunchKernelGGL(k_piece_compat, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, ix->d_bits, ix->a_pad,
                       pieces, masks, n_pieces, compat, ix->w64, chunks);
    HIPCHK(hipGetLastError());
    return HGX_OK;
}

// ------------------------------------------------------------------------------------------------
// 8a-5/6 stage 2: one wavefront per pair.  Per level the per-allele counts live as bit-sliced
// counters (NP planes of 64-allele words, lane = word), a ref adds one bit per allele by ripple
// carry, and the arg-max set falls out of a top-down plane scan with a wave-wide "any":
//     cand = level mask;  for k = NP-1 .. 0:  t = cand & plane[k];  if any(t) cand = t
// which leaves exactly {a : count[a] == max count} -- add_stat's class (core:1177-1190) -- without
// ever materialising a count.
// ------------------------------------------------------------------------------------------------
#define NP 8
template <int KW>
__device__ __forceinline__ void class_for_level(const uint64_t *__restrict__ compat, int w64, const uint32_t *__restrict__ refs,
                                                int r0, int r1, uint32_t level, const uint64_t *__restrict__ mask,
                                                uint64_t *__restrict__ out_row, uint64_t *__restrict__ out_hash, int lane) {
    uint64_t plane[NP][KW];
#pragma unroll
    for (int k = 0; k < NP; ++k)
#pragma unroll
        for (int s = 0; s < KW; ++s) plane[k][s] = 0;
    for (int r = r0; r < r1; ++r) {
        const uint32_t ref = __builtin_amdgcn_readfirstlane(refs[r]);
        if ((ref >> 31) != level) continue;
        const uint64_t *row = compat + (size_t)(ref & 0x7fffffffu) * w64;
        uint64_t carry[KW];
#pragma unroll
        for (int s = 0; s < KW; ++s) carry[s] = (lane + 64 * s < w64) ? row[lane + 64 * s] : 0ull;
#pragma unroll
        for (int k = 0; k < NP; ++k)
#pragma unroll
            for (int s = 0; s < KW; ++s) {
                const uint64_t t = plane[k][s] & carry[s];
                plane[k][s] ^= carry[s];
                carry[s] = t;
            }
    }
    uint64_t cand[KW];
#pragma unroll
    for (int s = 0; s < KW; ++s) cand[s] = (lane + 64 * s < w64) ? mask[lane + 64 * s] : 0ull;
#pragma unroll
    for (int k = NP - 1; k >= 0; --k) {
        bool nz = false;
#pragma unroll
        for (int s = 0; s < KW; ++s) nz = nz || ((cand[s] & plane[k][s]) != 0);
        if (__any(nz)) {
#pragma unroll
            for (int s = 0; s < KW; ++s) cand[s] &= plane[k][s];
        }
    }
    uint64_t h = 0;
    bool nz = false;
#pragma unroll
    for (int s = 0; s < KW; ++s) {
        const int w = lane + 64 * s;
        if (w < w64) {
            if (out_row) out_row[w] = cand[s];
            h += word_hash(cand[s], w);
            nz = nz || cand[s] != 0;
        }
    }
    if (out_hash) {
        h = wave_sum_u64(h);
        const bool any_nz = __any(nz);
        if (lane == 0) *out_hash = finish_hash(h, any_nz);
    }
}

template <int KW>
__global__ __launch_bounds__(256) void k_pair_classes(const uint64_t *__restrict__ compat, int w64,
                                                      const int32_t *__restrict__ pair_off, const uint32_t *__restrict__ refs,
                                                      int n_pairs, const uint64_t *__restrict__ exon_mask,
                                                      const uint64_t *__restrict__ gene_mask, uint64_t *__restrict__ exon_bits,
                                                      uint64_t *__restrict__ gene_bits, uint64_t *__restrict__ exon_hash,
                                                      uint64_t *__restrict__ gene_hash) {
    const int lane = threadIdx.x & 63;
    const long pair = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (pair >= n_pairs) return;
    const int r0 = __builtin_amdgcn_readfirstlane(pair_off[pair]);
    const int r1 = __builtin_amdgcn_readfirstlane(pair_off[pair + 1]);
    if (exon_bits || exon_hash)
        class_for_level<KW>(compat, w64, refs, r0, r1, 0u, exon_mask, exon_bits ? exon_bits + (size_t)pair * w64 : nullptr,
                            exon_hash ? exon_hash + pair : nullptr, lane);
    if (gene_bits || gene_hash)
        class_for_level<KW>(compat, w64, refs, r0, r1, 1u, gene_mask, gene_bits ? gene_bits + (size_t)pair * w64 : nullptr,
                            gene_hash ? gene_hash + pair : nullptr, lane);
}

extern "C" int hgx_pair_classes(const hgx_index *ix, const uint64_t *compat, const int32_t *pair_off, const uint32_t *refs,
                                int32_t n_pairs, uint64_t *eb, uint64_t *gb, uint64_t *eh, uint64_t *gh, void *stream) {
    ARGCHK(ix && n_pairs >= 0);
    if (n_pairs == 0) return HGX_OK;
    ARGCHK(compat && pair_off && refs);
    const long blocks = ((long)n_pairs + 3) / 4;
    const int kw = (ix->w64 + 63) / 64;
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_PC(KW_)                                                                                             \
    hipLaunchKernelGGL(k_pair_classes<KW_>, dim3((unsigned)blocks), dim3(256), 0, st, compat, ix->w64, pair_off, refs, \
                       n_pairs, ix->d_exon_mask, ix->d_gene_mask, eb, gb, eh, gh)
    if (kw <= 1) LAUNCH_PC(1);
    else if (kw <= 2) LAUNCH_PC(2);
    else if (kw <= 4) LAUNCH_PC(4);
    else if (kw <= 8) LAUNCH_PC(8);
    else {
        hgx_set_error("more than 32768 alleles per locus are not supported (a_pad=%d)", ix->a_pad);
        return HGX_EINVAL;
    }
#undef LAUNCH_PC
    HIPCHK(hipGetLastError());
    return HGX_OK;
}

extern "C" int hgx_score_pairs(const hgx_index *ix, const hgx_piece *pieces, const uint32_t *masks, int32_t n_pieces,
                               const int32_t *pair_off, const uint32_t *refs, int32_t n_pairs, uint64_t *compat,
                               uint64_t *eb, uint64_t *gb, uint64_t *eh, uint64_t *gh, void *stream) {
    int rc = hgx_piece_compat(ix, pieces, masks, n_pieces, compat, stream);
    if (rc) return rc;
    return hgx_pair_classes(ix, compat, pair_off, refs, n_pairs, eb, gb, eh, gh, stream);
}

// ------------------------------------------------------------------------------------------------
// 8a-7 class dedup: hash (optional AND mask) -> radix sort -> run heads -> exact verify ->
// first-seen order -> gather.
// ------------------------------------------------------------------------------------------------
struct hgx_classes {
    int32_t n_classes, a_pad, w64, c64;
    uint64_t *d_bits;        // [n_classes][w64]
    int64_t *d_count;        // [n_classes]
    int64_t *d_first_row;    // [n_classes]
    uint64_t *d_bitsT;       // lazily built [a_pad][c64]
};

// one wavefront per row: hash of (row & mask)
__global__ __launch_bounds__(256) void k_hash_rows(const uint64_t *__restrict__ rows, long n_rows, int w64,
                                                   const uint64_t *__restrict__ mask, uint64_t *__restrict__ hash) {
    const int lane = threadIdx.x & 63;
    const long row = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (row >= n_rows) return;
    uint64_t h = 0;
    bool nz = false;
    for (int w = lane; w < w64; w += 64) {
        uint64_t x = rows[(size_t)row * w64 + w];
        if (mask) x &= mask[w];
        h += word_hash(x, w);
        nz = nz || x != 0;
    }
    h = wave_sum_u64(h);
    const bool any_nz = __any(nz);
    if (lane == 0) hash[row] = finish_hash(h, any_nz);
}

__global__ void k_iota(uint32_t *v, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (uint32_t)i;
}

// head[i] = 1 if sorted key i starts a run of a non-empty key
// (empty rows carry the largest key and sort last: *n_valid = number of non-empty rows)
__global__ void k_heads(const uint64_t *__restrict__ key, long n, uint32_t *__restrict__ head, uint32_t *__restrict__ n_valid) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = key[i];
    head[i] = (k != HGX_EMPTY_KEY && (i == 0 || key[i - 1] != k)) ? 1u : 0u;
    if (k != HGX_EMPTY_KEY && (i == n - 1 || key[i + 1] == HGX_EMPTY_KEY)) *n_valid = (uint32_t)(i + 1);
}

// for every run head: remember where the run starts and which original row is its first member
__global__ void k_run_starts(const uint32_t *__restrict__ head, const uint32_t *__restrict__ cls, const uint32_t *__restrict__ idx,
                             long n, uint32_t *__restrict__ run_start, uint32_t *__restrict__ run_first) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !head[i]) return;
    run_start[cls[i]] = (uint32_t)i;
    run_first[cls[i]] = idx[i];   // stable sort => smallest original row of the run
}

// run weight = sum of member weights (prefix sums) ; unit weights => run length
__global__ void k_run_counts(const uint32_t *__restrict__ run_start, int n_runs, long n_valid, const int64_t *__restrict__ wsum,
                             int64_t *__restrict__ run_count) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_runs) return;
    const long b = run_start[r];
    const long e = (r + 1 < n_runs) ? (long)run_start[r + 1] : n_valid;
    if (wsum) run_count[r] = wsum[e - 1] - (b ? wsum[b - 1] : 0);
    else run_count[r] = e - b;
}

__global__ void k_gather_weights(const int64_t *__restrict__ w, const uint32_t *__restrict__ idx, long n, int64_t *__restrict__ out) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = w[idx[i]];
}

// exact check: every member row equals its run's first row (under the mask)
__global__ __launch_bounds__(256) void k_verify(const uint64_t *__restrict__ rows, int w64, const uint64_t *__restrict__ mask,
                                                const uint32_t *__restrict__ idx, const uint32_t *__restrict__ cls,
                                                const uint32_t *__restrict__ head, const uint32_t *__restrict__ run_first,
                                                long n_valid, int *__restrict__ bad) {
    const int lane = threadIdx.x & 63;
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= n_valid || head[i]) return;
    const uint64_t *a = rows + (size_t)idx[i] * w64;
    const uint64_t *b = rows + (size_t)run_first[cls[i] - 1] * w64;   // non-head: exclusive scan counts its own head
    bool diff = false;
    for (int w = lane; w < w64; w += 64) {
        uint64_t x = a[w], y = b[w];
        if (mask) { x &= mask[w]; y &= mask[w]; }
        diff = diff || x != y;
    }
    if (__any(diff) && lane == 0) atomicOr(bad, 1);
}

// out[c] = rows[first_row[c]] & mask
__global__ __launch_bounds__(256) void k_gather_rows(const uint64_t *__restrict__ rows, int w64, const uint64_t *__restrict__ mask,
                                                     const uint32_t *__restrict__ first_sorted, const uint32_t *__restrict__ run_sorted,
                                                     const int64_t *__restrict__ run_count, int n_classes,
                                                     uint64_t *__restrict__ out_bits, int64_t *__restrict__ out_count,
                                                     int64_t *__restrict__ out_first) {
    const int lane = threadIdx.x & 63;
    const long c = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (c >= n_classes) return;
    const uint64_t *src = rows + (size_t)first_sorted[c] * w64;
    for (int w = lane; w < w64; w += 64) {
        uint64_t x = src[w];
        if (mask) x &= mask[w];
        out_bits[(size_t)c * w64 + w] = x;
    }
    if (lane == 0) {
        out_count[c] = run_count[run_sorted[c]];
        out_first[c] = first_sorted[c];
    }
}

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n) { return hipMalloc(&p, n ? n : 8) == hipSuccess ? 0 : -1; }
    template <class T> T *as() { return (T *)p; }
};
#define ALLOC(buf, bytes)                                              \
    do {                                                               \
        if ((buf).alloc(bytes)) {                                      \
            hgx_set_error("hipMalloc(%zu) failed", (size_t)(bytes));   \
            return HGX_ENOMEM;                                         \
        }                                                              \
    } while (0)

static inline unsigned nblk(long n, int per) { return (unsigned)((n + per - 1) / per); }

extern "C" int hgx_dedup_classes(hgx_classes **out, const uint64_t *rows, const uint64_t *row_hash, const int64_t *row_weight,
                                 int64_t n_rows, int32_t a_pad, const uint64_t *and_mask, void *stream) {
    ARGCHK(out && n_rows >= 0 && a_pad > 0 && a_pad % 64 == 0);
    ARGCHK(n_rows < (1ll << 31));
    hipStream_t st = (hipStream_t)stream;
    const int w64 = a_pad / 64;
    hgx_classes *cl = new hgx_classes();
    cl->a_pad = a_pad; cl->w64 = w64; cl->n_classes = 0; cl->c64 = 0;
    cl->d_bits = nullptr; cl->d_count = nullptr; cl->d_first_row = nullptr; cl->d_bitsT = nullptr;
    *out = cl;
    if (n_rows == 0) return HGX_OK;
    ARGCHK(rows);
    const long n = n_rows;
    DevBuf b_hash, b_key, b_idx0, b_idx, b_head, b_cls, b_tmp, b_bad;
    const uint64_t *keys_in = row_hash;
    if (!row_hash || and_mask) {   // hashes of masked rows must be recomputed
        ALLOC(b_hash, n * 8);
        hipLaunchKernelGGL(k_hash_rows, dim3(nblk(n, 4)), dim3(256), 0, st, rows, n, w64, and_mask, b_hash.as<uint64_t>());
        keys_in = b_hash.as<uint64_t>();
    }
    ALLOC(b_key, n * 8); ALLOC(b_idx0, n * 4); ALLOC(b_idx, n * 4); ALLOC(b_head, n * 4); ALLOC(b_cls, n * 4); ALLOC(b_bad, 8);
    hipLaunchKernelGGL(k_iota, dim3(nblk(n, 256)), dim3(256), 0, st, b_idx0.as<uint32_t>(), n);
    size_t tmp_bytes = 0, t2 = 0;
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys_in, b_key.as<uint64_t>(), b_idx0.as<uint32_t>(),
                                              b_idx.as<uint32_t>(), (int)n, 0, 64, st));
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, t2, b_head.as<uint32_t>(), b_cls.as<uint32_t>(), (int)n, st));
    tmp_bytes = std::max(tmp_bytes, t2);
    HIPCHK(hipcub::DeviceScan::InclusiveSum(nullptr, t2, (int64_t *)nullptr, (int64_t *)nullptr, (int)n, st));
    tmp_bytes = std::max(tmp_bytes, t2);
    ALLOC(b_tmp, tmp_bytes);
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(b_tmp.p, tmp_bytes, keys_in, b_key.as<uint64_t>(), b_idx0.as<uint32_t>(),
                                              b_idx.as<uint32_t>(), (int)n, 0, 64, st));
    HIPCHK(hipMemsetAsync(b_bad.p, 0, 8, st));   // [0] collision flag, [1] n_valid
    hipLaunchKernelGGL(k_heads, dim3(nblk(n, 256)), dim3(256), 0, st, b_key.as<uint64_t>(), n, b_head.as<uint32_t>(),
                       b_bad.as<uint32_t>() + 1);
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(b_tmp.p, tmp_bytes, b_head.as<uint32_t>(), b_cls.as<uint32_t>(), (int)n, st));
    uint32_t last_cls = 0, last_head = 0, nv32 = 0;
    HIPCHK(hipMemcpyAsync(&last_cls, b_cls.as<uint32_t>() + (n - 1), 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&last_head, b_head.as<uint32_t>() + (n - 1), 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&nv32, b_bad.as<uint32_t>() + 1, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const int n_runs = (int)(last_cls + last_head);
    if (n_runs == 0) return HGX_OK;
    const long n_valid = nv32;
    DevBuf b_rs, b_rf, b_rc, b_ws, b_w;
    ALLOC(b_rs, (size_t)n_runs * 4); ALLOC(b_rf, (size_t)n_runs * 4); ALLOC(b_rc, (size_t)n_runs * 8);
    hipLaunchKernelGGL(k_run_starts, dim3(nblk(n, 256)), dim3(256), 0, st, b_head.as<uint32_t>(), b_cls.as<uint32_t>(),
                       b_idx.as<uint32_t>(), n, b_rs.as<uint32_t>(), b_rf.as<uint32_t>());
    const int64_t *wsum = nullptr;
    if (row_weight) {
        ALLOC(b_w, n * 8); ALLOC(b_ws, n * 8);
        hipLaunchKernelGGL(k_gather_weights, dim3(nblk(n, 256)), dim3(256), 0, st, row_weight, b_idx.as<uint32_t>(), n, b_w.as<int64_t>());
        HIPCHK(hipcub::DeviceScan::InclusiveSum(b_tmp.p, tmp_bytes, b_w.as<int64_t>(), b_ws.as<int64_t>(), (int)n, st));
        wsum = b_ws.as<int64_t>();
    }
    hipLaunchKernelGGL(k_run_counts, dim3(nblk(n_runs, 256)), dim3(256), 0, st, b_rs.as<uint32_t>(), n_runs, n_valid, wsum,
                       b_rc.as<int64_t>());
    hipLaunchKernelGGL(k_verify, dim3(nblk(n_valid, 4)), dim3(256), 0, st, rows, w64, and_mask, b_idx.as<uint32_t>(),
                       b_cls.as<uint32_t>(), b_head.as<uint32_t>(), b_rf.as<uint32_t>(), n_valid, b_bad.as<int>());
    // first-seen order: sort runs by their first row
    DevBuf b_fs, b_rid0, b_rid, b_tmp2;
    ALLOC(b_fs, (size_t)n_runs * 4); ALLOC(b_rid0, (size_t)n_runs * 4); ALLOC(b_rid, (size_t)n_runs * 4);
    hipLaunchKernelGGL(k_iota, dim3(nblk(n_runs, 256)), dim3(256), 0, st, b_rid0.as<uint32_t>(), (long)n_runs);
    size_t tb2 = 0;
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb2, b_rf.as<uint32_t>(), b_fs.as<uint32_t>(), b_rid0.as<uint32_t>(),
                                              b_rid.as<uint32_t>(), n_runs, 0, 32, st));
    ALLOC(b_tmp2, tb2);
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(b_tmp2.p, tb2, b_rf.as<uint32_t>(), b_fs.as<uint32_t>(), b_rid0.as<uint32_t>(),
                                              b_rid.as<uint32_t>(), n_runs, 0, 32, st));
    HIPCHK(hipMalloc((void **)&cl->d_bits, (size_t)n_runs * w64 * 8));
    HIPCHK(hipMalloc((void **)&cl->d_count, (size_t)n_runs * 8));
    HIPCHK(hipMalloc((void **)&cl->d_first_row, (size_t)n_runs * 8));
    hipLaunchKernelGGL(k_gather_rows, dim3(nblk(n_runs, 4)), dim3(256), 0, st, rows, w64, and_mask, b_fs.as<uint32_t>(),
                       b_rid.as<uint32_t>(), b_rc.as<int64_t>(), n_runs, cl->d_bits, cl->d_count, cl->d_first_row);
    int bad = 0;
    HIPCHK(hipMemcpyAsync(&bad, b_bad.p, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    cl->n_classes = n_runs;
    if (bad) {
        hgx_set_error("64-bit class hash collision detected by the exact verify pass");
        return HGX_ECOLLISION;
    }
    return HGX_OK;
}

extern "C" int hgx_classes_destroy(hgx_classes *c) {
    if (!c) return HGX_OK;
    (void)hipFree(c->d_bits); (void)hipFree(c->d_count); (void)hipFree(c->d_first_row); (void)hipFree(c->d_bitsT);
    delete c;
    return HGX_OK;
}
extern "C" int hgx_classes_dims(const hgx_classes *c, int32_t *n, int32_t *a_pad) {
    ARGCHK(c);
    if (n) *n = c->n_classes;
    if (a_pad) *a_pad = c->a_pad;
    return HGX_OK;
}
extern "C" int hgx_classes_device(const hgx_classes *c, void **bits, void **count, void **first_row) {
    ARGCHK(c);
    if (bits) *bits = c->d_bits;
    if (count) *count = c->d_count;
    if (first_row) *first_row = c->d_first_row;
    return HGX_OK;
}
extern "C" int hgx_classes_to_host(const hgx_classes *c, uint64_t *bits, int64_t *count, int64_t *first_row) {
    ARGCHK(c);
    if (c->n_classes == 0) return HGX_OK;
    if (bits) HIPCHK(hipMemcpy(bits, c->d_bits, (size_t)c->n_classes * c->w64 * 8, hipMemcpyDeviceToHost));
    if (count) HIPCHK(hipMemcpy(count, c->d_count, (size_t)c->n_classes * 8, hipMemcpyDeviceToHost));
    if (first_row) HIPCHK(hipMemcpy(first_row, c->d_first_row, (size_t)c->n_classes * 8, hipMemcpyDeviceToHost));
    return HGX_OK;
}
extern "C" int hgx_classes_from_host(hgx_classes **out, const uint64_t *bits, const int64_t *count, int32_t n_classes, int32_t a_pad) {
    ARGCHK(out && n_classes >= 0 && a_pad > 0 && a_pad % 64 == 0);
    hgx_classes *cl = new hgx_classes();
    cl->a_pad = a_pad; cl->w64 = a_pad / 64; cl->n_classes = n_classes; cl->c64 = 0;
    cl->d_bits = nullptr; cl->d_count = nullptr; cl->d_first_row = nullptr; cl->d_bitsT = nullptr;
    *out = cl;
    if (n_classes == 0) return HGX_OK;
    ARGCHK(bits && count);
    HIPCHK(hipMalloc((void **)&cl->d_bits, (size_t)n_classes * cl->w64 * 8));
    HIPCHK(hipMalloc((void **)&cl->d_count, (size_t)n_classes * 8));
    HIPCHK(hipMalloc((void **)&cl->d_first_row, (size_t)n_classes * 8));
    HIPCHK(hipMemcpy(cl->d_bits, bits, (size_t)n_classes * cl->w64 * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(cl->d_count, count, (size_t)n_classes * 8, hipMemcpyHostToDevice));
    std::vector<int64_t> fr(n_classes);
    for (int i = 0; i < n_classes; ++i) fr[i] = i;
    HIPCHK(hipMemcpy(cl->d_first_row, fr.data(), (size_t)n_classes * 8, hipMemcpyHostToDevice));
    return HGX_OK;
}

// ------------------------------------------------------------------------------------------------
// bit-matrix transpose [C][w64] -> [a_pad][c64]: one wavefront per 64x64 tile; lane r loads row r's
// word, then 64 ballots peel the columns.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_transpose(const uint64_t *__restrict__ bits, int n_classes, int w64, int c64,
                                                   uint64_t *__restrict__ bitsT) {
    const int lane = threadIdx.x & 63;
    const long tile = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long n_tiles = (long)c64 * w64;
    if (tile >= n_tiles) return;
    const int cw = (int)(tile / w64), aw = (int)(tile % w64);
    const int c = cw * 64 + lane;
    const uint64_t x = (c < n_classes) ? bits[(size_t)c * w64 + aw] : 0ull;
    uint64_t mine = 0;
#pragma unroll 8
    for (int b = 0; b < 64; ++b) {
        const uint64_t col = __ballot((x >> b) & 1ull);
        if (lane == b) mine = col;
    }
    bitsT[(size_t)(aw * 64 + lane) * c64 + cw] = mine;
}

static int ensure_transposed(hgx_classes *c, hipStream_t st) {
    if (c->d_bitsT || c->n_classes == 0) return HGX_OK;
    c->c64 = (c->n_classes + 63) / 64;
    HIPCHK(hipMalloc((void **)&c->d_bitsT, (size_t)c->a_pad * c->c64 * 8));
    const long tiles = (long)c->c64 * c->w64;
    hipLaunchKernelGGL(k_transpose, dim3(nblk(tiles, 4)), dim3(256), 0, st, c->d_bits, c->n_classes, c->w64, c->c64, c->d_bitsT);
    HIPCHK(hipGetLastError());
    return HGX_OK;
}

// Gene_counts: per allele sum of class counts + first class containing it (one wavefront per allele
// over the transposed matrix, lane = class within a 64-class word)
__global__ __launch_bounds__(256) void k_allele_counts(const uint64_t *__restrict__ bitsT, int a_pad, int c64, int n_classes,
                                                       const int64_t *__restrict__ count, int64_t *__restrict__ out_count,
                                                       int32_t *__restrict__ out_first) {
    const int lane = threadIdx.x & 63;
    const long a = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (a >= a_pad) return;
    const uint64_t *row = bitsT + (size_t)a * c64;
    uint64_t s = 0;
    int first = 0x7fffffff;
    for (int w = 0; w < c64; ++w) {
        const uint64_t x = row[w];
        const int c = w * 64 + lane;
        if ((x >> lane) & 1ull) {
            s += (uint64_t)count[c];
            first = min(first, c);
        }
    }
    s = wave_sum_u64(s);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) first = min(first, __shfl_xor(first, m, 64));
    if (lane == 0) {
        out_count[a] = (int64_t)s;
        out_first[a] = first == 0x7fffffff ? -1 : first;
    }
}

extern "C" int hgx_allele_counts(const hgx_classes *cc, int64_t *count_host, int32_t *first_host) {
    ARGCHK(cc && count_host && first_host);
    hgx_classes *c = const_cast<hgx_classes *>(cc);
    const int A = c->a_pad;
    if (c->n_classes == 0) {
        for (int a = 0; a < A; ++a) { count_host[a] = 0; first_host[a] = -1; }
        return HGX_OK;
    }
    int rc = ensure_transposed(c, nullptr);
    if (rc) return rc;
    DevBuf b_c, b_f;
    ALLOC(b_c, (size_t)A * 8); ALLOC(b_f, (size_t)A * 4);
    hipLaunchKernelGGL(k_allele_counts, dim3(nblk(A, 4)), dim3(256), 0, nullptr, c->d_bitsT, A, c->c64, c->n_classes, c->d_count,
                       b_c.as<int64_t>(), b_f.as<int32_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(count_host, b_c.p, (size_t)A * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(first_host, b_f.p, (size_t)A * 4, hipMemcpyDeviceToHost));
    return HGX_OK;
}

// ------------------------------------------------------------------------------------------------
// 8a-8 EM (single_abundance, typing_common.py:1282-1410), FP64.
//   T(p)_a  proportional to  p_a * sum_{c contains a, s_c > 0} n_c / s_c,   s_c = sum_{b in c, b present} p_b
// rows pass (one wavefront per class) -> columns pass (one wavefront per allele over the transposed
// matrix) -> single-workgroup vector kernels for normalise / SQUAREM / diff / pruning.
// Invariant: p[a] == 0 wherever pres[a] == 0.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_em_rows(const uint64_t *__restrict__ bits, int n_classes, int w64,
                                                 const double *__restrict__ p, const int64_t *__restrict__ count,
                                                 double *__restrict__ wc, int init) {
    const int lane = threadIdx.x & 63;
    const long c = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (c >= n_classes) return;
    const uint64_t *row = bits + (size_t)c * w64;
    double s = 0.0;
    for (int w = 0; w < w64; ++w) {
        const uint64_t x = row[w];   // wave-uniform
        if ((x >> lane) & 1ull) s += init ? 1.0 : p[w * 64 + lane];
    }
    s = wave_sum_f64(s);
    if (lane == 0) wc[c] = s > 0.0 ? (double)count[c] / s : 0.0;
}

__global__ __launch_bounds__(256) void k_em_cols(const uint64_t *__restrict__ bitsT, int a_pad, int c64, int n_classes,
                                                 const double *__restrict__ wc, const double *__restrict__ p,
                                                 const uint8_t *__restrict__ pres, const double *__restrict__ len,
                                                 double *__restrict__ q, uint8_t *__restrict__ pres_out, int init) {
    const int lane = threadIdx.x & 63;
    const long a = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (a >= a_pad) return;
    if (!init && !pres[a]) {
        if (lane == 0) { q[a] = 0.0; pres_out[a] = 0; }
        return;
    }
    const uint64_t *row = bitsT + (size_t)a * c64;
    double t = 0.0;
    bool touched = false;
    for (int w = 0; w < c64; ++w) {
        const uint64_t x = row[w];
        const int c = w * 64 + lane;
        if (c < n_classes && ((x >> lane) & 1ull)) {
            const double v = wc[c];
            t += v;
            touched = touched || v > 0.0;
        }
    }
    t = wave_sum_f64(t);
    const bool any_t = __any(touched);
    if (lane == 0) {
        double v = init ? t : p[a] * t;
        if (len) v = v / len[a];
        q[a] = any_t ? v : 0.0;
        pres_out[a] = any_t ? 1 : 0;
    }
}

// scal[]: 0 total, 1 sum r^2, 2 sum v^2, 3 flag(extrapolated), 4 diff, 5 keyerror, 6 max
__device__ double block_sum(double v, double *sh) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    v = wave_sum_f64(v);
    __syncthreads();
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
    return t;
}
__device__ double block_max(double v, double *sh) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m, 64));
    __syncthreads();
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    double t = sh[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) t = fmax(t, sh[i]);
    return t;
}

// p_out = q / sum(q) over pres   (normalize / normalize_len: the division by len happened in k_em_cols)
__global__ __launch_bounds__(1024) void k_em_normalize(const double *__restrict__ q, const uint8_t *__restrict__ pres, int a_pad,
                                                       double *__restrict__ p_out) {
    __shared__ double sh[16];
    double s = 0.0;
    for (int a = threadIdx.x; a < a_pad; a += blockDim.x) if (pres[a]) s += q[a];
    const double tot = block_sum(s, sh);
    for (int a = threadIdx.x; a < a_pad; a += blockDim.x) p_out[a] = pres[a] ? q[a] / tot : 0.0;
}

// SQUAREM extrapolation (common:1361-1380).  Writes p2 <- max(0, p - 2 g r + g^2 v) when sum v^2 > 0.
__global__ __launch_bounds__(1024) void k_em_squarem(const double *__restrict__ p, const uint8_t *__restrict__ pres,
                                                     const double *__restrict__ p1, const uint8_t *__restrict__ pres1,
                                                     double *__restrict__ p2, uint8_t *__restrict__ pres2, int a_pad,
                                                     double *__restrict__ scal) {
    __shared__ double sh[16];
    double sr = 0.0, sv = 0.0, key = 0.0;
    for (int a = threadIdx.x; a < a_pad; a += blockDim.x) {
        if (!pres[a]) continue;
        if (!pres1[a] || !pres2[a]) { key = 1.0; continue; }
        const double r = p1[a] - p[a];
        const double v = p2[a] - p1[a] - r;
        sr += r * r;
        sv += v * v;
    }
    const double tsr = block_sum(sr, sh), tsv = block_sum(sv, sh), tkey = block_sum(key, sh);
    if (tsv > 0.0 && tkey == 0.0) {
        const double g = -sqrt(tsr / tsv);
        for (int a = threadIdx.x; a < a_pad; a += blockDim.x) {
            if (!pres[a]) continue;
            const double r = p1[a] - p[a];
            const double v = p2[a] - p1[a] - r;
            const double x = p[a] - 2 * g * r + g * g * v;
            p2[a] = fmax(0.0, x);
            pres2[a] = 1;
        }
    }
    if (threadIdx.x == 0) {
        scal[1] = tsr; scal[2] = tsv; scal[3] = (tsv > 0.0) ? 1.0 : 0.0; scal[5] = tkey;
    }
}

// diff = prob_diff(p, pn) (common:1272-1279); then p <- pn; optional pruning (common:1338-1346)
__global__ __launch_bounds__(1024) void k_em_advance(double *__restrict__ p, uint8_t *__restrict__ pres,
                                                     const double *__restrict__ pn, const uint8_t *__restrict__ presn,
                                                     int a_pad, int prune, double *__restrict__ scal) {
    __shared__ double sh[16];
    double d = 0.0, mx = 0.0;
    for (int a = threadIdx.x; a < a_pad; a += blockDim.x) {
        if (pres[a]) d += presn[a] ? fabs(p[a] - pn[a]) : p[a];
        if (presn[a]) mx = fmax(mx, pn[a]);
    }
    const double td = block_sum(d, sh);
    const double tm = block_max(mx, sh);
    for (int a = threadIdx.x; a < a_pad; a += blockDim.x) {
        bool keep = presn[a];
        if (prune && keep) keep = pn[a] >= tm / 10.0;
        pres[a] = keep ? 1 : 0;
        p[a] = keep ? pn[a] : 0.0;
    }
    if (threadIdx.x == 0) { scal[4] = td; scal[6] = tm; }
}

// final select_alleles + normalise (common:1402-1407); q_out = -1 for alleles not in the dict
__global__ __launch_bounds__(1024) void k_em_finish(const double *__restrict__ p, const uint8_t *__restrict__ pres,
                                                    const double *__restrict__ len, int a_pad, int prune,
                                                    double *__restrict__ out) {
    __shared__ double sh[16];
    double mx = 0.0;
    for (int a = threadIdx.x; a < a_pad; a += blockDim.x) if (pres[a]) mx = fmax(mx, p[a]);
    const double tm = block_max(mx, sh);
    double s = 0.0;
    for (int a = threadIdx.x; a < a_pad; a += blockDim.x) {
        const bool keep = pres[a] && (!prune || p[a] >= tm / 10.0);
        if (keep) s += len ? p[a] / len[a] : p[a];
    }
    const double tot = block_sum(s, sh);
    for (int a = threadIdx.x; a < a_pad; a += blockDim.x) {
        const bool keep = pres[a] && (!prune || p[a] >= tm / 10.0);
        out[a] = keep ? (len ? p[a] / len[a] / tot : p[a] / tot) : -1.0;
    }
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
    int rc = ensure_transposed(c, st);
    if (rc) return rc;
    DevBuf b_p, b_p1, b_p2, b_q, b_wc, b_pr, b_pr1, b_pr2, b_len, b_scal, b_out;
    ALLOC(b_p, A * 8); ALLOC(b_p1, A * 8); ALLOC(b_p2, A * 8); ALLOC(b_q, A * 8); ALLOC(b_out, A * 8);
    ALLOC(b_wc, (size_t)C * 8); ALLOC(b_pr, A); ALLOC(b_pr1, A); ALLOC(b_pr2, A); ALLOC(b_scal, 8 * 8);
    double *d_len = nullptr;
    if (allele_len) {
        std::vector<double> l(A, 1.0);
        for (int a = 0; a < n_alleles; ++a) l[a] = (double)allele_len[a];
        ALLOC(b_len, A * 8);
        HIPCHK(hipMemcpyAsync(b_len.p, l.data(), A * 8, hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
        d_len = b_len.as<double>();
    }
    double *p = b_p.as<double>(), *p1 = b_p1.as<double>(), *p2 = b_p2.as<double>(), *q = b_q.as<double>();
    uint8_t *pr = b_pr.as<uint8_t>(), *pr1 = b_pr1.as<uint8_t>(), *pr2 = b_pr2.as<uint8_t>();
    double *wc = b_wc.as<double>(), *scal = b_scal.as<double>();
    const dim3 g_rows(nblk(C, 4)), g_cols(nblk(A, 4)), b256(256), b1024(1024);
    HIPCHK(hipMemsetAsync(scal, 0, 64, st));

    auto next_prob = [&](const double *pin, const uint8_t *prin, double *pout, uint8_t *prout, int init) {
        hipLaunchKernelGGL(k_em_rows, g_rows, b256, 0, st, c->d_bits, C, c->w64, pin, c->d_count, wc, init);
        hipLaunchKernelGGL(k_em_cols, g_cols, b256, 0, st, c->d_bitsT, A, c->c64, C, wc, pin, prin, d_len, q, prout, init);
        hipLaunchKernelGGL(k_em_normalize, dim3(1), b1024, 0, st, q, prout, A, pout);
    };
    // initial mass: sum_c n_c / |S_c|   (common:1299-1309)
    next_prob(p, pr, p, pr, 1);
    double h_scal[8];
    double diff = 1.0;
    int iter = 0;
    while (diff > 0.0001 && iter < 1000) {
        next_prob(p, pr, p1, pr1, 0);
        next_prob(p1, pr1, p2, pr2, 0);
        hipLaunchKernelGGL(k_em_squarem, dim3(1), b1024, 0, st, p, pr, p1, pr1, p2, pr2, A, scal);
        HIPCHK(hipMemcpyAsync(h_scal, scal, 64, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (h_scal[5] != 0.0) {
            hgx_set_error("EM: allele missing from the next estimate (the reference raises KeyError here, common:1365-1369)");
            return HGX_EKEY;
        }
        if (h_scal[3] != 0.0) next_prob(p2, pr2, p1, pr1, 0);
        hipLaunchKernelGGL(k_em_advance, dim3(1), b1024, 0, st, p, pr, p1, pr1, A, (iter >= 10 && remove_low) ? 1 : 0, scal);
        HIPCHK(hipMemcpyAsync(h_scal, scal, 64, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        diff = h_scal[4];
        iter += 1;
    }
    hipLaunchKernelGGL(k_em_finish, dim3(1), b1024, 0, st, p, pr, d_len, A, remove_low ? 1 : 0, b_out.as<double>());
    HIPCHK(hipGetLastError());
    std::vector<double> out(A);
    HIPCHK(hipMemcpyAsync(out.data(), b_out.p, A * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int a = 0; a < n_alleles; ++a) prob_host[a] = out[a];
    if (n_iter_host) *n_iter_host = iter;
    return HGX_OK;
}
