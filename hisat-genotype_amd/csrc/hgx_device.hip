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

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

#include <map>
#include <mutex>
#include <unordered_map>

#include "hgx_common.hpp"

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


extern "C" int hgx_device_count(int *n) { HIPCHK(hipGetDeviceCount(n)); return HGX_OK; }
// ---- pinned staging (see hgx_common.hpp) -------------------------------------------------------------------
namespace {
struct Staging {
    char *buf = nullptr;
    size_t cap = 0, used = 0;
    struct Pending { void *dst; const char *src; size_t n; };
    std::vector<Pending> pending;
    std::vector<hipStream_t> streams;       // streams with staged transfers in flight (almost always exactly one)
    void note(hipStream_t st) {
        for (auto s : streams) if (s == st) return;
        streams.push_back(st);
    }
    // buffers are recycled through a process-wide free list: short-lived worker threads (one per typed sample) must not
    // pay hipHostMalloc / hipHostFree, which serialise with every other thread's launches inside the runtime
    static std::mutex &mu() { static std::mutex m; return m; }
    static std::vector<char *> &free_list() { static std::vector<char *> v; return v; }
    ~Staging() {
        if (buf) { std::lock_guard<std::mutex> g(mu()); free_list().push_back(buf); }
    }
    char *take(size_t n) {
        const size_t need = (n + 63) & ~(size_t)63;
        if (!buf) {
            cap = 1u << 20;
            {
                std::lock_guard<std::mutex> g(mu());
                if (!free_list().empty()) { buf = free_list().back(); free_list().pop_back(); }
            }
            if (!buf && hipHostMalloc((void **)&buf, cap) != hipSuccess) { buf = nullptr; cap = 0; return nullptr; }
        }
        if (used + need > cap) return nullptr;
        char *r = buf + used;
        used += need;
        return r;
    }
};
thread_local Staging g_stage;
constexpr size_t STAGE_MAX = 256u << 10;      // larger transfers go directly (the staging hop no longer dominates)
}  // namespace

int hgx_d2h(void *dst, const void *src, size_t n, hipStream_t st) {
    char *s = n <= STAGE_MAX ? g_stage.take(n) : nullptr;
    if (!s) { HIPCHK(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, st)); return HGX_OK; }
    HIPCHK(hipMemcpyAsync(s, src, n, hipMemcpyDeviceToHost, st));
    g_stage.pending.push_back({dst, s, n});
    g_stage.note(st);
    return HGX_OK;
}
int hgx_h2d(void *dst, const void *src, size_t n, hipStream_t st) {
    char *s = n <= STAGE_MAX ? g_stage.take(n) : nullptr;
    if (!s) { HIPCHK(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, st)); return HGX_OK; }
    memcpy(s, src, n);                            // the region stays reserved until this thread's next hgx_sync
    HIPCHK(hipMemcpyAsync(dst, s, n, hipMemcpyHostToDevice, st));
    g_stage.note(st);
    return HGX_OK;
}
int hgx_sync(hipStream_t st) {
    HIPCHK(hipStreamSynchronize(st));
    // the staging buffer is recycled below: transfers this thread staged on OTHER streams must be complete too
    for (auto s2 : g_stage.streams) if (s2 != st) HIPCHK(hipStreamSynchronize(s2));
    g_stage.streams.clear();
    for (auto &p : g_stage.pending) memcpy(p.dst, p.src, p.n);
    g_stage.pending.clear();
    g_stage.used = 0;
    return HGX_OK;
}

extern "C" int hgx_set_device(int dev) { HIPCHK(hipSetDevice(dev)); return HGX_OK; }
void *hgx_pool_alloc(size_t bytes);
void hgx_pool_free(void *p);
extern "C" int hgx_dev_alloc(void **p, size_t bytes) {
    ARGCHK(p != nullptr);
    if (bytes <= (16u << 20)) {
        // small buffers (masks, vectors, per-sample tables) come from the caching pool: hipMalloc / hipFree cost tens of
        // microseconds and hipFree synchronises the whole device, which would stall every other sample in flight
        *p = hgx_pool_alloc(bytes);
        if (!*p) { hgx_set_error("device allocation of %zu bytes failed", bytes); return HGX_ENOMEM; }
        return HGX_OK;
    }
    HIPCHK(hipMalloc(p, bytes));
    return HGX_OK;
}
extern "C" int hgx_dev_free(void *p) { hgx_pool_free(p); return HGX_OK; }   // pool blocks are recycled, foreign ones hipFree'd
extern "C" int hgx_memcpy_h2d(void *d, const void *s, size_t n, void *st) {
    int rc = hgx_h2d(d, s, n, (hipStream_t)st);
    return rc ? rc : hgx_sync((hipStream_t)st);
}
extern "C" int hgx_memcpy_h2d_async(void *d, const void *s, size_t n, void *st) {
    if (n > STAGE_MAX) return hgx_memcpy_h2d(d, s, n, st);
    return hgx_h2d(d, s, n, (hipStream_t)st);
}
extern "C" int hgx_memcpy_d2h(void *d, const void *s, size_t n, void *st) {
    int rc = hgx_d2h(d, s, n, (hipStream_t)st);
    return rc ? rc : hgx_sync((hipStream_t)st);
}
extern "C" int hgx_memset(void *d, int v, size_t n, void *st) {
    HIPCHK(hipMemsetAsync(d, v, n, (hipStream_t)st));
    return HGX_OK;
}
extern "C" int hgx_stream_sync(void *st) { return hgx_sync((hipStream_t)st); }
extern "C" int hgx_stream_create(void **st) {
    ARGCHK(st != nullptr);
    hipStream_t s;
    HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *st = (void *)s;
    return HGX_OK;
}
extern "C" int hgx_stream_create_prio(void **st, int high_priority) {
    ARGCHK(st != nullptr);
    int least = 0, greatest = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t s;
    HIPCHK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, high_priority ? greatest : least));
    *st = (void *)s;
    return HGX_OK;
}
bool hgx_ss_forget_stream(void *stream);      // hgx_type.hip: placed streams (hgx_stream_create_placed) are known to the stream placement
extern "C" int hgx_stream_destroy(void *st) {
    if (st && !hgx_ss_forget_stream(st)) HIPCHK(hipStreamDestroy((hipStream_t)st));
    return HGX_OK;
}
extern "C" int hgx_event_create(void **ev) {
    ARGCHK(ev != nullptr);
    hipEvent_t e;
    HIPCHK(hipEventCreate(&e));
    *ev = (void *)e;
    return HGX_OK;
}
extern "C" int hgx_event_destroy(void *ev) { if (ev) HIPCHK(hipEventDestroy((hipEvent_t)ev)); return HGX_OK; }
extern "C" int hgx_event_record(void *ev, void *st) { HIPCHK(hipEventRecord((hipEvent_t)ev, (hipStream_t)st)); return HGX_OK; }
extern "C" int hgx_stream_wait_event(void *st, void *ev) { HIPCHK(hipStreamWaitEvent((hipStream_t)st, (hipEvent_t)ev, 0)); return HGX_OK; }
extern "C" int hgx_event_elapsed_ms(void *a, void *b, float *ms) {
    ARGCHK(a && b && ms);
    HIPCHK(hipEventSynchronize((hipEvent_t)b));
    HIPCHK(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
    return HGX_OK;
}


// ------------------------------------------------------------------------------------------------
// caching device allocator: scratch buffers and class matrices are recycled instead of going through
// hipMalloc/hipFree (which synchronise the device) on every step.  Blocks are binned by size rounded
// up to a power of two >= 256 B; hgx_pool_trim() returns everything to the driver.
// ------------------------------------------------------------------------------------------------
namespace {
struct Pool {
    std::mutex mu;
    std::multimap<std::pair<int, size_t>, void *> free_blocks;      // (device, size) -> block: a block never changes device
    std::unordered_map<void *, std::pair<int, size_t>> size_of;
};
int cur_device() { int d = 0; (void)hipGetDevice(&d); return d; }
Pool &pool() { static Pool p; return p; }
size_t round_size(size_t n) {
    size_t r = 256;
    while (r < n) r <<= 1;
    if (r > (1u << 20)) {                 // large blocks: 1/16-of-a-power-of-two granularity instead of 2x
        const size_t step = r >> 4;
        r = (n + step - 1) / step * step;
    }
    return r;
}
}   // namespace

void *hgx_pool_alloc(size_t bytes) {
    Pool &P = pool();
    const size_t sz = round_size(bytes ? bytes : 8);
    const int dev = cur_device();
    {
        std::lock_guard<std::mutex> g(P.mu);
        auto it = P.free_blocks.find({dev, sz});
        if (it != P.free_blocks.end()) {
            void *p = it->second;
            P.free_blocks.erase(it);
            return p;
        }
    }
    void *p = nullptr;
    if (hipMalloc(&p, sz) != hipSuccess) {
        // give this device's cached blocks back and retry once
        {
            std::lock_guard<std::mutex> g(P.mu);
            for (auto it = P.free_blocks.begin(); it != P.free_blocks.end();) {
                if (it->first.first == dev) { P.size_of.erase(it->second); (void)hipFree(it->second); it = P.free_blocks.erase(it); }
                else ++it;
            }
        }
        if (hipMalloc(&p, sz) != hipSuccess) return nullptr;
    }
    std::lock_guard<std::mutex> g(P.mu);
    P.size_of[p] = {dev, sz};
    return p;
}
void hgx_pool_free(void *p) {
    if (!p) return;
    Pool &P = pool();
    std::lock_guard<std::mutex> g(P.mu);
    auto it = P.size_of.find(p);
    if (it == P.size_of.end()) { (void)hipFree(p); return; }
    P.free_blocks.emplace(it->second, p);
}
void hgx_host_pool_trim();
extern "C" int hgx_pool_trim(void) {
    hgx_host_pool_trim();
    Pool &P = pool();
    std::lock_guard<std::mutex> g(P.mu);
    // (a hipFree from another device's context is legal: the runtime knows the owner of the pointer)
    for (auto &kv : P.free_blocks) { P.size_of.erase(kv.second); (void)hipFree(kv.second); }
    P.free_blocks.clear();
    return HGX_OK;
}

// ------------------------------------------------------------------------------------------------
// 8a-0 index
// ------------------------------------------------------------------------------------------------

extern "C" int32_t hgx_a_pad(int32_t n) { return (n + 511) / 512 * 512; }   // rows of a_pad/64 words: multiple of 8 words

// The three device tables of an index live in ONE allocation, [link bits | exon mask | gene mask], so that a rank that did not
// build the locus can receive them with a single collective straight into place (hgx_index_device_block, 8e).
static int index_alloc(hgx_index **out, int32_t n_alleles, int32_t n_vars) {
    ARGCHK(out && n_alleles > 0 && n_vars >= 0);
    ARGCHK(n_vars <= 65535 * 32);
    hgx_index *ix = new hgx_index();
    ix->n_alleles = n_alleles;
    ix->a_pad = hgx_a_pad(n_alleles);
    ix->n_vars = n_vars;
    ix->n_words = std::max(1, (n_vars + 31) / 32);
    ix->w64 = ix->a_pad / 64;
    const size_t nb = (size_t)ix->n_words * ix->a_pad * sizeof(uint32_t);
    void *block = nullptr;
    if (hipMalloc(&block, nb + 2 * (size_t)ix->w64 * 8) != hipSuccess) {
        delete ix;
        hgx_set_error("device allocation of the locus index failed");
        return HGX_ENOMEM;
    }
    ix->d_bits = (uint32_t *)block;
    ix->d_exon_mask = (uint64_t *)((char *)block + nb);
    ix->d_gene_mask = ix->d_exon_mask + ix->w64;
    *out = ix;
    return HGX_OK;
}
extern "C" int hgx_index_create(hgx_index **out, int32_t n_alleles, int32_t n_vars, const uint32_t *bits,
                                const uint64_t *exon_mask, const uint64_t *gene_mask) {
    ARGCHK(out && bits && exon_mask && gene_mask);
    *out = nullptr;
    hgx_index *ix = nullptr;
    { int rc_ = index_alloc(&ix, n_alleles, n_vars); if (rc_) return rc_; }
    const size_t nb = (size_t)ix->n_words * ix->a_pad * sizeof(uint32_t);
    if (hipMemcpy(ix->d_bits, bits, nb, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(ix->d_exon_mask, exon_mask, ix->w64 * 8, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(ix->d_gene_mask, gene_mask, ix->w64 * 8, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(ix->d_bits);
        delete ix;
        hgx_set_error("upload of the locus index failed");
        return HGX_EHIP;
    }
    *out = ix;
    return HGX_OK;
}
// an index of the given shape with UNINITIALISED tables: the receiving side of an index broadcast
extern "C" int hgx_index_create_device(hgx_index **out, int32_t n_alleles, int32_t n_vars) {
    ARGCHK(out);
    *out = nullptr;
    return index_alloc(out, n_alleles, n_vars);
}
extern "C" int hgx_index_destroy(hgx_index *ix) {
    if (!ix) return HGX_OK;
    (void)hipFree(ix->d_bits);                 // one block: the masks live behind the bit matrix
    hgx_pool_free(ix->d_pid); hgx_pool_free(ix->d_vals); hgx_pool_free(ix->d_nval);
    delete ix->pat_mu;
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
extern "C" int hgx_index_device_block(const hgx_index *ix, void **p, size_t *bytes) {
    ARGCHK(ix && p && bytes);
    *p = ix->d_bits;
    *bytes = (size_t)ix->n_words * ix->a_pad * sizeof(uint32_t) + 2 * (size_t)ix->w64 * 8;
    return HGX_OK;
}

// ------------------------------------------------------------------------------------------------
// 8a-5 stage 1: piece x allele compatibility.  compat(a) <=> AND_i ((bits[lo+i][a] & MP_i) == P_i)
// ------------------------------------------------------------------------------------------------
#ifdef HGX_LAB
#include "lab/hgx_compat_lab.inc"         // rounds 1-3: the L2-served and the LDS-tiled kernels (comparison forms)
#endif

// ------------------------------------------------------------------------------------------------
// Pattern form (round 4; the one hgx_piece_compat launches when the locus fits).  Of the ~7 000 alleles of an HLA locus only
// ~100 DISTINCT values occur in any one 32-variant word (max 222 at the bench's HLA-A, 367 at B): the word tests of a piece
// need to be made once per distinct value, not once per allele.  Per LOCUS (made once from the index, ensure_patterns):
// vals[w][id] = the distinct values of word w, pid[w][a] = the id of allele a's value.  Per window of 8 variant words and group
// of 64 pieces, a workgroup
//   phase 1  tests every (window word j, value id) against the 64 pieces' (MP, P) words of j at once: match[j][id] = 64 verdict
//            bits (a piece that does not cover j passes), ~8 x 100 items of 64 tests -- what 7 alleles cost in the tiled form;
//   phase 2  every allele ANDs the eight words match[j][pid[j][a]] (eight LDS reads for 64 pieces, where the tiled form makes
//            64 x 3.85) and a 64 x 64 bit transpose per wavefront (wave_transpose64) turns "lane = allele, bit = piece" into the
//            compat words "lane = piece, bit = allele", stored 32 bytes per lane.
// Exact (the same tests on the same words), any piece order is correct; pieces wider than the window or over a word with more
// than HGX_PAT_D values are tested straight from the index.
// ------------------------------------------------------------------------------------------------
#define PP_T 1024
#define PP_W 8
#define PP_NW 8
#define PP_G 8                // 64-allele words per wavefront: 8 x 64 x 16 waves = 8192 alleles per workgroup -- a whole locus: the tests of a
                              // (word, value) are then made once per 256 pieces, and the 1 M-read sample's 249 workgroups are one per CU
                              // (measured: PB x G = 256 x 4 0.136 ms, 512 x 4 0.077, 256 x 8 0.072, 384 x 8 0.107, 512 x 8 0.132, 1024 x 8 0.239)
template <int PP_PB>            // pieces per workgroup (hgx_piece_compat launches 256)
__global__ __launch_bounds__(PP_T) void k_piece_compat_pat(const uint32_t *__restrict__ bits, int a_pad, int n_index_words,
                                                           const uint16_t *__restrict__ pid, const uint32_t *__restrict__ vals,
                                                           const int32_t *__restrict__ nval, const hgx_piece *__restrict__ pieces,
                                                           const uint32_t *__restrict__ masks, int n_pieces, uint64_t *__restrict__ compat,
                                                           int w64) {
    __shared__ uint32_t smask[PP_PB][2 * PP_NW];
    __shared__ int s_lo[PP_PB], s_nw[PP_PB];
    __shared__ int s_end, s_big;
    __shared__ uint32_t s_vals[PP_W][HGX_PAT_D];
    __shared__ int s_nval[PP_W], s_voff[PP_W + 1];
    __shared__ uint2 s_wm[PP_W][64];
    __shared__ uint64_t s_match[PP_W][HGX_PAT_D];
    const int tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int a_base = blockIdx.y * (nthr * PP_G) + wv * (64 * PP_G);          // this wave's first allele
    const int word0 = a_base >> 6;                                          // ... = its first 64-allele word of a compat row
    const int p0 = blockIdx.x * PP_PB;
    const int np = min(PP_PB, n_pieces - p0);
    for (int t = tid; t < PP_PB * 4; t += nthr) {       // descriptors and masks: t = (piece, quarter of its 16 mask words)
        const int p = t >> 2, q = t & 3;
        if (p < np) {
            const hgx_piece pc = pieces[p0 + p];
            const int nw2 = 2 * (int)pc.n_words;
            if (q == 0) { s_lo[p] = pc.lo_word; s_nw[p] = pc.n_words; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = 4 * q + k;
                smask[p][i] = i < nw2 ? masks[pc.mask_off + i] : 0u;
            }
        }
    }
    __syncthreads();
    int cur = 0;
    while (cur < np) {
        const int win_lo = s_lo[cur];
        if (tid == 0) { s_end = np; s_big = 0; }
        __syncthreads();
        if (tid < PP_W) {
            const int w = win_lo + tid;
            const int nv = w < n_index_words ? nval[w] : 0;
            s_nval[tid] = nv;
            if (nv > HGX_PAT_D) s_big = 1;
        }
        for (int t = cur + tid; t < np; t += nthr) {                 // (a workgroup of a small locus has fewer threads than pieces)
            const int l = s_lo[t];
            if (l < win_lo || s_nw[t] > PP_NW || l + s_nw[t] - win_lo > PP_W) { atomicMin(&s_end, t); break; }
        }
        __syncthreads();
        if (s_nw[cur] > PP_NW || s_big) {
            // a piece wider than the window, or a word with too many values: straight from the index
            const hgx_piece pc = pieces[p0 + cur];
            const uint32_t *m = masks + pc.mask_off;
#pragma unroll
            for (int g = 0; g < PP_G; ++g) {
                const int a = a_base + 64 * g + lane;
                bool ok = true;
                for (int i = 0; i < (int)pc.n_words; ++i) {
                    const uint32_t r = a < a_pad ? bits[(size_t)(pc.lo_word + i) * a_pad + a] : 0u;
                    ok = ok && ((r & m[2 * i]) == m[2 * i + 1]);
                }
                const uint64_t b = __ballot(ok);
                if (lane == 0 && word0 + g < w64) compat[(size_t)(p0 + cur) * w64 + word0 + g] = b;
            }
            ++cur;
            __syncthreads();
            continue;
        }
        const int end = s_end;
        if (tid == 0) {
            int o = 0;
            for (int j = 0; j < PP_W; ++j) { s_voff[j] = o; o += s_nval[j]; }
            s_voff[PP_W] = o;
        }
        for (int j = 0; j < PP_W; ++j) {
            const int nv = s_nval[j];
            for (int id = tid; id < nv; id += nthr) s_vals[j][id] = vals[(size_t)(win_lo + j) * HGX_PAT_D + id];
        }
        // this thread's alleles' value ids in the window's words, two per register
        uint32_t pr[PP_G][PP_W / 2];
#pragma unroll
        for (int g = 0; g < PP_G; ++g) {
            const int a = a_base + 64 * g + lane;
#pragma unroll
            for (int j = 0; j < PP_W; j += 2) {
                const int w = win_lo + j;
                const uint32_t lo16 = (a < a_pad && w < n_index_words) ? pid[(size_t)w * a_pad + a] : 0u;
                const uint32_t hi16 = (a < a_pad && w + 1 < n_index_words) ? pid[(size_t)(w + 1) * a_pad + a] : 0u;
                pr[g][j >> 1] = lo16 | (hi16 << 16);
            }
        }
        __syncthreads();
        const int total = s_voff[PP_W];
        for (int g0 = cur; g0 < end; g0 += 64) {
            const int cnt = min(end, g0 + 64) - g0;
            for (int t = tid; t < PP_W * 64; t += nthr) {          // the group's (MP, P) word of every window word (0, 0 = passes)
                const int j = t >> 6, k = t & 63;
                uint2 wm = make_uint2(0u, 0u);
                if (k < cnt) {
                    const int off = s_lo[g0 + k] - win_lo, nw = s_nw[g0 + k];
                    if (j >= off && j < off + nw) wm = make_uint2(smask[g0 + k][2 * (j - off)], smask[g0 + k][2 * (j - off) + 1]);
                }
                s_wm[j][k] = wm;
            }
            __syncthreads();
            for (int t = tid; t < total; t += nthr) {              // phase 1: one (word, value) per thread, 64 pieces each
                int j = 0;
#pragma unroll
                for (int q = 1; q < PP_W; ++q) j += t >= s_voff[q] ? 1 : 0;
                const int id = t - s_voff[j];
                const uint32_t v = s_vals[j][id];
                uint32_t mlo = 0, mhi = 0;
#pragma unroll 8
                for (int k = 0; k < 32; ++k) {
                    const uint2 a = s_wm[j][k], b = s_wm[j][k + 32];
                    mlo |= (uint32_t)((v & a.x) == a.y) << k;
                    mhi |= (uint32_t)((v & b.x) == b.y) << k;
                }
                s_match[j][id] = ((uint64_t)mhi << 32) | mlo;
            }
            __syncthreads();
            uint64_t out[PP_G];                                     // phase 2
#pragma unroll
            for (int g = 0; g < PP_G; ++g) {
                uint64_t v = ~0ull;
#pragma unroll
                for (int j = 0; j < PP_W; ++j) {
                    const uint32_t id = (pr[g][j >> 1] >> (16 * (j & 1))) & 0xffffu;
                    if (s_nval[j] > 0) v &= s_match[j][id];
                }
                out[g] = wave_transpose64(v);                       // lane k: piece g0 + k over this wave's g-th 64 alleles
            }
            if (lane < cnt) {
                uint64_t *row = compat + (size_t)(p0 + g0 + lane) * w64 + word0;
#pragma unroll
                for (int g = 0; g < PP_G; ++g)
                    if (word0 + g < w64) row[g] = out[g];
            }
        }
        __syncthreads();
        cur = end;
    }
}

// the pattern tables of an index, made once from its bit matrix (host: sort + unique per word; 72 words x 7 168 alleles at HLA-A)
static int ensure_patterns(hgx_index *ix, hipStream_t st) {
    if (ix->pat_state) return HGX_OK;
    static std::mutex mu;
    std::lock_guard<std::mutex> g(mu);
    if (ix->pat_state) return HGX_OK;
    const size_t nw = (size_t)ix->n_words, ap = (size_t)ix->a_pad;
    std::vector<uint32_t> hb(nw * ap);
    HIPCHK(hipMemcpyAsync(hb.data(), ix->d_bits, hb.size() * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    std::vector<uint16_t> pid(nw * ap);
    std::vector<uint32_t> vals(nw * HGX_PAT_D, 0u);
    std::vector<int32_t> nval(nw, 0);
    std::vector<uint32_t> u;
    int dmax = 0;
    bool fits = true;
    for (size_t w = 0; w < nw; ++w) {
        u.assign(hb.begin() + w * ap, hb.begin() + (w + 1) * ap);
        std::sort(u.begin(), u.end());
        u.erase(std::unique(u.begin(), u.end()), u.end());
        dmax = std::max(dmax, (int)u.size());
        nval[w] = (int32_t)u.size();
        if (u.size() > 65535) { fits = false; break; }
        if (u.size() <= HGX_PAT_D) std::copy(u.begin(), u.end(), vals.begin() + w * HGX_PAT_D);
        for (size_t a = 0; a < ap; ++a) pid[w * ap + a] = (uint16_t)(std::lower_bound(u.begin(), u.end(), hb[w * ap + a]) - u.begin());
    }
    ix->pat_dmax = dmax;
    if (!fits) { ix->pat_state = 2; return HGX_OK; }
    ix->d_pid = (uint16_t *)hgx_pool_alloc(std::max<size_t>(pid.size() * 2, 16));
    ix->d_vals = (uint32_t *)hgx_pool_alloc(std::max<size_t>(vals.size() * 4, 16));
    ix->d_nval = (int32_t *)hgx_pool_alloc(std::max<size_t>(nval.size() * 4, 16));
    if (!ix->d_pid || !ix->d_vals || !ix->d_nval) { hgx_set_error("device allocation of the pattern tables failed"); return HGX_ENOMEM; }
    HIPCHK(hipMemcpyAsync(ix->d_pid, pid.data(), pid.size() * 2, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ix->d_vals, vals.data(), vals.size() * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ix->d_nval, nval.data(), nval.size() * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    ix->pat_state = 1;
    return HGX_OK;
}

extern "C" int hgx_piece_compat(const hgx_index *ix, const hgx_piece *pieces, const uint32_t *masks, int32_t n_pieces,
                                uint64_t *compat, void *stream) {
    ARGCHK(ix && n_pieces >= 0);
    if (n_pieces == 0) return HGX_OK;
    ARGCHK(pieces && masks && compat);
#ifdef HGX_LAB
    const bool untiled = hgx_test_switch("piece_untiled") != nullptr;     // the L2-served kernel of round 1
    const bool tiled = hgx_test_switch("piece_tiled") != nullptr;         // the LDS-tiled kernel of rounds 1-3
    if (untiled) {
        const int chunks = (ix->w64 + PC_GROUPS - 1) / PC_GROUPS;
        const long waves = (long)n_pieces * chunks;
        const long blocks = (waves + 3) / 4;
        hipLaunchKernelGGL(k_piece_compat, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, ix->d_bits, ix->a_pad,
                           pieces, masks, n_pieces, compat, ix->w64, chunks);
        HIPCHK(hipGetLastError());
        return HGX_OK;
    }
    if (tiled) {
        hipLaunchKernelGGL(k_piece_compat_tiled, dim3((n_pieces + PT_PB - 1) / PT_PB, (ix->a_pad + 1023) / 1024), dim3(PT_T), 0,
                           (hipStream_t)stream, ix->d_bits, ix->a_pad, ix->n_words, pieces, masks, n_pieces, compat, ix->w64);
        HIPCHK(hipGetLastError());
        return HGX_OK;
    }
#endif
    {
        const int rc_ = ensure_patterns(const_cast<hgx_index *>(ix), (hipStream_t)stream);
        if (rc_) return rc_;
    }
    if (ix->pat_state != 1) { hgx_set_error("the locus does not fit the pattern tables of the piece x allele kernel (more than 65535 distinct values in a variant word)"); return HGX_EINVAL; }
    {
        const int thr = std::min(PP_T, std::max(64, ((ix->a_pad + 64 * PP_G - 1) / (64 * PP_G)) * 64));
        const int per_wg = thr * PP_G;
        // (pieces per workgroup: 256.  More -- 512, 1024 -- amortise the tests of a (word, value) over more pieces but are slower at every
        // size tried: 63 000 pieces 0.071 / 0.109 / 0.196 ms, 177 000 pieces 0.188 / 0.208 / 0.205 ms: tools/compat_probe.py)
        hipLaunchKernelGGL(k_piece_compat_pat<256>, dim3((n_pieces + 255) / 256, (ix->a_pad + per_wg - 1) / per_wg), dim3(thr), 0, (hipStream_t)stream,
                           ix->d_bits, ix->a_pad, ix->n_words, ix->d_pid, ix->d_vals, ix->d_nval, pieces, masks, n_pieces, compat, ix->w64);
    }
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
#ifdef HGX_LAB
#include "lab/hgx_fused_lab.inc"        // round 2's fused gene-level form (FusedArgs, fused_claim): lab build only
#else
struct FusedArgs;                        // (the hooks below compile away)
#endif

// NP = number of counter planes: 2 when the pair has <= 3 refs (nearly every pair), 4 up to 15, 8 up to 255
// arg-max set of the bit-sliced counters under the level mask, its row hash, and the stores
template <int KW, int NP, bool FUSE = false>
__device__ __forceinline__ void emit_class(const uint64_t (&plane)[NP][KW], int w64, const uint64_t *__restrict__ mask,
                                           uint64_t *__restrict__ out_row, uint64_t *__restrict__ out_hash, int lane,
                                           const FusedArgs *fa = nullptr, long pair = 0) {
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
            if (!FUSE && out_row) out_row[w] = cand[s];
            h += word_hash(cand[s], w);
            nz = nz || cand[s] != 0;
        }
    }
#ifdef HGX_LAB
    if (FUSE) {
        h = wave_sum_u64(h);
        fused_claim<KW>(cand, w64, finish_hash(h, __any(nz)), pair, *fa, lane);
        return;
    }
#endif
    if (out_hash) {
        h = wave_sum_u64(h);
        const bool any_nz = __any(nz);
        if (lane == 0) *out_hash = finish_hash(h, any_nz);
    }
}

template <int KW, int NP, bool FUSE = false>
__device__ __forceinline__ void class_for_level(const uint64_t *__restrict__ compat, int w64, const uint32_t *__restrict__ refs,
                                                int r0, int r1, uint32_t level, const uint64_t *__restrict__ mask,
                                                uint64_t *__restrict__ out_row, uint64_t *__restrict__ out_hash, int lane,
                                                const FusedArgs *fa = nullptr, long pair = 0) {
    uint64_t plane[NP][KW];
#pragma unroll
    for (int k = 0; k < NP; ++k)
#pragma unroll
        for (int s = 0; s < KW; ++s) plane[k][s] = 0;
    // software pipeline: the next matching ref's row is in flight while the current one is added
    auto next_ref = [&](int r) {
        while (r < r1 && ((uint32_t)__builtin_amdgcn_readfirstlane(refs[r]) >> 31) != level) ++r;
        return r;
    };
    auto load_row = [&](int r, uint64_t (&x)[KW]) {
        const uint64_t *row = compat + (size_t)((uint32_t)__builtin_amdgcn_readfirstlane(refs[r]) & 0x7fffffffu) * w64;
#pragma unroll
        for (int s = 0; s < KW; ++s) x[s] = (lane + 64 * s < w64) ? row[lane + 64 * s] : 0ull;
    };
    uint64_t cur[KW], nxt[KW];
    int r = next_ref(r0);
    if (r < r1) load_row(r, cur);
    while (r < r1) {
        const int rn = next_ref(r + 1);
        if (rn < r1) load_row(rn, nxt);
        uint64_t carry[KW];
#pragma unroll
        for (int s = 0; s < KW; ++s) carry[s] = cur[s];
#pragma unroll
        for (int k = 0; k < NP; ++k)
#pragma unroll
            for (int s = 0; s < KW; ++s) {
                const uint64_t t = plane[k][s] & carry[s];
                plane[k][s] ^= carry[s];
                carry[s] = t;
            }
#pragma unroll
        for (int s = 0; s < KW; ++s) cur[s] = nxt[s];
        r = rn;
    }
    emit_class<KW, NP, FUSE>(plane, w64, mask, out_row, out_hash, lane, fa, pair);
}

// More than 255 refs at a level (a pair whose mates have hundreds of alternative alignments: long STR alleles): 16-plane counters
// (counts up to 65535) would not fit the register file for every 64-allele slab of the row at once, so the row is walked slab
// by slab, twice: pass 0 finds the largest count over the level's alleles (top-down scan per slab, maximum over the slabs),
// pass 1 recounts and keeps the alleles whose counter equals it bit for bit.  Same class, same hash as the register-resident form.
__device__ __noinline__ void class_for_level_wide(const uint64_t *__restrict__ compat, int w64, const uint32_t *__restrict__ refs,
                                                  int r0, int r1, uint32_t level, const uint64_t *__restrict__ mask,
                                                  uint64_t *__restrict__ out_row, uint64_t *__restrict__ out_hash, int lane,
                                                  const FusedArgs *fa = nullptr, long pair = 0) {
    constexpr int NP = 16;
#ifdef HGX_LAB
    if (fa) out_row = fa->rows + (size_t)pair * w64;      // (fused: every such row is stored -- the path is rare -- and compared from memory)
#endif
    const int kw = (w64 + 63) / 64;
    uint32_t gmax = 0;
    uint64_t h = 0;
    bool nz = false;
    for (int pass = 0; pass < 2; ++pass)
        for (int s = 0; s < kw; ++s) {
            const int w = lane + 64 * s;
            const bool live = w < w64;
            uint64_t plane[NP];
#pragma unroll
            for (int k = 0; k < NP; ++k) plane[k] = 0;
            for (int r = r0; r < r1; ++r) {
                const uint32_t ref = (uint32_t)__builtin_amdgcn_readfirstlane(refs[r]);
                if ((ref >> 31) != level) continue;
                uint64_t carry = live ? compat[(size_t)(ref & 0x7fffffffu) * w64 + w] : 0ull;
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    const uint64_t t = plane[k] & carry;
                    plane[k] ^= carry;
                    carry = t;
                }
            }
            uint64_t cand = live ? mask[w] : 0ull;
            if (pass == 0) {
                uint32_t v = 0;
#pragma unroll
                for (int k = NP - 1; k >= 0; --k) {
                    const uint64_t t = cand & plane[k];
                    if (__any(t != 0)) { cand = t; v |= 1u << k; }
                }
                gmax = max(gmax, v);                             // wave-uniform
            } else {
#pragma unroll
                for (int k = 0; k < NP; ++k) cand &= ((gmax >> k) & 1u) ? plane[k] : ~plane[k];
                if (live) {
#ifdef HGX_LAB
                    if (fa) __hip_atomic_store((unsigned long long *)&out_row[w], (unsigned long long)cand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else
#endif
                    if (out_row) out_row[w] = cand;
                    h += word_hash(cand, w);
                    nz = nz || cand != 0;
                }
            }
        }
#ifdef HGX_LAB
    if (fa) {
        h = wave_sum_u64(h);
        const uint64_t key = finish_hash(h, __any(nz));
        if (key == HGX_EMPTY_KEY) { if (lane == 0) fa->slot_of[pair] = 0xFFFFFFFFu; return; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int won;
        const uint32_t slot = fused_find_slot(*fa, key, lane, won);
        if (lane == 0) fa->slot_of[pair] = slot;
        if (won) { if (lane == 0) __hip_atomic_store(&fa->rep[slot], (uint32_t)pair, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
        const uint32_t r = fused_wait_rep(*fa, slot, lane);
        bool diff = false;
        for (int w = lane; w < w64; w += 64)
            diff = diff || __hip_atomic_load((unsigned long long *)&fa->rows[(size_t)r * w64 + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) !=
                               __hip_atomic_load((unsigned long long *)&out_row[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__any(diff) && lane == 0) atomicOr(fa->bad, 1);
        return;
    }
#endif
    if (out_hash) {
        h = wave_sum_u64(h);
        const bool any_nz = __any(nz);
        if (lane == 0) *out_hash = finish_hash(h, any_nz);
    }
}

// SEL: rows for a selection of pairs (one representative per distinct ref list, hgx_level_classes) instead of every pair
template <int KW, bool SEL>
__global__ __launch_bounds__(256) void k_pair_classes(const uint64_t *__restrict__ compat, int w64,
                                                      const int32_t *__restrict__ pair_off, const uint32_t *__restrict__ refs,
                                                      const int64_t *__restrict__ sel,
                                                      int n_pairs, const uint64_t *__restrict__ exon_mask,
                                                      const uint64_t *__restrict__ gene_mask, uint64_t *__restrict__ exon_bits,
                                                      uint64_t *__restrict__ gene_bits, uint64_t *__restrict__ exon_hash,
                                                      uint64_t *__restrict__ gene_hash) {
    const int lane = threadIdx.x & 63;
    // `out` = output row; with a selection (hgx_level_classes: one representative pair per distinct ref list) the input
    // pair is sel[out]
    const long out = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (out >= n_pairs) return;
    const long pair = SEL ? (long)sel[out] : out;
    const int r0 = __builtin_amdgcn_readfirstlane(pair_off[pair]);
    const int r1 = __builtin_amdgcn_readfirstlane(pair_off[pair + 1]);
    // counts never exceed the pair's number of refs: pick the narrowest counter that holds it (wave-uniform)
    const int n_refs = r1 - r0;
#define HGX_LEVELS(NP_)                                                                                                          \
    do {                                                                                                                         \
        if (exon_bits || exon_hash)                                                                                              \
            class_for_level<KW, NP_>(compat, w64, refs, r0, r1, 0u, exon_mask, exon_bits ? exon_bits + (size_t)out * w64 : nullptr, \
                                     exon_hash ? exon_hash + out : nullptr, lane);                                              \
        if (gene_bits || gene_hash)                                                                                              \
            class_for_level<KW, NP_>(compat, w64, refs, r0, r1, 1u, gene_mask, gene_bits ? gene_bits + (size_t)out * w64 : nullptr, \
                                     gene_hash ? gene_hash + out : nullptr, lane);                                              \
    } while (0)
    if (n_refs <= 3) HGX_LEVELS(2);
    else if (n_refs <= 15) HGX_LEVELS(4);
    else if (n_refs <= 255) HGX_LEVELS(8);
    else {
        if (exon_bits || exon_hash)
            class_for_level_wide(compat, w64, refs, r0, r1, 0u, exon_mask, exon_bits ? exon_bits + (size_t)out * w64 : nullptr,
                                 exon_hash ? exon_hash + out : nullptr, lane);
        if (gene_bits || gene_hash)
            class_for_level_wide(compat, w64, refs, r0, r1, 1u, gene_mask, gene_bits ? gene_bits + (size_t)out * w64 : nullptr,
                                 gene_hash ? gene_hash + out : nullptr, lane);
    }
#undef HGX_LEVELS
}

// One LEVEL, TWO pairs per wavefront (the launches of the typing path).  A wavefront per pair is a chain of dependent round trips --
// offsets -> refs -> piece rows -- and hides them only through occupancy: 0.28 ms for 500 k pairs at eight waves per SIMD is 4.6 us
// per wavefront, whatever is taken out of it (row stores, dependent ref loads, the hash: DESIGN.md 5.2).  Here the refs of both pairs
// sit in the lanes (one vector load each, picked per level with a ballot and broadcast with v_readlane) and every step of the walk
// has a row of EACH pair in flight, plus the next one of each.
template <int KW, int NP, bool FUSE = false>
__device__ __forceinline__ void two_classes(const uint64_t *__restrict__ compat, int w64, uint32_t ref_a, uint32_t ref_b,
                                            uint64_t todo_a, uint64_t todo_b, const uint64_t *__restrict__ mask,
                                            uint64_t *__restrict__ row_a, uint64_t *__restrict__ row_b,
                                            uint64_t *__restrict__ hash_a, uint64_t *__restrict__ hash_b, bool has_b, int lane,
                                            const FusedArgs *fa = nullptr, long pair_a = 0, long pair_b = 0) {
    uint64_t pa[NP][KW], pb[NP][KW];
#pragma unroll
    for (int k = 0; k < NP; ++k)
#pragma unroll
        for (int s = 0; s < KW; ++s) { pa[k][s] = 0; pb[k][s] = 0; }
    auto load_row = [&](uint32_t myref, uint64_t todo, uint64_t (&x)[KW]) {
        const uint64_t *row = compat + (size_t)((uint32_t)__builtin_amdgcn_readlane((int)myref, __builtin_ctzll(todo)) & 0x7fffffffu) * w64;
#pragma unroll
        for (int s = 0; s < KW; ++s) x[s] = (lane + 64 * s < w64) ? row[lane + 64 * s] : 0ull;
    };
    auto ripple = [&](uint64_t (&plane)[NP][KW], const uint64_t (&x)[KW]) {
        uint64_t carry[KW];
#pragma unroll
        for (int s = 0; s < KW; ++s) carry[s] = x[s];
#pragma unroll
        for (int k = 0; k < NP; ++k)
#pragma unroll
            for (int s = 0; s < KW; ++s) {
                const uint64_t t = plane[k][s] & carry[s];
                plane[k][s] ^= carry[s];
                carry[s] = t;
            }
    };
    uint64_t ca[KW], cb[KW], na[KW], nb[KW];
#pragma unroll
    for (int s = 0; s < KW; ++s) { ca[s] = 0; cb[s] = 0; na[s] = 0; nb[s] = 0; }
    if (todo_a) load_row(ref_a, todo_a, ca);
    if (todo_b) load_row(ref_b, todo_b, cb);
    while (todo_a | todo_b) {
        const bool have_a = todo_a != 0, have_b = todo_b != 0;
        todo_a &= todo_a - 1;                                  // (0 & anything = 0: an exhausted pair stays exhausted)
        todo_b &= todo_b - 1;
        if (todo_a) load_row(ref_a, todo_a, na);
        if (todo_b) load_row(ref_b, todo_b, nb);
        if (have_a) ripple(pa, ca);
        if (have_b) ripple(pb, cb);
#pragma unroll
        for (int s = 0; s < KW; ++s) { ca[s] = na[s]; cb[s] = nb[s]; }
    }
    emit_class<KW, NP, FUSE>(pa, w64, mask, row_a, hash_a, lane, fa, pair_a);
    if (has_b) emit_class<KW, NP, FUSE>(pb, w64, mask, row_b, hash_b, lane, fa, pair_b);
}

template <int KW, bool SEL>
__global__ __launch_bounds__(256) void k_pair_classes_x2(const uint64_t *__restrict__ compat, int w64,
                                                         const int32_t *__restrict__ pair_off, const uint32_t *__restrict__ refs,
                                                         const int64_t *__restrict__ sel, int n_pairs, uint32_t level,
                                                         const uint64_t *__restrict__ mask, uint64_t *__restrict__ bits,
                                                         uint64_t *__restrict__ hash) {
    const int lane = threadIdx.x & 63;
    const long out_a = (((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 2, out_b = out_a + 1;
    if (out_a >= n_pairs) return;
    const bool has_b = out_b < n_pairs;
    const long pair_a = SEL ? (long)sel[out_a] : out_a;
    const long pair_b = has_b ? (SEL ? (long)sel[out_b] : out_b) : pair_a;
    const int r0a = __builtin_amdgcn_readfirstlane(pair_off[pair_a]), r1a = __builtin_amdgcn_readfirstlane(pair_off[pair_a + 1]);
    const int r0b = __builtin_amdgcn_readfirstlane(pair_off[pair_b]), r1b = __builtin_amdgcn_readfirstlane(pair_off[pair_b + 1]);
    const int n_a = r1a - r0a, n_b = has_b ? r1b - r0b : 0;
    uint64_t *row_a = bits ? bits + (size_t)out_a * w64 : nullptr, *row_b = bits ? bits + (size_t)out_b * w64 : nullptr;
    uint64_t *hash_a = hash ? hash + out_a : nullptr, *hash_b = hash ? hash + out_b : nullptr;
    if (max(n_a, n_b) > 15) {        // many refs (rare): 8-plane counters up to 255 refs, the slab-wise 16-plane form beyond; one pair after the other
        if (n_a > 255) class_for_level_wide(compat, w64, refs, r0a, r1a, level, mask, row_a, hash_a, lane);
        else class_for_level<KW, 8>(compat, w64, refs, r0a, r1a, level, mask, row_a, hash_a, lane);
        if (has_b) {
            if (n_b > 255) class_for_level_wide(compat, w64, refs, r0b, r1b, level, mask, row_b, hash_b, lane);
            else class_for_level<KW, 8>(compat, w64, refs, r0b, r1b, level, mask, row_b, hash_b, lane);
        }
        return;
    }
    const uint32_t ref_a = lane < n_a ? refs[r0a + lane] : 0u, ref_b = lane < n_b ? refs[r0b + lane] : 0u;
    const uint64_t todo_a = __ballot(lane < n_a && (ref_a >> 31) == level), todo_b = __ballot(lane < n_b && (ref_b >> 31) == level);
    const int n_max = max(n_a, n_b);             // counts never exceed the number of refs: narrowest counters that hold them
    if (n_max <= 3) two_classes<KW, 2>(compat, w64, ref_a, ref_b, todo_a, todo_b, mask, row_a, row_b, hash_a, hash_b, has_b, lane);
    else two_classes<KW, 4>(compat, w64, ref_a, ref_b, todo_a, todo_b, mask, row_a, row_b, hash_a, hash_b, has_b, lane);
}

#ifdef HGX_LAB
#include "lab/hgx_fused_kernel_lab.inc"        // k_pair_classes_fused + hgx_pair_classes_fused_launch: lab build only
#endif

// sel: see k_pair_classes (NULL = every pair in order)
int hgx_pair_classes_sel(const hgx_index *ix, const uint64_t *compat, const int32_t *pair_off, const uint32_t *refs,
                         const int64_t *sel, int32_t n_pairs, uint64_t *eb, uint64_t *gb, uint64_t *eh,
                         uint64_t *gh, hipStream_t st) {
    ARGCHK(ix && n_pairs >= 0);
    if (n_pairs == 0) return HGX_OK;
    ARGCHK(compat && pair_off && refs);
    const long blocks = ((long)n_pairs + 3) / 4;
    const int kw = (ix->w64 + 63) / 64;
    const bool exon = eb || eh, gene = gb || gh;
    if (exon != gene && kw <= 8 && !HGX_LAB_SWITCH("pair_x1")) {       // one level: two pairs per wavefront
        const long blocks2 = ((long)n_pairs + 7) / 8;
        const uint32_t level = gene ? 1u : 0u;
        const uint64_t *mask = gene ? ix->d_gene_mask : ix->d_exon_mask;
        uint64_t *bits = gene ? gb : eb, *hash = gene ? gh : eh;
#define LAUNCH_X2(KW_)                                                                                                        \
    do {                                                                                                                      \
        if (sel)                                                                                                              \
            hipLaunchKernelGGL((k_pair_classes_x2<KW_, true>), dim3((unsigned)blocks2), dim3(256), 0, st, compat, ix->w64, pair_off, \
                               refs, sel, n_pairs, level, mask, bits, hash);                                                  \
        else                                                                                                                  \
            hipLaunchKernelGGL((k_pair_classes_x2<KW_, false>), dim3((unsigned)blocks2), dim3(256), 0, st, compat, ix->w64, pair_off, \
                               refs, sel, n_pairs, level, mask, bits, hash);                                                  \
    } while (0)
        if (kw <= 1) LAUNCH_X2(1);
        else if (kw <= 2) LAUNCH_X2(2);
        else if (kw <= 4) LAUNCH_X2(4);
        else LAUNCH_X2(8);
#undef LAUNCH_X2
        HIPCHK(hipGetLastError());
        return HGX_OK;
    }
#define LAUNCH_PC(KW_)                                                                                                        \
    do {                                                                                                                      \
        if (sel)                                                                                                              \
            hipLaunchKernelGGL((k_pair_classes<KW_, true>), dim3((unsigned)blocks), dim3(256), 0, st, compat, ix->w64, pair_off, \
                               refs, sel, n_pairs, ix->d_exon_mask, ix->d_gene_mask, eb, gb, eh, gh);                        \
        else                                                                                                                  \
            hipLaunchKernelGGL((k_pair_classes<KW_, false>), dim3((unsigned)blocks), dim3(256), 0, st, compat, ix->w64, pair_off, \
                               refs, sel, n_pairs, ix->d_exon_mask, ix->d_gene_mask, eb, gb, eh, gh);                        \
    } while (0)
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

extern "C" int hgx_pair_classes(const hgx_index *ix, const uint64_t *compat, const int32_t *pair_off, const uint32_t *refs,
                                int32_t n_pairs, uint64_t *eb, uint64_t *gb, uint64_t *eh, uint64_t *gh, void *stream) {
    return hgx_pair_classes_sel(ix, compat, pair_off, refs, nullptr, n_pairs, eb, gb, eh, gh, (hipStream_t)stream);
}

extern "C" int hgx_score_pairs(const hgx_index *ix, const hgx_piece *pieces, const uint32_t *masks, int32_t n_pieces,
                               const int32_t *pair_off, const uint32_t *refs, int32_t n_pairs, uint64_t *compat,
                               uint64_t *eb, uint64_t *gb, uint64_t *eh, uint64_t *gh, void *stream) {
    int rc = hgx_piece_compat(ix, pieces, masks, n_pieces, compat, stream);
    if (rc) return rc;
    return hgx_pair_classes(ix, compat, pair_off, refs, n_pairs, eb, gb, eh, gh, stream);
}

