// hgx_dedup.hip -- class dedup (8a-7), bit-matrix transpose and Gene_counts for libhgx (gfx950).
#include <algorithm>
#include <vector>

#include "hgx_common.hpp"

// ------------------------------------------------------------------------------------------------
// 8a-7 class dedup: hash (optional AND mask) -> hash table insert -> first rows numbered by a single-pass scan ->
// exact verify -> gather (every kernel here is this library's own: no CUB / rocPRIM on the path).
// ------------------------------------------------------------------------------------------------

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

// ------------------------------------------------------------------------------------------------
// Exclusive prefix sum of n uint32 values in ONE launch (chained scan with decoupled look-back).  A workgroup takes the next
// tile in START order (ticket), so every tile it waits for belongs to a workgroup that is already running: no deadlock
// whatever the dispatch order.  Tile state = one 64-bit word (status in the top two bits: 1 = the tile's own sum, 2 = the
// inclusive prefix up to and including the tile), published and polled with device-scope atomics -- one word, so no fence.
// `scratch` = [tiles] state words + 1 ticket word, zeroed before the launch; *total_out = sum of all values.
// ------------------------------------------------------------------------------------------------
#define SCAN_T 1024
#define SCAN_TILE (4 * SCAN_T)
__global__ __launch_bounds__(SCAN_T) void k_scan_u32(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, long n,
                                                     unsigned long long *__restrict__ state, uint32_t *__restrict__ ticket,
                                                     uint32_t *__restrict__ total_out) {
    __shared__ uint32_t s_tile;
    __shared__ uint32_t s_wave[SCAN_T / 64];
    __shared__ unsigned long long s_prefix;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_tile = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t tile = s_tile;
    const long i0 = (long)tile * SCAN_TILE + 4 * tid;
    uint32_t v[4];
    if (i0 + 3 < n) {
        const uint4 q = *(const uint4 *)(in + i0);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = i0 + k < n ? in[i0 + k] : 0u;
    }
    const uint32_t mine = v[0] + v[1] + v[2] + v[3];
    uint32_t incl = mine;                                   // inclusive scan of the threads' sums inside the wavefront
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
    }
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    if (wv == 0) {
        uint32_t w = lane < SCAN_T / 64 ? s_wave[lane] : 0u, wi = w;
#pragma unroll
        for (int d = 1; d < SCAN_T / 64; d <<= 1) {
            const uint32_t up = __shfl_up(wi, d, 64);
            if (lane >= d) wi += up;
        }
        if (lane < SCAN_T / 64) s_wave[lane] = wi - w;      // exclusive offsets of the wavefronts
        if (lane == SCAN_T / 64 - 1) {
            const unsigned long long total = wi;
            unsigned long long before = 0;
            if (tile == 0) {
                __hip_atomic_store(&state[0], (2ull << 62) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                __hip_atomic_store(&state[tile], (1ull << 62) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (long j = (long)tile - 1;; --j) {
                    unsigned long long st;
                    do st = __hip_atomic_load(&state[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); while ((st >> 62) == 0);
                    before += st & ((1ull << 62) - 1);
                    if ((st >> 62) == 2) break;
                }
                __hip_atomic_store(&state[tile], (2ull << 62) | (before + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            s_prefix = before;
            if ((long)(tile + 1) * SCAN_TILE >= n && total_out) *total_out = (uint32_t)(before + total);
        }
    }
    __syncthreads();
    uint32_t run = (uint32_t)s_prefix + s_wave[wv] + (incl - mine);
    if (i0 + 3 < n) {
        uint4 o;
        o.x = run; run += v[0];
        o.y = run; run += v[1];
        o.z = run; run += v[2];
        o.w = run;
        *(uint4 *)(out + i0) = o;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i0 + k < n) out[i0 + k] = run;
            run += v[k];
        }
    }
}
static size_t scan_scratch_bytes(long n) { return ((size_t)((n + SCAN_TILE - 1) / SCAN_TILE) + 2) * 8; }
// out = exclusive prefix sums of in[0..n), *total_dev = their total; scratch from scan_scratch_bytes(n)
// (scratch_is_zero: the caller's own kernel has zeroed the scratch -- k_ht_init does it with the table -- so that the chain of a
// dedup carries no memset of its own)
static int scan_u32(const uint32_t *in, uint32_t *out, long n, void *scratch, uint32_t *total_dev, hipStream_t st, bool scratch_is_zero = false) {
    const long tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (!scratch_is_zero) HIPCHK(hipMemsetAsync(scratch, 0, scan_scratch_bytes(n), st));
    hipLaunchKernelGGL(k_scan_u32, dim3((unsigned)tiles), dim3(SCAN_T), 0, st, in, out, n, (unsigned long long *)scratch,
                       (uint32_t *)((unsigned long long *)scratch + tiles), total_dev);
    HIPCHK(hipGetLastError());
    return HGX_OK;
}

// ------------------------------------------------------------------------------------------------
// Hash-table form of the dedup (default).  Sorting 500 k 64-bit keys costs ~25 short library launches; the dict
// semantics the reference needs -- {class: count} in first-seen order -- fall out of one insert pass:
//   k_ht_insert   a workgroup of 1024 rows first merges equal keys in an LDS table (min row, summed weight; the hot
//                 classes hold a fifth of all rows, so un-merged global atomics would serialise on a few addresses),
//                 then one global insert (CAS on the key, min on the first row, add on the count) per distinct key;
//   k_ht_mark     flags the first row of every class; an exclusive scan of the flags over the ROWS is the class id in
//                 first-seen order (= Python dict order);
//   k_ht_finalize class id -> first row, count;   k_verify_ht: exact check of every row against its class' first row
//                 (streaming, original order);   k_ht_gather: class rows.
// min / add / CAS are order independent, so the result is deterministic.
// ------------------------------------------------------------------------------------------------
#define HT_NONE 0xFFFFFFFFu

__device__ __forceinline__ uint32_t ht_global_insert(unsigned long long *keys, uint32_t tmask, uint64_t key) {
    // A slot never changes once it holds a key: a plain look (served by the L2, where the atomics execute) settles most inserts --
    // the hot classes are in the table after the first few workgroups -- and only an empty-looking slot costs a CAS.
    uint32_t h = (uint32_t)key & tmask;
    for (;;) {
        unsigned long long old = __hip_atomic_load(&keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == HGX_EMPTY_KEY) old = atomicCAS(&keys[h], (unsigned long long)HGX_EMPTY_KEY, (unsigned long long)key);
        if (old == HGX_EMPTY_KEY || old == key) return h;
        h = (h + 1) & tmask;
    }
}

template <int BS>
__global__ __launch_bounds__(BS) void k_ht_insert(const uint64_t *__restrict__ hash, const int64_t *__restrict__ weight, long n,
                                                    unsigned long long *__restrict__ keys, uint32_t *__restrict__ first,
                                                    unsigned long long *__restrict__ cnt, uint32_t tmask,
                                                    uint32_t *__restrict__ slot_of) {
    constexpr int SLOTS = 2 * BS;
    __shared__ unsigned long long lkey[SLOTS], lcnt[SLOTS];
    __shared__ uint32_t lmin[SLOTS], lg[SLOTS];
    const int tid = threadIdx.x;
    for (int s = tid; s < SLOTS; s += BS) { lkey[s] = HGX_EMPTY_KEY; lcnt[s] = 0; lmin[s] = HT_NONE; }
    __syncthreads();
    const long i = (long)blockIdx.x * BS + tid;
    const uint64_t key = i < n ? hash[i] : HGX_EMPTY_KEY;
    const bool valid = key != HGX_EMPTY_KEY;
    uint32_t ls = 0;
    if (valid) {
        ls = (uint32_t)(key >> 40) & (SLOTS - 1);
        for (;;) {
            const unsigned long long old = atomicCAS(&lkey[ls], (unsigned long long)HGX_EMPTY_KEY, (unsigned long long)key);
            if (old == HGX_EMPTY_KEY || old == key) break;
            ls = (ls + 1) & (SLOTS - 1);
        }
        atomicMin(&lmin[ls], (uint32_t)i);
        atomicAdd(&lcnt[ls], (unsigned long long)(weight ? weight[i] : 1));
    }
    __syncthreads();
    for (int s = tid; s < SLOTS; s += BS) {
        if (lkey[s] != HGX_EMPTY_KEY) {
            const uint32_t g = ht_global_insert(keys, tmask, lkey[s]);
            // (the first row only ever goes down: a value already below mine needs no atomic; a stale, larger one merely costs it)
            if (__hip_atomic_load(&first[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > lmin[s]) atomicMin(&first[g], lmin[s]);
            atomicAdd(&cnt[g], lcnt[s]);
            lg[s] = g;
        }
    }
    __syncthreads();
    if (i < n) slot_of[i] = valid ? lg[ls] : HT_NONE;
}
__global__ void k_ht_init(unsigned long long *__restrict__ keys, uint32_t *__restrict__ first, unsigned long long *__restrict__ cnt,
                          long T, uint32_t *__restrict__ flag, long n, uint32_t *__restrict__ meta,
                          unsigned long long *__restrict__ scan_state, long n_state) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < T) { keys[i] = HGX_EMPTY_KEY; first[i] = HT_NONE; cnt[i] = 0; }
    if (i < n) flag[i] = 0;
    if (i < 4) meta[i] = 0;
    if (i < n_state) scan_state[i] = 0;                 // (the tile states + ticket of the scan that numbers the classes)
}
__global__ void k_ht_mark(const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ first, long T,
                          uint32_t *__restrict__ is_first) {
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s < T && keys[s] != HGX_EMPTY_KEY) is_first[first[s]] = 1u;
}
__global__ void k_ht_finalize(const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ first,
                              const unsigned long long *__restrict__ cnt, long T, const uint32_t *__restrict__ rank,
                              int64_t *__restrict__ out_first, int64_t *__restrict__ out_count) {
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= T || keys[s] == HGX_EMPTY_KEY) return;
    const uint32_t c = rank[first[s]];
    out_first[c] = (int64_t)first[s];
    out_count[c] = (int64_t)cnt[s];
}
// Four rows per wavefront, their dependent loads (slot -> first row of the class -> row words) issued level by level: with one
// row per wave the kernel is a chain of three memory round trips per row and runs at a quarter of the bandwidth.
__global__ __launch_bounds__(256) void k_verify_ht(const uint64_t *__restrict__ rows, int w64, const uint64_t *__restrict__ mask,
                                                   const uint32_t *__restrict__ slot_of, const uint32_t *__restrict__ first, long n,
                                                   int *__restrict__ bad, uint32_t *__restrict__ bad_rows = nullptr,
                                                   uint32_t *__restrict__ n_bad = nullptr, const uint32_t *__restrict__ seg = nullptr) {
    // seg (many tasks in one dedup): rows of different segments never share a class -- their keys are salted with the segment,
    // and a row whose slot was founded by another segment's row (a 64-bit key collision) counts as different here
    constexpr int R = 4;
    const int lane = threadIdx.x & 63;
    const long r0 = (((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * R;
    if (r0 >= n) return;
    uint32_t sl[R], h[R];
#pragma unroll
    for (int k = 0; k < R; ++k) sl[k] = r0 + k < n ? slot_of[r0 + k] : HT_NONE;
#pragma unroll
    for (int k = 0; k < R; ++k) h[k] = sl[k] != HT_NONE ? first[sl[k]] : 0u;
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    const u64x2 *m = (const u64x2 *)mask;
    bool diff = false;
    if (seg) {
#pragma unroll
        for (int k = 0; k < R; ++k)
            if (sl[k] != HT_NONE && h[k] != (uint32_t)(r0 + k)) diff = diff || seg[r0 + k] != seg[h[k]];
    }
    for (int w = lane; w < w64 / 2; w += 64) {
        u64x2 x[R], y[R];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const bool live = sl[k] != HT_NONE && h[k] != (uint32_t)(r0 + k);
            x[k] = live ? ((const u64x2 *)(rows + (size_t)(r0 + k) * w64))[w] : u64x2{0, 0};
            y[k] = live ? ((const u64x2 *)(rows + (size_t)h[k] * w64))[w] : u64x2{0, 0};
        }
#pragma unroll
        for (int k = 0; k < R; ++k) {
            if (mask) { x[k] &= m[w]; y[k] &= m[w]; }
            diff = diff || x[k].x != y[k].x || x[k].y != y[k].y;
        }
    }
    if (!__any(diff)) return;
    // (never taken with honest keys) some row of the four shares its key with a DIFFERENT row: find out which
    if (lane == 0) atomicOr(bad, 1);
    if (!bad_rows) return;
    for (int k = 0; k < R; ++k) {
        if (sl[k] == HT_NONE || h[k] == (uint32_t)(r0 + k)) continue;
        bool d = seg && seg[r0 + k] != seg[h[k]];
        for (int w = lane; w < w64; w += 64) {
            uint64_t x = rows[(size_t)(r0 + k) * w64 + w], y = rows[(size_t)h[k] * w64 + w];
            if (mask) { x &= mask[w]; y &= mask[w]; }
            d = d || x != y;
        }
        if (__any(d) && lane == 0) bad_rows[atomicAdd(n_bad, 1u)] = (uint32_t)(r0 + k);
    }
}
// ---- key collisions are resolved, not reported ------------------------------------------------------------------------------
// A row that differs from the first row of its slot leaves the slot (its weight is taken back) and is inserted again under a
// salted hash of its content; it is then checked against ITS new slot's first row, and so on for a few rounds.  The slot's first row is never
// among the leavers (it equals itself), so whichever class owns the smallest row keeps the slot, and the first-seen order is
// recomputed afterwards.  With the 64-bit row hash this path never runs outside the forged-key tests.
__global__ __launch_bounds__(256) void k_fix_reinsert(const uint64_t *__restrict__ rows, int w64, const uint64_t *__restrict__ mask,
                                                      const uint32_t *__restrict__ bad_rows, uint32_t n_bad,
                                                      const int64_t *__restrict__ weight, int round,
                                                      unsigned long long *__restrict__ keys, uint32_t *__restrict__ first,
                                                      unsigned long long *__restrict__ cnt, uint32_t tmask,
                                                      uint32_t *__restrict__ slot_of, const uint32_t *__restrict__ seg = nullptr) {
    // one wavefront per row: the new key is a (salted) hash of the row's CONTENT -- rows that shared a key part ways at once
    const int lane = threadIdx.x & 63;
    const uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (t >= n_bad) return;
    const uint32_t i = bad_rows[t];
    uint64_t h = 0;
    for (int w = lane; w < w64; w += 64) {
        uint64_t x = rows[(size_t)i * w64 + w];
        if (mask) x &= mask[w];
        h += word_hash(x, w);
    }
    h = wave_sum_u64(h);
    if (lane != 0) return;
    uint64_t key = mix64(h + 0x9e3779b97f4a7c15ull * (uint64_t)round);
    if (seg) key = mix64(key ^ (0xd6e8feb86659fd93ull * (uint64_t)(seg[i] + 1u)));
    if (key == HGX_EMPTY_KEY) key = HGX_EMPTY_KEY - 1;
    const unsigned long long w = (unsigned long long)(weight ? weight[i] : 1);
    atomicAdd(&cnt[slot_of[i]], 0ull - w);
    const uint32_t g = ht_global_insert(keys, tmask, key);
    atomicMin(&first[g], i);
    atomicAdd(&cnt[g], w);
    slot_of[i] = g;
}
__global__ __launch_bounds__(256) void k_fix_verify(const uint64_t *__restrict__ rows, int w64, const uint64_t *__restrict__ mask,
                                                    const uint32_t *__restrict__ bad_rows, uint32_t n_bad,
                                                    const uint32_t *__restrict__ slot_of, const uint32_t *__restrict__ first,
                                                    uint32_t *__restrict__ still_bad, uint32_t *__restrict__ n_still,
                                                    const uint32_t *__restrict__ seg = nullptr) {
    const int lane = threadIdx.x & 63;
    const uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (t >= n_bad) return;
    const uint32_t i = bad_rows[t], f = first[slot_of[i]];
    if (f == i) return;
    bool diff = seg && seg[i] != seg[f];
    for (int w = lane; w < w64; w += 64) {
        uint64_t x = rows[(size_t)i * w64 + w], y = rows[(size_t)f * w64 + w];
        if (mask) { x &= mask[w]; y &= mask[w]; }
        diff = diff || x != y;
    }
    if (__any(diff) && lane == 0) still_bad[atomicAdd(n_still, 1u)] = i;
}
__global__ __launch_bounds__(256) void k_ht_gather(const uint64_t *__restrict__ rows, int w64, const uint64_t *__restrict__ mask,
                                                   const int64_t *__restrict__ out_first, int n_classes,
                                                   uint64_t *__restrict__ out_bits) {
    const int lane = threadIdx.x & 63;
    const long c = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (c >= n_classes) return;
    const uint64_t *src = rows + (size_t)out_first[c] * w64;
    for (int w = lane; w < w64; w += 64) {
        uint64_t x = src[w];
        if (mask) x &= mask[w];
        out_bits[(size_t)c * w64 + w] = x;
    }
}
__global__ __launch_bounds__(256) void k_ht_gather_slots(const uint64_t *__restrict__ rows, int w64, const uint64_t *__restrict__ mask,
                                                         const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ first,
                                                         const uint32_t *__restrict__ rank, long T, uint64_t *__restrict__ out_bits) {
    const int lane = threadIdx.x & 63;
    const long s = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (s >= T || keys[s] == HGX_EMPTY_KEY) return;
    const uint32_t f = first[s];
    const uint64_t *src = rows + (size_t)f * w64;
    uint64_t *dst = out_bits + (size_t)rank[f] * w64;
    for (int w = lane; w < w64; w += 64) {
        uint64_t x = src[w];
        if (mask) x &= mask[w];
        dst[w] = x;
    }
}

// keys of a dedup over MANY tasks' rows: a row's key is salted with its segment, so equal rows of different segments found
// different classes (the exact check, k_verify_ht, compares segments too)
__global__ void k_salt_keys(const uint64_t *__restrict__ in, const uint32_t *__restrict__ seg, long n, uint64_t *__restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t k = in[i];
    if (k != HGX_EMPTY_KEY) {
        k = mix64(k ^ (0xd6e8feb86659fd93ull * (uint64_t)(seg[i] + 1u)));
        if (k == HGX_EMPTY_KEY) k = HGX_EMPTY_KEY - 1;
    }
    out[i] = k;
}

static hgx_classes *new_classes(int32_t a_pad);
// small_blocks: insert with 256-row workgroups instead of 1024-row ones.  Alone the large ones win (4x fewer global atomics
// on the hot classes); beside a kernel that fills the chip (hgx_level_classes runs next to the gene level's per-pair rows)
// a 16-wave workgroup with 48 KB of LDS waits for a whole CU's worth of slots and the insert takes 3-4x longer.
static int dedup_hash_table(hgx_classes *cl, const uint64_t *rows, const uint64_t *keys_in, const int64_t *row_weight, long n,
                            int w64, const uint64_t *and_mask, hipStream_t st, bool small_blocks = false,
                            const uint32_t *seg = nullptr) {
    long T = 1024;
    while (T < 2 * n) T <<= 1;
    DevBuf b_keys, b_first, b_cnt, b_slot, b_flag, b_rank, b_tmp, b_meta, b_bad, b_salted;
    if (seg) {
        ALLOC(b_salted, (size_t)n * 8);
        hipLaunchKernelGGL(k_salt_keys, dim3(nblk(n, 256)), dim3(256), 0, st, keys_in, seg, n, b_salted.as<uint64_t>());
        keys_in = b_salted.as<uint64_t>();
    }
    ALLOC(b_keys, (size_t)T * 8); ALLOC(b_first, (size_t)T * 4); ALLOC(b_cnt, (size_t)T * 8);
    ALLOC(b_slot, (size_t)n * 4); ALLOC(b_flag, (size_t)n * 4); ALLOC(b_rank, (size_t)n * 4); ALLOC(b_meta, 16);
    ALLOC(b_bad, (size_t)n * 4);
    ALLOC(b_tmp, scan_scratch_bytes(n));
    // first rows of the classes -> flags -> class ids in first-seen order (single-pass scan), class count in meta[1]
    auto number_classes = [&](bool scratch_is_zero) -> int {
        hipLaunchKernelGGL(k_ht_mark, dim3(nblk(T, 256)), dim3(256), 0, st, b_keys.as<unsigned long long>(), b_first.as<uint32_t>(), T,
                           b_flag.as<uint32_t>());
        return scan_u32(b_flag.as<uint32_t>(), b_rank.as<uint32_t>(), n, b_tmp.p, b_meta.as<uint32_t>() + 1, st, scratch_is_zero);
    };
    uint32_t meta[4] = {0, 0, 0, 0};           // {some key collided, number of classes, number of collided rows, -}
    // rows that share a key with a different row: re-keyed and re-checked (see k_fix_reinsert); classes renumbered; meta refreshed
    auto resolve_collisions = [&]() -> int {
        DevBuf b_bad2;
        ALLOC(b_bad2, (size_t)n * 4);
        uint32_t *cur = b_bad.as<uint32_t>(), *nxt = b_bad2.as<uint32_t>();
        uint32_t nb = meta[2];
        for (int round = 1; round <= 8 && nb; ++round) {
            HIPCHK(hipMemsetAsync(b_meta.as<uint32_t>() + 2, 0, 4, st));
            hipLaunchKernelGGL(k_fix_reinsert, dim3(nblk(nb, 4)), dim3(256), 0, st, rows, w64, and_mask, cur, nb, row_weight, round,
                               b_keys.as<unsigned long long>(), b_first.as<uint32_t>(), b_cnt.as<unsigned long long>(), (uint32_t)(T - 1),
                               b_slot.as<uint32_t>(), seg);
            hipLaunchKernelGGL(k_fix_verify, dim3(nblk(nb, 4)), dim3(256), 0, st, rows, w64, and_mask, cur, nb, b_slot.as<uint32_t>(),
                               b_first.as<uint32_t>(), nxt, b_meta.as<uint32_t>() + 2, seg);
            HIPCHK(hipGetLastError());
            { int rc_ = hgx_d2h(meta, b_meta.p, 16, st); if (rc_) return rc_; }
            { int rc_ = hgx_sync(st); if (rc_) return rc_; }
            nb = meta[2];
            std::swap(cur, nxt);
        }
        if (nb) {
            hgx_set_error("64-bit class hash collisions not resolved after 8 re-keying rounds (%u rows)", nb);
            return HGX_ECOLLISION;
        }
        HIPCHK(hipMemsetAsync(b_flag.p, 0, (size_t)n * 4, st));
        { int rc_ = number_classes(false); if (rc_) return rc_; }
        HIPCHK(hipGetLastError());
        { int rc_ = hgx_d2h(meta, b_meta.p, 16, st); if (rc_) return rc_; }
        { int rc_ = hgx_sync(st); if (rc_) return rc_; }
        return HGX_OK;
    };
    // one launch instead of five memsets: table = empty, flags = 0, meta = {collision flag, number of classes} = 0
    hipLaunchKernelGGL(k_ht_init, dim3(nblk(std::max(T, n), 256)), dim3(256), 0, st, b_keys.as<unsigned long long>(),
                       b_first.as<uint32_t>(), b_cnt.as<unsigned long long>(), T, b_flag.as<uint32_t>(), n, b_meta.as<uint32_t>(),
                       b_tmp.as<unsigned long long>(), (long)(scan_scratch_bytes(n) / 8));
    if (small_blocks)
        hipLaunchKernelGGL(k_ht_insert<256>, dim3(nblk(n, 256)), dim3(256), 0, st, keys_in, row_weight, n, b_keys.as<unsigned long long>(),
                           b_first.as<uint32_t>(), b_cnt.as<unsigned long long>(), (uint32_t)(T - 1), b_slot.as<uint32_t>());
    else
        hipLaunchKernelGGL(k_ht_insert<1024>, dim3(nblk(n, 1024)), dim3(1024), 0, st, keys_in, row_weight, n, b_keys.as<unsigned long long>(),
                           b_first.as<uint32_t>(), b_cnt.as<unsigned long long>(), (uint32_t)(T - 1), b_slot.as<uint32_t>());
    { int rc_ = number_classes(true); if (rc_) return rc_; }
    // the exact check does not need the class count: queue it before the D2H that sizes the output
    hipLaunchKernelGGL(k_verify_ht, dim3(nblk((n + 3) / 4, 4)), dim3(256), 0, st, rows, w64, and_mask, b_slot.as<uint32_t>(),
                       b_first.as<uint32_t>(), n, b_meta.as<int>(), b_bad.as<uint32_t>(), b_meta.as<uint32_t>() + 2, seg);
    // Small inputs (the hand-off dedup: a few thousand gene classes) size the output for the worst case and finish in
    // ONE host round trip; large ones first learn the class count (the worst case would be the whole input again).
    const bool one_trip = n <= 65536;
    int n_alloc = (int)n;
    if (!one_trip) {
        { int rc_ = hgx_d2h(meta, b_meta.p, 16, st); if (rc_) return rc_; }
        { int rc_ = hgx_sync(st); if (rc_) return rc_; }
        HIPCHK(hipGetLastError());
        if (meta[2]) { int rc_ = resolve_collisions(); if (rc_) return rc_; }
        n_alloc = (int)meta[1];
        if (n_alloc == 0) return HGX_OK;
    }
    cl->d_bits = (decltype(cl->d_bits))hgx_pool_alloc((size_t)n_alloc * w64 * 8);
    cl->d_count = (decltype(cl->d_count))hgx_pool_alloc((size_t)n_alloc * 8);
    cl->d_first_row = (decltype(cl->d_first_row))hgx_pool_alloc((size_t)n_alloc * 8);
    if (!cl->d_bits || !cl->d_count || !cl->d_first_row) { hgx_set_error("device allocation failed"); return HGX_ENOMEM; }
    hipLaunchKernelGGL(k_ht_finalize, dim3(nblk(T, 256)), dim3(256), 0, st, b_keys.as<unsigned long long>(), b_first.as<uint32_t>(),
                       b_cnt.as<unsigned long long>(), T, b_rank.as<uint32_t>(), cl->d_first_row, cl->d_count);
    if (one_trip)      // gather per table slot (class id = rank of the slot's first row): needs no host-side class count
        hipLaunchKernelGGL(k_ht_gather_slots, dim3(nblk(T, 4)), dim3(256), 0, st, rows, w64, and_mask, b_keys.as<unsigned long long>(),
                           b_first.as<uint32_t>(), b_rank.as<uint32_t>(), T, cl->d_bits);
    else
        hipLaunchKernelGGL(k_ht_gather, dim3(nblk(n_alloc, 4)), dim3(256), 0, st, rows, w64, and_mask, cl->d_first_row, n_alloc, cl->d_bits);
    HIPCHK(hipGetLastError());
    if (one_trip) {
        { int rc_ = hgx_d2h(meta, b_meta.p, 16, st); if (rc_) return rc_; }
        { int rc_ = hgx_sync(st); if (rc_) return rc_; }
        if (meta[2]) {      // collided rows: after re-keying, first rows / counts / class rows are produced again
            { int rc_ = resolve_collisions(); if (rc_) return rc_; }
            hipLaunchKernelGGL(k_ht_finalize, dim3(nblk(T, 256)), dim3(256), 0, st, b_keys.as<unsigned long long>(), b_first.as<uint32_t>(),
                               b_cnt.as<unsigned long long>(), T, b_rank.as<uint32_t>(), cl->d_first_row, cl->d_count);
            hipLaunchKernelGGL(k_ht_gather_slots, dim3(nblk(T, 4)), dim3(256), 0, st, rows, w64, and_mask, b_keys.as<unsigned long long>(),
                               b_first.as<uint32_t>(), b_rank.as<uint32_t>(), T, cl->d_bits);
            HIPCHK(hipGetLastError());
            { int rc_ = hgx_sync(st); if (rc_) return rc_; }
        }
    } else {
        // large input: class count and collision flag are known since the first round trip; the finalize / gather kernels are
        // queued, and the scratch they read stays with the class set (freed with it) so that the caller's next launches
        // follow without another host sync
        cl->made_on = st;
        if (hipEventCreateWithFlags(&cl->ready, hipEventDisableTiming) == hipSuccess) (void)hipEventRecord(cl->ready, st);
        DevBuf *keep[] = {&b_keys, &b_first, &b_cnt, &b_slot, &b_flag, &b_rank, &b_tmp, &b_meta};
        for (int i = 0; i < 8; ++i) { cl->d_keep[i] = keep[i]->p; keep[i]->p = nullptr; }
        // (b_salted and b_bad were last read by kernels that completed before the round trip above)
    }
    const int n_runs = (int)meta[1];
    cl->n_classes = n_runs;
    return HGX_OK;
}

// On failure nothing is handed out: queued work is waited for, the half-built object goes back to the pool, *out = NULL.
template <class T, class Destroy> static int fail_clean(int rc, T **out, hipStream_t st, Destroy destroy) {
    if (rc != HGX_OK && out && *out) {
        (void)hipStreamSynchronize(st);
        destroy(*out);
        *out = nullptr;
    }
    return rc;
}

static int dedup_classes_impl(hgx_classes **out, const uint64_t *rows, const uint64_t *row_hash, const int64_t *row_weight,
                              int64_t n_rows, int32_t a_pad, const uint64_t *and_mask, void *stream, const uint32_t *seg = nullptr) {
    ARGCHK(out && n_rows >= 0 && a_pad > 0 && a_pad % 512 == 0);
    ARGCHK(n_rows < (1ll << 31));
    hipStream_t st = (hipStream_t)stream;
    const int w64 = a_pad / 64;
    hgx_classes *cl = new_classes(a_pad);
    *out = cl;
    if (n_rows == 0) return HGX_OK;
    ARGCHK(rows);
    const long n = n_rows;
    DevBuf b_hash;
    const uint64_t *keys_in = row_hash;
    if (!row_hash || and_mask) {   // hashes of masked rows must be recomputed
        ALLOC(b_hash, n * 8);
        hipLaunchKernelGGL(k_hash_rows, dim3(nblk(n, 4)), dim3(256), 0, st, rows, n, w64, and_mask, b_hash.as<uint64_t>());
        keys_in = b_hash.as<uint64_t>();
    }
    return dedup_hash_table(cl, rows, keys_in, row_weight, n, w64, and_mask, st, false, seg);
}

// ------------------------------------------------------------------------------------------------
// hgx_level_classes: pairs -> classes of one level without materialising a row per pair.
// A pair's class row is a function of its (ordered) list of piece refs at that level, and deep coverage repeats those
// lists: at HLA-A (1 M reads) 500 k pairs carry 134 k distinct exon-level lists.  So the pairs are first grouped by
// their ref LIST (8-byte keys, the same hash table as the row dedup, verified list against list), a row is computed
// for one representative pair per list, and the row dedup runs on those rows with the group sizes as weights.
// Same result as hgx_pair_classes + hgx_dedup_classes (classes in first-seen order, counts, first pair), with the
// 896-byte-per-pair row traffic divided by the repeat factor.
// ------------------------------------------------------------------------------------------------
// (the launch also empties the hash table, the flags, the meta words and the scan state of the grouping that follows -- what
// k_ht_init does for a dedup: one dispatch less in front of the insert)
struct HtInit { unsigned long long *keys; uint32_t *first; unsigned long long *cnt; long T; uint32_t *flag; uint32_t *meta;
                unsigned long long *scan_state; long n_state; };
__global__ void k_sig_keys(const int32_t *__restrict__ pair_off, const uint32_t *__restrict__ refs, int n_pairs, uint32_t level,
                           uint64_t *__restrict__ key, const uint32_t *__restrict__ seg, HtInit init) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long s = p; s < init.T; s += (long)gridDim.x * blockDim.x) { init.keys[s] = HGX_EMPTY_KEY; init.first[s] = HT_NONE; init.cnt[s] = 0; }
    if (p < n_pairs) init.flag[p] = 0;
    if (p < 4) init.meta[p] = 0;
    if (p < init.n_state) init.scan_state[p] = 0;
    if (p >= n_pairs) return;
    uint64_t h = 0x243f6a8885a308d3ull;
    uint32_t cnt = 0;
    for (int r = pair_off[p]; r < pair_off[p + 1]; ++r) {
        const uint32_t x = refs[r];
        if ((x >> 31) != level) continue;
        h = mix64(h + 0x9e3779b97f4a7c15ull * (uint64_t)((x & 0x7fffffffu) + 1u));     // order dependent on purpose: cheap, and a
        ++cnt;                                                                         // permuted list merely stays a separate group
    }
    h = mix64(h ^ cnt);
    if (seg) h = mix64(h ^ (0xd6e8feb86659fd93ull * (uint64_t)(seg[p] + 1u)));      // pairs of different tasks never share a group
    key[p] = h == HGX_EMPTY_KEY ? HGX_EMPTY_KEY - 1 : h;       // never the "dropped row" key: a pair without refs has a class (Q4)
}
// exact check of the grouping: every pair's ref list equals the list of its group's first pair
__global__ void k_sig_verify(const int32_t *__restrict__ pair_off, const uint32_t *__restrict__ refs, int n_pairs, uint32_t level,
                             const uint32_t *__restrict__ slot_of, const uint32_t *__restrict__ first, int *__restrict__ bad,
                             const uint32_t *__restrict__ seg = nullptr) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    const uint32_t f = first[slot_of[p]];
    if (f == (uint32_t)p) return;
    if (seg && seg[p] != seg[f]) { atomicOr(bad, 1); return; }
    int a = pair_off[p], b = pair_off[f];
    const int a1 = pair_off[p + 1], b1 = pair_off[f + 1];
    bool same = true;
    for (;;) {
        while (a < a1 && (refs[a] >> 31) != level) ++a;
        while (b < b1 && (refs[b] >> 31) != level) ++b;
        if (a >= a1 || b >= b1) { same = (a >= a1) == (b >= b1); break; }
        if (refs[a] != refs[b]) { same = false; break; }
        ++a; ++b;
    }
    if (!same) atomicOr(bad, 1);
}
__global__ void k_remap_first(int64_t *__restrict__ first_row, int n, const int64_t *__restrict__ sig_first) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n) first_row[c] = sig_first[first_row[c]];
}

static hgx_classes *new_classes(int32_t a_pad) {
    hgx_classes *cl = new hgx_classes();
    cl->a_pad = a_pad; cl->w64 = a_pad / 64; cl->n_classes = 0; cl->c64 = 0;
    cl->d_bits = nullptr; cl->d_count = nullptr; cl->d_first_row = nullptr; cl->d_bitsT = nullptr;
    return cl;
}

// stage 1: the grouping.  Needs only the pair -> ref lists, not the piece bitsets: callers queue it on its own stream
// beside hgx_piece_compat (hgx_group_pairs), or let hgx_level_classes do both stages in sequence.
struct hgx_groups {
    int32_t n_pairs = 0, level = 0;
    int64_t n_groups = 0;              // 0 with n_pairs > 0: key collision between different lists (never seen) -> per-pair form
    int64_t *d_first = nullptr;        // [n_groups] first pair of every group, groups in first-seen order
    int64_t *d_count = nullptr;        // [n_groups] pairs per group
    // hgx_group_pairs only queues; the first call that needs the group count completes it (same host thread)
    hipStream_t made_on = nullptr;
    bool finished = true;
    uint32_t meta[4] = {0, 0, 0, 0};   // {collision flag, number of groups}
    void *scratch[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};

static int groups_finish(hgx_groups *g) {
    if (g->finished) return HGX_OK;
    int rc = hgx_sync(g->made_on);      // delivers the staged meta words
    for (void *&p : g->scratch) { hgx_pool_free(p); p = nullptr; }
    g->finished = true;
    if (rc) return rc;
    g->n_groups = g->meta[0] ? 0 : (int64_t)g->meta[1];
    if (hgx_test_switch("test_group_collision")) g->n_groups = 0;      // test aid: take the per-pair fallback of a list-key collision
    return HGX_OK;
}

extern "C" int hgx_groups_destroy(hgx_groups *g) {
    if (!g) return HGX_OK;
    (void)groups_finish(g);
    hgx_pool_free(g->d_first); hgx_pool_free(g->d_count);
    delete g;
    return HGX_OK;
}
extern "C" int hgx_groups_dims(hgx_groups *g, int64_t *n_groups, int32_t *n_pairs) {
    ARGCHK(g);
    { int rc_ = groups_finish(g); if (rc_) return rc_; }
    if (n_groups) *n_groups = g->n_groups;
    if (n_pairs) *n_pairs = g->n_pairs;
    return HGX_OK;
}

static int group_pairs_impl(hgx_groups **out, const int32_t *pair_off, const uint32_t *refs, int32_t n_pairs, int32_t level,
                            void *stream, const uint32_t *seg = nullptr) {
    ARGCHK(out && n_pairs >= 0 && (level == 0 || level == 1));
    hipStream_t st = (hipStream_t)stream;
    hgx_groups *g = new hgx_groups();
    g->n_pairs = n_pairs; g->level = level; g->made_on = st;
    *out = g;
    if (n_pairs == 0) return HGX_OK;
    ARGCHK(pair_off && refs);
    const long n = n_pairs;
    g->d_first = (int64_t *)hgx_pool_alloc((size_t)n * 8);
    g->d_count = (int64_t *)hgx_pool_alloc((size_t)n * 8);
    if (!g->d_first || !g->d_count) { hgx_set_error("device allocation failed"); return HGX_ENOMEM; }
    long T = 1024;
    while (T < 2 * n) T <<= 1;
    DevBuf b_key, b_keys, b_first, b_cnt, b_slot, b_flag, b_rank, b_tmp, b_meta;
    ALLOC(b_key, (size_t)n * 8);
    ALLOC(b_keys, (size_t)T * 8); ALLOC(b_first, (size_t)T * 4); ALLOC(b_cnt, (size_t)T * 8);
    ALLOC(b_slot, (size_t)n * 4); ALLOC(b_flag, (size_t)n * 4); ALLOC(b_rank, (size_t)n * 4); ALLOC(b_meta, 16);
    ALLOC(b_tmp, scan_scratch_bytes(n));
    const HtInit init{b_keys.as<unsigned long long>(), b_first.as<uint32_t>(), b_cnt.as<unsigned long long>(), T, b_flag.as<uint32_t>(),
                      b_meta.as<uint32_t>(), b_tmp.as<unsigned long long>(), (long)(scan_scratch_bytes(n) / 8)};
    hipLaunchKernelGGL(k_sig_keys, dim3(nblk(std::max<long>(n, 1024), 256)), dim3(256), 0, st, pair_off, refs, n_pairs, (uint32_t)level, b_key.as<uint64_t>(),
                       seg, init);
    hipLaunchKernelGGL(k_ht_insert<256>, dim3(nblk(n, 256)), dim3(256), 0, st, b_key.as<uint64_t>(), (const int64_t *)nullptr, n,
                       b_keys.as<unsigned long long>(), b_first.as<uint32_t>(), b_cnt.as<unsigned long long>(), (uint32_t)(T - 1),
                       b_slot.as<uint32_t>());
    hipLaunchKernelGGL(k_ht_mark, dim3(nblk(T, 256)), dim3(256), 0, st, b_keys.as<unsigned long long>(), b_first.as<uint32_t>(), T,
                       b_flag.as<uint32_t>());
    { int rc_ = scan_u32(b_flag.as<uint32_t>(), b_rank.as<uint32_t>(), n, b_tmp.p, b_meta.as<uint32_t>() + 1, st, true); if (rc_) return rc_; }
    hipLaunchKernelGGL(k_sig_verify, dim3(nblk(n, 256)), dim3(256), 0, st, pair_off, refs, n_pairs, (uint32_t)level,
                       b_slot.as<uint32_t>(), b_first.as<uint32_t>(), b_meta.as<int>(), seg);
    // group id (first-seen order) -> first pair, group size; sized for the worst case: no host-side group count needed yet
    hipLaunchKernelGGL(k_ht_finalize, dim3(nblk(T, 256)), dim3(256), 0, st, b_keys.as<unsigned long long>(), b_first.as<uint32_t>(),
                       b_cnt.as<unsigned long long>(), T, b_rank.as<uint32_t>(), g->d_first, g->d_count);
    HIPCHK(hipGetLastError());
    { int rc_ = hgx_d2h(g->meta, b_meta.p, 16, st); if (rc_) return rc_; }
    // everything is queued; the scratch stays with the group set until groups_finish
    DevBuf *keep[] = {&b_key, &b_keys, &b_first, &b_cnt, &b_slot, &b_flag, &b_rank, &b_tmp, &b_meta};
    for (int i = 0; i < 9; ++i) { g->scratch[i] = keep[i]->p; keep[i]->p = nullptr; }
    g->finished = false;
    return HGX_OK;
}

extern "C" int hgx_group_pairs(hgx_groups **out, const int32_t *pair_off, const uint32_t *refs, int32_t n_pairs, int32_t level,
                               void *stream) {
    if (out) *out = nullptr;
    return fail_clean(group_pairs_impl(out, pair_off, refs, n_pairs, level, stream), out, (hipStream_t)stream, hgx_groups_destroy);
}
// the pairs of MANY tasks at once (hgx_type_many): pair_seg[p] = task of pair p; a group never spans two tasks
int hgx_group_pairs_seg(hgx_groups **out, const int32_t *pair_off, const uint32_t *refs, int32_t n_pairs, int32_t level,
                        const uint32_t *pair_seg, void *stream) {
    if (out) *out = nullptr;
    return fail_clean(group_pairs_impl(out, pair_off, refs, n_pairs, level, stream, pair_seg), out, (hipStream_t)stream, hgx_groups_destroy);
}

// stage 2: one row per group, then the row dedup weighted by the group sizes (st must be ordered behind hgx_piece_compat)
__global__ void k_seg_of_first(const int64_t *__restrict__ first, long n, const uint32_t *__restrict__ seg, uint32_t *__restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = seg[first[i]];
}

static int level_classes_grouped_impl(hgx_classes **out, const hgx_index *ix, const uint64_t *compat, const int32_t *pair_off,
                                      const uint32_t *refs, hgx_groups *g, uint64_t *rows_scratch, uint64_t *hash_scratch,
                                      void *stream, const uint32_t *pair_seg = nullptr) {
    ARGCHK(out && ix && g);
    { int rc_ = groups_finish(g); if (rc_) return rc_; }
    hipStream_t st = (hipStream_t)stream;
    hgx_classes *cl = new_classes(ix->a_pad);
    *out = cl;
    const long n = g->n_pairs;
    if (n == 0) return HGX_OK;
    ARGCHK(compat && pair_off && refs);
    const int w64 = ix->w64, level = g->level;
    const bool plain = g->n_groups == 0;
    const long n_rows = plain ? n : (long)g->n_groups;
    DevBuf b_rows, b_hash;      // read by kernels still queued when this returns: handed to the class set below
    uint64_t *rows = rows_scratch, *hash = hash_scratch;
    if (!rows) { ALLOC(b_rows, (size_t)n_rows * w64 * 8); rows = b_rows.as<uint64_t>(); }
    if (!hash) { ALLOC(b_hash, (size_t)n_rows * 8); hash = b_hash.as<uint64_t>(); }
    int rc = hgx_pair_classes_sel(ix, compat, pair_off, refs, plain ? nullptr : g->d_first, (int32_t)n_rows,
                                  level == 0 ? rows : nullptr, level == 1 ? rows : nullptr, level == 0 ? hash : nullptr,
                                  level == 1 ? hash : nullptr, st);
    if (rc) return rc;
    DevBuf b_rseg;                      // segment of every row: that of the group's first pair (the pairs themselves in the plain form)
    const uint32_t *row_seg = pair_seg;
    if (pair_seg && !plain) {
        ALLOC(b_rseg, (size_t)n_rows * 4);
        hipLaunchKernelGGL(k_seg_of_first, dim3(nblk(n_rows, 256)), dim3(256), 0, st, g->d_first, n_rows, pair_seg, b_rseg.as<uint32_t>());
        row_seg = b_rseg.as<uint32_t>();
    }
    rc = dedup_hash_table(cl, rows, hash, plain ? nullptr : g->d_count, n_rows, w64, nullptr, st, true, row_seg);
    if (rc) return rc;
    // (the row segments are read by k_verify_ht / the collision rounds only: all complete at the round trips inside the dedup)
    if (!plain && cl->n_classes > 0)
        hipLaunchKernelGGL(k_remap_first, dim3(nblk(cl->n_classes, 256)), dim3(256), 0, st, cl->d_first_row, cl->n_classes, g->d_first);
    HIPCHK(hipGetLastError());
    // consumers on other streams wait for `ready`; (re-)record it behind the last kernel queued here.  The group tables are
    // read by those kernels: the caller keeps `g` alive until the class set is consumed (or synchronises), see hgx.h.
    cl->made_on = st;
    if (!cl->ready && hipEventCreateWithFlags(&cl->ready, hipEventDisableTiming) != hipSuccess) cl->ready = nullptr;
    if (cl->ready) (void)hipEventRecord(cl->ready, st);
    else { int rc_ = hgx_sync(st); if (rc_) return rc_; }
    DevBuf *keep[] = {&b_rows, &b_hash};
    for (int i = 0; i < 2; ++i) { cl->d_keep[8 + i] = keep[i]->p; keep[i]->p = nullptr; }
    return HGX_OK;
}

extern "C" int hgx_level_classes_grouped(hgx_classes **out, const hgx_index *ix, const uint64_t *compat, const int32_t *pair_off,
                                         const uint32_t *refs, hgx_groups *g, uint64_t *rows_scratch, uint64_t *hash_scratch,
                                         void *stream) {
    if (out) *out = nullptr;
    return fail_clean(level_classes_grouped_impl(out, ix, compat, pair_off, refs, g, rows_scratch, hash_scratch, stream), out,
                      (hipStream_t)stream, hgx_classes_destroy);
}

int hgx_level_classes_grouped_seg(hgx_classes **out, const hgx_index *ix, const uint64_t *compat, const int32_t *pair_off,
                                  const uint32_t *refs, hgx_groups *g, uint64_t *rows_scratch, uint64_t *hash_scratch,
                                  const uint32_t *pair_seg, void *stream) {
    if (out) *out = nullptr;
    return fail_clean(level_classes_grouped_impl(out, ix, compat, pair_off, refs, g, rows_scratch, hash_scratch, stream, pair_seg), out,
                      (hipStream_t)stream, hgx_classes_destroy);
}

extern "C" int hgx_level_classes(hgx_classes **out, const hgx_index *ix, const uint64_t *compat, const int32_t *pair_off,
                                 const uint32_t *refs, int32_t n_pairs, int32_t level, uint64_t *rows_scratch,
                                 uint64_t *hash_scratch, void *stream) {
    ARGCHK(out && ix && n_pairs >= 0 && (level == 0 || level == 1));
    *out = nullptr;
    hgx_groups *g = nullptr;
    int rc = hgx_group_pairs(&g, pair_off, refs, n_pairs, level, stream);
    if (rc == HGX_OK) rc = hgx_level_classes_grouped(out, ix, compat, pair_off, refs, g, rows_scratch, hash_scratch, stream);
    if (rc == HGX_OK && *out) {
        // the group tables travel with the class set (its queued kernels read them)
        (*out)->d_keep[10] = g->d_first; (*out)->d_keep[11] = g->d_count;
        g->d_first = nullptr; g->d_count = nullptr;
    }
    hgx_groups_destroy(g);
    return rc;
}

extern "C" int hgx_classes_destroy(hgx_classes *c);
int hgx_dedup_classes_seg(hgx_classes **out, const uint64_t *rows, const uint64_t *row_hash, int64_t n_rows, int32_t a_pad,
                          const uint32_t *row_seg, void *stream) {
    if (out) *out = nullptr;
    return fail_clean(dedup_classes_impl(out, rows, row_hash, nullptr, n_rows, a_pad, nullptr, stream, row_seg), out, (hipStream_t)stream,
                      hgx_classes_destroy);
}
extern "C" int hgx_dedup_classes(hgx_classes **out, const uint64_t *rows, const uint64_t *row_hash, const int64_t *row_weight,
                                 int64_t n_rows, int32_t a_pad, const uint64_t *and_mask, void *stream) {
    if (out) *out = nullptr;
    return fail_clean(dedup_classes_impl(out, rows, row_hash, row_weight, n_rows, a_pad, and_mask, stream), out, (hipStream_t)stream,
                      hgx_classes_destroy);
}

#ifdef HGX_LAB
#include "lab/hgx_fused_dedup_lab.inc"        // hgx_pair_classes_dedup: round 2's fused gene-level form (measured slower), lab build only
#else
extern "C" int hgx_pair_classes_dedup(hgx_classes **out, const hgx_index *, const uint64_t *, const int32_t *, const uint32_t *, int32_t, int32_t,
                                      uint64_t *, void *) {
    if (out) *out = nullptr;
    hgx_set_error("hgx_pair_classes_dedup is lab code (measured slower than hgx_pair_classes + hgx_dedup_classes): build libhgx_lab.so");
    return HGX_EINVAL;
}
#endif

extern "C" int hgx_classes_destroy(hgx_classes *c) {
    if (!c) return HGX_OK;
    // Kernels of the stream that filled this set may still be queued (error paths, a set dropped before its results were
    // fetched): its blocks must not reach another stream through the pool before they are done.  A no-op on the happy paths,
    // where the caller has synchronised to read the results.
    if (c->ready) (void)hipEventSynchronize(c->ready);
    else if (c->made_on) (void)hipStreamSynchronize(c->made_on);
    hgx_pool_free(c->d_bits); hgx_pool_free(c->d_count); hgx_pool_free(c->d_first_row); hgx_pool_free(c->d_bitsT);
    hgx_pool_free(c->d_prow); hgx_pool_free(c->d_pcol);
    hgx_pool_free(c->d_act); hgx_pool_free(c->d_bitsC); hgx_pool_free(c->d_bitsTC);
    hgx_pool_free(c->d_wrow); hgx_pool_free(c->d_wcol);
    hgx_pool_free(c->d_setup0); hgx_pool_free(c->d_setup1);
    for (void *p : c->d_keep) hgx_pool_free(p);
    if (c->ready) (void)hipEventDestroy(c->ready);
    delete[] c->h_act;
    delete[] c->h_rank;
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
    HIPCHK(hipStreamSynchronize(c->made_on));        // the gather kernels of a large dedup may still be queued there
    if (bits) HIPCHK(hipMemcpy(bits, c->d_bits, (size_t)c->n_classes * c->w64 * 8, hipMemcpyDeviceToHost));
    if (count) HIPCHK(hipMemcpy(count, c->d_count, (size_t)c->n_classes * 8, hipMemcpyDeviceToHost));
    if (first_row) HIPCHK(hipMemcpy(first_row, c->d_first_row, (size_t)c->n_classes * 8, hipMemcpyDeviceToHost));
    return HGX_OK;
}
extern "C" int hgx_classes_from_host(hgx_classes **out, const uint64_t *bits, const int64_t *count, int32_t n_classes, int32_t a_pad) {
    ARGCHK(out && n_classes >= 0 && a_pad > 0 && a_pad % 512 == 0);
    hgx_classes *cl = new hgx_classes();
    cl->a_pad = a_pad; cl->w64 = a_pad / 64; cl->n_classes = n_classes; cl->c64 = 0;
    cl->d_bits = nullptr; cl->d_count = nullptr; cl->d_first_row = nullptr; cl->d_bitsT = nullptr;
    *out = cl;
    if (n_classes == 0) return HGX_OK;
    ARGCHK(bits && count);
    cl->d_bits = (decltype(cl->d_bits))hgx_pool_alloc((size_t)n_classes * cl->w64 * 8);
    if (!cl->d_bits) { hgx_set_error("device allocation failed"); return HGX_ENOMEM; }
    cl->d_count = (decltype(cl->d_count))hgx_pool_alloc((size_t)n_classes * 8);
    if (!cl->d_count) { hgx_set_error("device allocation failed"); return HGX_ENOMEM; }
    cl->d_first_row = (decltype(cl->d_first_row))hgx_pool_alloc((size_t)n_classes * 8);
    if (!cl->d_first_row) { hgx_set_error("device allocation failed"); return HGX_ENOMEM; }
    HIPCHK(hipMemcpy(cl->d_bits, bits, (size_t)n_classes * cl->w64 * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(cl->d_count, count, (size_t)n_classes * 8, hipMemcpyHostToDevice));
    std::vector<int64_t> fr(n_classes);
    for (int i = 0; i < n_classes; ++i) fr[i] = i;
    HIPCHK(hipMemcpy(cl->d_first_row, fr.data(), (size_t)n_classes * 8, hipMemcpyHostToDevice));
    return HGX_OK;
}

// ------------------------------------------------------------------------------------------------
// bit-matrix transpose [C][w64] -> [a_pad][c64]: one wavefront per 64x64 tile; lane r loads row r's
// word, wave_transpose64 turns the tile around.
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
    bitsT[(size_t)(aw * 64 + lane) * c64 + cw] = wave_transpose64(x);
}

int hgx_ensure_transposed(hgx_classes *c, hipStream_t st) {
    if (c->d_bitsT || c->n_classes == 0) return HGX_OK;
    c->c64 = ((c->n_classes + 63) / 64 + 7) / 8 * 8;   // row stride of the transposed matrix: multiple of 8 words, zero padded
    c->d_bitsT = (decltype(c->d_bitsT))hgx_pool_alloc((size_t)c->a_pad * c->c64 * 8);
    if (!c->d_bitsT) { hgx_set_error("device allocation failed"); return HGX_ENOMEM; }
    const long tiles = (long)c->c64 * c->w64;
    hipLaunchKernelGGL(k_transpose, dim3(nblk(tiles, 4)), dim3(256), 0, st, c->d_bits, c->n_classes, c->w64, c->c64, c->d_bitsT);
    HIPCHK(hipGetLastError());
    return HGX_OK;
}

