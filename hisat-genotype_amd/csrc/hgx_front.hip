// hgx_front.hip -- the DEVICE front end (gfx950): rows 8a-2 .. 8a-5 of the scope table as kernels.
//
// The host stages (hgx_sam.cpp: tokenise, record filters, grouping of the records by decode key) hand over the DISTINCT decode
// keys with their text (cigar | seq | Zs | MD: ~200 bytes per key, 0.28 keys per read at 1 M reads) and one word per record that
// passed the filters.  Everything after that runs here, in HBM, and ends as an hgx_dbatch that the scoring kernels read in place:
//   k_fe_pileup     get_mpileup (typing_common.py:1059-1134): one wavefront per key, lanes over the bases of an M op, counters
//                   privatised in LDS per workgroup (84 KB for a 3.5 kb locus), flushed with one global atomic per non-zero cell
//   k_fe_nt_set     the 20 % / >= 7 rule per position (typing_common.py:1124-1134)
//   k_fe_decode     ONE LANE PER KEY: CIGAR x MD x Zs walk, error correction against the pileup, novel variants, cmp_list2,
//                   identify_ambigious_diffs, haplotypes, exon clipping and piece masks (hgx_front_core.hpp: typing_core.py:
//                   899-1164, 119-243, 1351-1406, 718-792, 641-670; typing_common.py:1663-1955); haplotype records, candidate
//                   pieces and mask words go to pools through atomic cursors
//   piece table     candidates radix-sorted by a 64-bit content key (hipcub), run heads = distinct pieces (every candidate is
//                   compared word for word with its run's predecessor), heads ordered by (first word, width, PieceTable::hash,
//                   bytes) -- the host's canonical order, so the batch is the host's batch byte for byte
//   k_fe_pair_*     the pair protocol (typing_core.py:1238-1347, 1545-1587): one lane per run of records with one read id, set
//                   union of the mates' haplotypes, refs counted, scanned, written
// Whatever the kernels cannot take (a record the reference would raise on, a scratch limit, a full pool) sets a decline code:
// the call then runs the host stages, which are pinned to the reference and reproduce its failures.  The host front end is the
// checker of this file (tests/test_gpu_front.py: device batch == host batch on every fixture and on fuzz cases).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <atomic>
#include <chrono>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include <sys/stat.h>

#include "hgx_common.hpp"
#include "hgx_internal.hpp"

int hgx_bgzf_inflate_dev(const unsigned char *d_in, const hgx_bgzf_block *blocks, size_t n_blocks, unsigned char *d_out, hipStream_t st, int *bad,
                         void *staging);
size_t hgx_bgzf_inflate_staging_bytes(size_t n_blocks);

namespace {

// ---- pinned staging blocks (the key table of a 1 M-read sample is ~65 MB: pinning that much per call costs more than the copy) ----
struct PinnedPool {
    std::mutex mu;
    std::multimap<size_t, void *> free_blocks;
    std::map<void *, size_t> size_of;
    size_t idle_bytes = 0;                 // bytes of the blocks waiting in free_blocks
    // idle blocks beyond this are unregistered and freed instead of kept (a long-lived process that sees inputs of many sizes
    // must not accumulate pinned memory without bound); HGX_PINNED_IDLE_MB overrides the 4 GB default
    size_t idle_cap = [] { const char *e = getenv("HGX_PINNED_IDLE_MB"); return e ? (size_t)strtoull(e, nullptr, 10) << 20 : (size_t)4 << 30; }();
};
PinnedPool &pinned() { static PinnedPool *p = new PinnedPool(); return *p; }
void *pinned_alloc(size_t n) {
    PinnedPool &P = pinned();
    // (a block is at least 2 MB -- the registration's grain below -- and a request takes a block of up to 4x its size + 1 MB: a
    // request below 256 KB would never find its own block again and register a new one per call, 0.5 ms each)
    size_t need = std::max<size_t>(n, 512u << 10);
    {
        std::lock_guard<std::mutex> g(P.mu);
        auto it = P.free_blocks.lower_bound(need);
        if (it != P.free_blocks.end() && it->first <= 4 * need + (1u << 20)) {
            void *p = it->second;
            P.idle_bytes -= it->first;
            P.free_blocks.erase(it);
            return p;
        }
    }
    // ordinary (first-touch, huge-page eligible) memory registered with the runtime: the host's workers fill it at malloc speed --
    // tools/pinned_probe.hip: 185 GB/s against 130 GB/s into hipHostMalloc memory, and pread() into the latter took twice as long --
    // and it travels at the same 57 GB/s.  Registering costs ~0.1 ms per MB: blocks are kept for the life of the process.
    need += need / 8;
    need = (need + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
    void *p = aligned_alloc(2u << 20, need);
    if (!p) return nullptr;
    if (hipHostRegister(p, need, hipHostRegisterDefault) != hipSuccess) { free(p); return nullptr; }
    std::lock_guard<std::mutex> g(P.mu);
    P.size_of[p] = need;
    return p;
}
void pinned_release(void *p) {
    if (!p) return;
    PinnedPool &P = pinned();
    std::lock_guard<std::mutex> g(P.mu);
    auto it = P.size_of.find(p);
    if (it == P.size_of.end()) return;
    if (P.idle_bytes + it->second > P.idle_cap) {          // enough idle staging already: give this one back to the system
        P.size_of.erase(it);
        (void)hipHostUnregister(p);
        free(p);
        return;
    }
    P.idle_bytes += it->second;
    P.free_blocks.emplace(it->second, p);
}

// ---- the locus tables on the device (once per locus and device) ---------------------------------------------------------------
struct DevLocus {
    hgx_front_tables T;
    struct PerDev { int dev; void *block; FeLocus view; };
    std::vector<PerDev> per_dev;
    std::mutex mu;
    ~DevLocus() { for (auto &d : per_dev) hgx_pool_free(d.block); }
};
void dev_locus_free(void *p) { delete (DevLocus *)p; }
std::mutex g_locus_mu;

int dev_locus(hgx_locus &L, hipStream_t st, const FeLocus **view, bool *usable) {
    DevLocus *D;
    {
        std::lock_guard<std::mutex> g(g_locus_mu);
        if (!L.fe_dev) {
            D = new DevLocus();
            const int rc = hgx_front_tables_build(L, D->T);
            if (rc) { delete D; return rc; }
            L.fe_dev = D;
            L.fe_dev_free = dev_locus_free;
        }
        D = (DevLocus *)L.fe_dev;
    }
    *usable = D->T.usable;
    if (!D->T.usable) return HGX_OK;
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> g(D->mu);
    for (auto &d : D->per_dev) if (d.dev == dev) { *view = &d.view; return HGX_OK; }
    // one allocation, every table 256-byte aligned
    const FeLocus H = hgx_front_view(L, D->T);
    struct Part { const void *src; size_t bytes; size_t off; };
    std::vector<Part> parts;
    size_t total = 0;
    auto add = [&](const void *src, size_t bytes) { parts.push_back({src, bytes, total}); total += (bytes + 255) & ~(size_t)255; return parts.size() - 1; };
    const size_t V = (size_t)L.V;
    const size_t i_pos = add(H.pos, V * 4), i_right = add(H.right, V * 4), i_len = add(H.len, V * 4), i_maxr = add(H.maxright, V * 4);
    const size_t i_type = add(H.type, V), i_linked = add(H.linked, V), i_base = add(H.base, V);
    const size_t i_lbits = add(H.linked_bits, L.linked_bits.size() * 4), i_bb = add(H.backbone, L.backbone.size());
    const size_t i_ex = add(H.exons, D->T.exons.size() * 4), i_hv = add(H.hv_index, L.hv_index.size() * 4);
    const size_t i_noff = add(H.name_off, D->T.name_off.size() * 4), i_npool = add(H.name_pool, D->T.name_pool.size());
    size_t i_anchor[2], i_koff[2], i_loff[2];         // (the keys' spellings stay on the host: the kernels compare variant ids, fe_key_contains)
    for (int d = 0; d < 2; ++d) {
        i_anchor[d] = add(H.alt_anchor[d], D->T.alt_anchor[d].size() * 4);
        i_koff[d] = add(H.alt_key_off[d], D->T.alt_key_off[d].size() * 4);
        i_loff[d] = add(H.alt_list_off[d], D->T.alt_list_off[d].size() * 4);
    }
    const size_t i_htoff = add(H.alt_ht_off, D->T.alt_ht_off.size() * 4), i_ints = add(H.alt_ints, D->T.alt_ints.size() * 4);
    char *block = (char *)hgx_pool_alloc(std::max<size_t>(total, 256));
    if (!block) { hgx_set_error("device allocation of the front end's locus tables failed"); return HGX_ENOMEM; }
    std::vector<char> host(total, 0);
    for (auto &p : parts) if (p.bytes) memcpy(host.data() + p.off, p.src, p.bytes);
    if (total) HIPCHK(hipMemcpyAsync(block, host.data(), total, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    FeLocus F = H;
    auto at = [&](size_t i) { return (const void *)(block + parts[i].off); };
    F.pos = (const int32_t *)at(i_pos); F.right = (const int32_t *)at(i_right); F.len = (const int32_t *)at(i_len);
    F.maxright = (const int32_t *)at(i_maxr); F.type = (const uint8_t *)at(i_type); F.linked = (const uint8_t *)at(i_linked);
    F.base = (const char *)at(i_base); F.linked_bits = (const uint32_t *)at(i_lbits); F.backbone = (const char *)at(i_bb);
    F.exons = (const int32_t *)at(i_ex); F.hv_index = (const int32_t *)at(i_hv); F.name_off = (const int32_t *)at(i_noff);
    F.name_pool = (const char *)at(i_npool);
    for (int d = 0; d < 2; ++d) {
        F.alt_anchor[d] = (const int32_t *)at(i_anchor[d]); F.alt_key_off[d] = (const int32_t *)at(i_koff[d]);
        F.alt_str_off[d] = nullptr; F.alt_list_off[d] = (const int32_t *)at(i_loff[d]);
    }
    F.alt_ht_off = (const int32_t *)at(i_htoff); F.alt_ints = (const int32_t *)at(i_ints); F.alt_chars = nullptr;
    D->per_dev.push_back({dev, block, F});
    *view = &D->per_dev.back().view;
    return HGX_OK;
}

// ---- control block of one call ----------------------------------------------------------------------------------------------------
struct FeCtl {
    uint32_t ht_cursor, cand_cursor, mask_cursor;
    int32_t decline;                 // first decline code seen (an FE_E_* value, negative), 0 = none
    uint32_t n_heads, n_masks;
    uint32_t n_keys, n_slots, n_rec, trace_cursor;            // record stage: distinct keys, decoded keys, records that passed the filters
    unsigned long long n_reads, n_gene_refs, pair_total;      // pair_total = pairs << 40 | refs
    uint32_t last_head;              // CODIS D18S51: index + 1 of the last run of records that yields a pair (choose_pairs applies to it)
};

__device__ __forceinline__ void fe_decline(FeCtl *ctl, int code) { atomicCAS(&ctl->decline, 0, code); }

// ---- kernels ----------------------------------------------------------------------------------------------------------------------
// grid (blocks per task, tiles of the backbone, tasks).  The keys are in stream order of their first records, so a task's keys
// are one run of the table: its ends are found by bisection on FeKey::task (one task: the whole table).
__global__ void __launch_bounds__(1024) k_fe_pileup(const FeKey *__restrict__ keys, uint32_t n_keys, const char *__restrict__ text, int n_ref,
                                                    int tile, uint32_t *__restrict__ counts, FeCtl *ctl) {
    extern __shared__ uint32_t hist[];
    const int t0 = blockIdx.y * tile, t1 = min(n_ref, t0 + tile);
    const int cells = (t1 - t0) * 6;
    for (int c = threadIdx.x; c < cells; c += blockDim.x) hist[c] = 0;
    uint32_t k_lo = 0, k_hi = n_keys;
    const uint32_t task = blockIdx.z;
    if (gridDim.z > 1) {
        uint32_t lo = 0, hi = n_keys;                      // first key with task >= `task`
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (keys[mid].task < task) lo = mid + 1; else hi = mid; }
        k_lo = lo;
        hi = n_keys;                                       // first key with task > `task`
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (keys[mid].task <= task) lo = mid + 1; else hi = mid; }
        k_hi = lo;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    for (uint32_t k = k_lo + blockIdx.x * wpb + wave; k < k_hi; k += gridDim.x * wpb) {
        const FeKey K = keys[k];
        if (K.n_pile == 0 || K.pos >= t1) continue;
        const int lo = t0 * 6;
        const int rc = fe_pileup_key(K, text, n_ref, lane, 64, [&](uint32_t cell, uint32_t w) {
            const int c = (int)cell - lo;
            if (c >= 0 && c < cells) atomicAdd(&hist[c], w);
        });
        if (rc < 0 && lane == 0) fe_decline(ctl, rc);
    }
    __syncthreads();
    uint32_t *mine = counts + (size_t)task * n_ref * 6;
    for (int c = threadIdx.x; c < cells; c += blockDim.x) {
        const uint32_t v = hist[c];
        if (v) atomicAdd(&mine[(size_t)t0 * 6 + c], v);
    }
}

__global__ void k_fe_nt_set(const uint32_t *__restrict__ counts, int n_ref, uint8_t *__restrict__ nt_set) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_ref) nt_set[i] = fe_nt_set(counts + (size_t)i * 6);
}

#define FE_SEQ_STAGE_DWORDS 43
#ifndef FE_DECODE_BLOCK
#define FE_DECODE_BLOCK 64
#endif
__global__ void __launch_bounds__(FE_DECODE_BLOCK) k_fe_decode(FeLocus L, FeParse o, FePile P, const FeKey *__restrict__ keys, uint32_t n_keys,
                                                   const char *__restrict__ text, FePools pools, uint8_t *__restrict__ state,
                                                   uint32_t *__restrict__ key_ht_off, uint32_t *__restrict__ key_n_ht,
                                                   uint16_t *__restrict__ slot_task, const uint32_t *__restrict__ order, FeCtl *ctl) {
    const uint32_t k0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (k0 >= n_keys) return;
    const uint32_t k = order ? order[k0] : k0;             // (keys in backbone order: the lanes of a wavefront meet the same variants)
    const FeKey K = keys[k];
    if (K.slot == FE_NO_SLOT) return;
    // The read's bases are looked at one by one, several times over (the MD walk, error correction against the pileup, insertions):
    // from global memory every such look is a load instruction whose 64 lanes touch 64 different cache lines.  They are staged in
    // LDS once, a few dword loads per lane -- 43 dwords per lane (an odd stride: no bank conflicts between the lanes; 172 bytes hold
    // 344 packed or 172 text bases; longer reads stay where they are).
    __shared__ uint32_t s_seq[FE_DECODE_BLOCK * FE_SEQ_STAGE_DWORDS];
    const unsigned char *seq_copy = nullptr;
    {
        const uint32_t n_bytes = (K.flags & FE_K_PACKED_SEQ) ? ((uint32_t)K.seq_len + 1) / 2 : (uint32_t)K.seq_len;
        if (n_bytes <= 4 * FE_SEQ_STAGE_DWORDS) {
            uint32_t *mine = s_seq + threadIdx.x * FE_SEQ_STAGE_DWORDS;
            const unsigned char *src = (const unsigned char *)text + K.seq_off;
            for (uint32_t w = 0; 4 * w < n_bytes; ++w) {                    // (a dword load may reach 3 bytes beyond the bases: inside the text's padding at worst)
                uint32_t v;
                __builtin_memcpy(&v, src + 4 * w, 4);
                mine[w] = v;
            }
            seq_copy = (const unsigned char *)mine;
        }
    }
    uint8_t st = 2;
    uint32_t off = 0, n = 0;
    FePile Pk = P;                                         // (a many-task batch: the pileup of the key's own sample)
    Pk.nt_set += (size_t)K.task * L.n_ref;
    Pk.counts += (size_t)K.task * L.n_ref * 6;
    if (slot_task) slot_task[K.slot] = (uint16_t)K.task;
    const int rc = fe_key(L, o, Pk, K, text, pools, st, off, n, seq_copy);
    if (rc < 0) { fe_decline(ctl, rc); st = 2; n = 0; }
    state[K.slot] = st;
    key_ht_off[K.slot] = off;
    key_n_ht[K.slot] = n;
}

__global__ void k_fe_key_pos(const FeKey *__restrict__ keys, uint32_t n, uint32_t *__restrict__ pos, uint32_t *__restrict__ idx) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { pos[i] = ((uint32_t)keys[i].task << 20) ^ (uint32_t)(keys[i].pos + 4096); idx[i] = i; }      // (task, position)
}

// backbone order of ONE task's keys by counting (round 6, late: 4 launches instead of a library sort's 17): any order that keeps the keys
// of a position together does -- the order inside a position is whatever the atomics give, the decode's result does not depend on it
__device__ __forceinline__ uint32_t fe_pos_bin(const FeKey &K, uint32_t bins) {
    const int b = K.pos + 4096;
    return b < 0 ? 0u : ((uint32_t)b >= bins ? bins - 1 : (uint32_t)b);
}
__global__ void k_fe_pos_hist(const FeKey *__restrict__ keys, uint32_t n, uint32_t bins, uint32_t *__restrict__ hist) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&hist[fe_pos_bin(keys[i], bins)], 1u);
}
__global__ void k_fe_pos_scatter(const FeKey *__restrict__ keys, uint32_t n, uint32_t bins, uint32_t *__restrict__ offs, uint32_t *__restrict__ order) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) order[atomicAdd(&offs[fe_pos_bin(keys[i], bins)], 1u)] = i;
}

// Distinct candidate pieces (round 6, late: a hash table instead of a 64-bit sort of the candidates -- 19 library launches per call):
// every candidate claims the slot of its 64-bit content key (linear probing, the key itself is the tag: two keys never share a slot);
// the slot's representative is the candidate with the smallest index; every other candidate of the slot must equal it word for word
// (two different pieces with one key: the call declines); the representatives, numbered in candidate order, are the heads.  The
// numbering carries no meaning: the piece table is ordered by content below.
__global__ void __launch_bounds__(256) k_fe_cand_insert(const uint64_t *__restrict__ ckey, uint32_t n, unsigned long long *__restrict__ tkeys,
                                                        uint32_t *__restrict__ rep, uint32_t mask, uint32_t *__restrict__ slot_of) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long k = ckey[i];
    if (k == ~0ull) k = ~0ull - 1;                                   // (all ones = an empty slot; the word-for-word check covers the alias)
    uint32_t slot = (uint32_t)(k ^ (k >> 32)) & mask;
    // (most candidates repeat a piece that is in the table already -- a popular piece thousands of times: a plain look first, the
    // atomics only where the slot is still empty or the representative still larger: 162 -> 117 us at 1 M reads -- ~1 M random
    // looks into a 16 MB table, about what the library's sort of the same keys took in 19 launches)
    for (uint32_t probe = 0; probe <= mask; ++probe) {
        unsigned long long cur = __hip_atomic_load(&tkeys[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == ~0ull) cur = atomicCAS(&tkeys[slot], ~0ull, k);
        if (cur == ~0ull || cur == k) break;
        slot = (slot + 1) & mask;
    }
    if (__hip_atomic_load(&rep[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > i) atomicMin(&rep[slot], i);
    slot_of[i] = slot;
}
__global__ void __launch_bounds__(256) k_fe_cand_flags(const uint32_t *__restrict__ slot_of, const uint32_t *__restrict__ rep, uint32_t n,
                                                       const uint16_t *__restrict__ lo, const uint16_t *__restrict__ nw, const uint32_t *__restrict__ mask_off,
                                                       const uint32_t *__restrict__ masks, uint32_t *__restrict__ flag, FeCtl *ctl) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = rep[slot_of[i]];
    if (r != i) {
        bool same = lo[i] == lo[r] && nw[i] == nw[r];
        if (same) {
            const uint32_t *ma = masks + mask_off[i], *mb = masks + mask_off[r];
            for (int w = 0; w < 2 * (int)nw[i] && same; ++w) same = ma[w] == mb[w];
        }
        if (!same) fe_decline(ctl, -HGX_FE_DECLINE_COLLISION);
    }
    flag[i] = r == i ? 1u : 0u;
}
__global__ void k_fe_cand_assign(const uint32_t *__restrict__ slot_of, const uint32_t *__restrict__ rep, const uint32_t *__restrict__ flag,
                                 const uint32_t *__restrict__ rank_ex, uint32_t n, uint32_t *__restrict__ head_of, uint32_t *__restrict__ head_cand, FeCtl *ctl) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    head_of[i] = rank_ex[rep[slot_of[i]]];
    if (flag[i]) head_cand[rank_ex[i]] = i;
    if (i == n - 1) ctl->n_heads = rank_ex[i] + flag[i];
}
__global__ void k_fe_head_keys(const uint32_t *__restrict__ head_cand, uint32_t n, const uint16_t *__restrict__ lo, const uint16_t *__restrict__ nw,
                               const uint32_t *__restrict__ mask_off, const uint32_t *__restrict__ masks, uint64_t *__restrict__ whash,
                               uint32_t *__restrict__ idx, int shift) {
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= n) return;
    const uint32_t c = head_cand[h];
    // ONE sort key for the canonical order (first word, width, PieceTable::hash, bytes): first word | width | the hash's upper half;
    // heads that still tie -- the hash's lower half, then the bytes -- are put in order by k_fe_tie_fix (round 6: two sorts before)
    // (`shift` = 32; the test switch front=tie_test keeps only the hash's top 8 bits, so that the tie path below has work to do)
    whash[h] = ((uint64_t)(((uint32_t)lo[c] << 16) | nw[c]) << 32) | ((fe_piece_hash(lo[c], nw[c], masks + mask_off[c]) >> shift) << (shift - 32));
    idx[h] = h;
}
// heads in (first word, width, upper half of PieceTable::hash) order: runs that still tie are ordered by the whole hash, then by their bytes
__global__ void k_fe_tie_fix(uint32_t *__restrict__ ord, uint32_t n, const uint32_t *__restrict__ head_cand, const uint16_t *__restrict__ lo,
                             const uint16_t *__restrict__ nw, const uint32_t *__restrict__ mask_off, const uint32_t *__restrict__ masks, int shift) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    auto same = [&](uint32_t x, uint32_t y) {
        const uint32_t a = head_cand[ord[x]], b = head_cand[ord[y]];
        return lo[a] == lo[b] && nw[a] == nw[b] &&
               (fe_piece_hash(lo[a], nw[a], masks + mask_off[a]) >> shift) == (fe_piece_hash(lo[b], nw[b], masks + mask_off[b]) >> shift);
    };
    if (k + 1 >= n || !same(k, k + 1) || (k > 0 && same(k - 1, k))) return;
    uint32_t e = k + 1;
    while (e + 1 < n && same(e, e + 1)) ++e;
    auto less = [&](uint32_t ha, uint32_t hb) {
        const uint32_t a = head_cand[ha], b = head_cand[hb];
        const uint64_t wa = fe_piece_hash(lo[a], nw[a], masks + mask_off[a]), wb = fe_piece_hash(lo[b], nw[b], masks + mask_off[b]);
        if (wa != wb) return wa < wb;
        const unsigned char *pa = (const unsigned char *)(masks + mask_off[a]), *pb = (const unsigned char *)(masks + mask_off[b]);
        for (int i = 0; i < 8 * (int)nw[a]; ++i) if (pa[i] != pb[i]) return pa[i] < pb[i];
        return false;
    };
    for (uint32_t i = k + 1; i <= e; ++i) {
        const uint32_t v = ord[i];
        uint32_t j = i;
        while (j > k && less(v, ord[j - 1])) { ord[j] = ord[j - 1]; --j; }
        ord[j] = v;
    }
}
__global__ void k_fe_piece_sizes(const uint32_t *__restrict__ ord, const uint32_t *__restrict__ head_cand, uint32_t n, const uint16_t *__restrict__ nw,
                                 uint32_t *__restrict__ nw2) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) nw2[k] = 2u * nw[head_cand[ord[k]]];
}
__global__ void k_fe_write_pieces(const uint32_t *__restrict__ ord, const uint32_t *__restrict__ head_cand, uint32_t n, const uint16_t *__restrict__ lo,
                                  const uint16_t *__restrict__ nw, const uint32_t *__restrict__ mask_off, const uint32_t *__restrict__ masks,
                                  const uint32_t *__restrict__ moff, hgx_piece *__restrict__ pieces, uint32_t *__restrict__ out_masks,
                                  uint32_t *__restrict__ new_id, FeCtl *ctl) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t h = ord[k], c = head_cand[h];
    hgx_piece pc;
    pc.mask_off = moff[k];
    pc.lo_word = lo[c];
    pc.n_words = nw[c];
    pieces[k] = pc;
    const uint32_t *src = masks + mask_off[c];
    for (int i = 0; i < 2 * (int)nw[c]; ++i) out_masks[moff[k] + i] = src[i];
    new_id[h] = k;
    if (k == n - 1) ctl->n_masks = moff[k] + 2u * nw[c];
}
__global__ void k_fe_cand_piece(const uint32_t *__restrict__ head_of, const uint32_t *__restrict__ new_id, uint32_t n, uint32_t *__restrict__ cand_piece) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n) cand_piece[c] = new_id[head_of[c]];
}

// a many-task batch: the distinct pieces each task's decoded keys produced (= the piece count of the task's own batch), as a
// bit per (task, piece), then a population count per task
__global__ void __launch_bounds__(256) k_fe_task_piece_bits(const uint8_t *__restrict__ state, const uint32_t *__restrict__ key_ht_off,
                                                            const uint32_t *__restrict__ key_n_ht, const int32_t *__restrict__ ht_pool,
                                                            const uint32_t *__restrict__ cand_piece, const uint16_t *__restrict__ slot_task, uint32_t n_slots,
                                                            uint32_t words_per_task, uint32_t *__restrict__ bits) {
    const uint32_t sl = blockIdx.x * blockDim.x + threadIdx.x;
    if (sl >= n_slots || state[sl] != 1) return;
    uint32_t *mine = bits + (size_t)slot_task[sl] * words_per_task;
    uint32_t at = key_ht_off[sl];
    for (uint32_t x = 0, n = key_n_ht[sl]; x < n; ++x) {
        const int32_t *rec = ht_pool + at;
        for (int e = 0; e <= rec[3]; ++e) {
            const uint32_t pc = cand_piece[(uint32_t)rec[4] + e];
            atomicOr(&mine[pc >> 5], 1u << (pc & 31));
        }
        at += FE_HT_HDR + (uint32_t)rec[2];
    }
}
__global__ void __launch_bounds__(256) k_fe_task_piece_count(const uint32_t *__restrict__ bits, uint32_t words_per_task, uint32_t *__restrict__ task_pieces) {
    __shared__ uint32_t part[4];
    const uint32_t *mine = bits + (size_t)blockIdx.x * words_per_task;
    uint32_t c = 0;
    for (uint32_t w = threadIdx.x; w < words_per_task; w += blockDim.x) c += (uint32_t)__popc(mine[w]);
    c = (uint32_t)wave_sum_u64(c);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) task_pieces[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// pair protocol, pass 1: per run of records with one read id -> (1 << 40 | refs) if it yields a pair
__global__ void __launch_bounds__(256) k_fe_pair_count(const uint32_t *__restrict__ rec_info, uint32_t n_rec, const uint8_t *__restrict__ state,
                                                       const uint32_t *__restrict__ key_ht_off, const uint32_t *__restrict__ key_n_ht,
                                                       const int32_t *__restrict__ ht_pool, unsigned long long *__restrict__ cnt, FeCtl *ctl,
                                                       const uint16_t *__restrict__ slot_task, uint32_t *__restrict__ task_reads,
                                                       uint32_t *__restrict__ task_pairs, unsigned long long *__restrict__ task_refs, int mark_last) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long c = 0, reads = 0, gene = 0;
    uint32_t my_task = 0xffffffffu;
    if (i < n_rec && FE_REC_HEAD(rec_info[i])) {
        uint32_t uni[FE_MAX_PAIR_HT];
        int n_uni = 0;
        const int ns = fe_pair_union(rec_info, i, n_rec, state, key_ht_off, key_n_ht, ht_pool, uni, n_uni);
        if (ns < 0) fe_decline(ctl, ns);
        else if (ns > 0) {
            unsigned long long n_exon = 0;
            for (int x = 0; x < n_uni; ++x) n_exon += (unsigned long long)ht_pool[uni[x] + 3];
            if (n_exon > 65535 || n_uni > 65535) fe_decline(ctl, FE_E_PAIR);
            c = (1ull << 40) | (n_exon + (unsigned long long)n_uni);
            reads = (unsigned long long)ns;
            gene = (unsigned long long)n_uni;
            if (task_reads) my_task = slot_task[FE_REC_SLOT(rec_info[i])];
            if (mark_last) atomicMax(&ctl->last_head, i + 1);          // (CODIS D18S51 only: a few thousand pairs)
        }
    }
    if (task_reads) {
        // a many-task batch: the totals of the pair's sample.  The records come task after task, so a wavefront nearly always
        // holds one task: one set of atomics per wavefront then (per lane, the 64 samples' counters are a hot spot)
        const unsigned long long have = __ballot(my_task != 0xffffffffu);
        if (have) {
            const uint32_t t0 = (uint32_t)__shfl((int)my_task, __ffsll((long long)have) - 1);
            if (__all(my_task == 0xffffffffu || my_task == t0)) {
                const unsigned long long r = wave_sum_u64(reads), p = wave_sum_u64(c >> 40), f = wave_sum_u64(c & ((1ull << 40) - 1));
                if ((threadIdx.x & 63) == 0) { atomicAdd(&task_reads[t0], (uint32_t)r); atomicAdd(&task_pairs[t0], (uint32_t)p); atomicAdd(&task_refs[t0], f); }
            } else if (my_task != 0xffffffffu) {
                atomicAdd(&task_reads[my_task], (uint32_t)reads);
                atomicAdd(&task_pairs[my_task], 1u);
                atomicAdd(&task_refs[my_task], c & ((1ull << 40) - 1));
            }
        }
    }
    if (i < n_rec) cnt[i] = c;
    // the call's totals: ONE pair of atomics per workgroup.  (One per wavefront was 31 000 atomics on two addresses at 1 M records:
    // they queue in one L2 channel, a wavefront ends only when its own has been served, and the kernel took 0.39 ms for 20
    // instructions per lane.)
    __shared__ unsigned long long s_reads[4], s_gene[4], s_pack[4];
    reads = wave_sum_u64(reads);
    gene = wave_sum_u64(gene);
    const unsigned long long pack = wave_sum_u64(c);                  // pairs << 40 | refs of the workgroup's runs (64-bit: the scan's offsets are 32-bit)
    if ((threadIdx.x & 63) == 0) { s_reads[threadIdx.x >> 6] = reads; s_gene[threadIdx.x >> 6] = gene; s_pack[threadIdx.x >> 6] = pack; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long r = s_reads[0] + s_reads[1] + s_reads[2] + s_reads[3], g = s_gene[0] + s_gene[1] + s_gene[2] + s_gene[3];
        const unsigned long long pk = s_pack[0] + s_pack[1] + s_pack[2] + s_pack[3];
        if (r | g | pk) { atomicAdd(&ctl->n_reads, r); atomicAdd(&ctl->n_gene_refs, g); atomicAdd(&ctl->pair_total, pk); }
    }
}
// choose_pairs (typing_core.py:680-716, 1547-1552): the stream's LAST pair of a CODIS D18S51 sample keeps the left x right haplotype
// pairs whose inner distance is closest to the sample's median one -- one thread re-counts that pair and corrects the totals
__global__ void k_fe_pair_choose(const uint32_t *__restrict__ rec_info, uint32_t n_rec, const uint8_t *__restrict__ state,
                                 const uint32_t *__restrict__ key_ht_off, const uint32_t *__restrict__ key_n_ht, const int32_t *__restrict__ ht_pool,
                                 unsigned long long *__restrict__ cnt, FeCtl *ctl, long long expected) {
    if (blockIdx.x || threadIdx.x || ctl->last_head == 0) return;
    const uint32_t i = ctl->last_head - 1;
    uint32_t uni[FE_MAX_PAIR_HT];
    int n_all = 0, n_uni = 0;
    if (fe_pair_union(rec_info, i, n_rec, state, key_ht_off, key_n_ht, ht_pool, uni, n_all) <= 0) return;
    if (fe_pair_union(rec_info, i, n_rec, state, key_ht_off, key_n_ht, ht_pool, uni, n_uni, true, expected) <= 0) return;
    unsigned long long n_exon = 0;
    for (int x = 0; x < n_uni; ++x) n_exon += (unsigned long long)ht_pool[uni[x] + 3];
    const unsigned long long before = cnt[i] & ((1ull << 40) - 1), now = n_exon + (unsigned long long)n_uni;
    cnt[i] = (1ull << 40) | now;
    ctl->pair_total -= before - now;                                 // (a subset of the pair's haplotypes: never more refs)
    ctl->n_gene_refs -= (unsigned long long)(n_all - n_uni);
}
__global__ void __launch_bounds__(256) k_fe_pair_emit(const uint32_t *__restrict__ rec_info, uint32_t n_rec, const uint8_t *__restrict__ state,
                                                      const uint32_t *__restrict__ key_ht_off, const uint32_t *__restrict__ key_n_ht,
                                                      const int32_t *__restrict__ ht_pool, const unsigned long long *__restrict__ cnt,
                                                      const uint32_t *__restrict__ off_pair, const uint32_t *__restrict__ off_ref, const uint32_t *__restrict__ cand_piece,
                                                      int32_t *__restrict__ pair_off, uint32_t *__restrict__ pair_ref, uint32_t n_pairs, uint32_t n_refs,
                                                      uint32_t choose_head, long long expected) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) pair_off[n_pairs] = (int32_t)n_refs;
    if (i >= n_rec || cnt[i] == 0) return;
    uint32_t uni[FE_MAX_PAIR_HT];
    int n_uni = 0;
    (void)fe_pair_union(rec_info, i, n_rec, state, key_ht_off, key_n_ht, ht_pool, uni, n_uni, i + 1 == choose_head, expected);
    const uint32_t p = off_pair[i];
    uint32_t r = off_ref[i];
    pair_off[p] = (int32_t)r;
    for (int x = 0; x < n_uni; ++x) {
        const int32_t *rec = ht_pool + uni[x];
        for (int e = 0; e < rec[3]; ++e) pair_ref[r++] = cand_piece[(uint32_t)rec[4] + e];
    }
    for (int x = 0; x < n_uni; ++x) {
        const int32_t *rec = ht_pool + uni[x];
        pair_ref[r++] = cand_piece[(uint32_t)rec[4] + rec[3]] | 0x80000000u;
    }
}

// inclusive prefix sum over the 64 lanes through the register file (row_shr 1 / 2 / 4 / 8, row_bcast:15, row_bcast:31)
__device__ __forceinline__ uint32_t wave_incl_scan_u32_front(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, true);
    return v;
}
// ---- prefix scans of the front end (round 5: hand-written; the nine hipcub::DeviceScan call sites -- two launches each -- are gone) ----
// Up to four 32-bit channels over the same n items in ONE single-pass launch (decoupled look-back, as k_scan_u32 of hgx_dedup.hip):
// exclusive sums, or an exclusive running maximum; a channel may be a field of the pair protocol's packed 64-bit counts.  Tile state =
// one 64-bit word per (channel, tile): status in the top two bits (1 = the tile's own aggregate, 2 = the inclusive prefix), value below;
// the look-back of the channels runs on the first lanes of a wavefront side by side.  totals[ch] = the aggregate over all items (the
// one-thread kernels that read "last offset + last count" are gone too).  `state` ([4][tiles] words + a ticket) is zeroed by the caller.
#define FSC_T 1024
#define FSC_TILE (4 * FSC_T)
enum { FSC_U32 = 0, FSC_PAIR_FLAG = 1, FSC_PAIR_REFS = 2 };
struct FeScanCh { const void *in; uint32_t *out; int is_max, kind; };
struct FeScanArgs { FeScanCh ch[4]; int n_ch; uint32_t *totals; };
__device__ __forceinline__ uint32_t fsc_load(const FeScanCh &c, long i) {
    if (c.kind == FSC_U32) return ((const uint32_t *)c.in)[i];
    const unsigned long long v = ((const unsigned long long *)c.in)[i];
    return c.kind == FSC_PAIR_FLAG ? (uint32_t)(v >> 40) : (uint32_t)(v & ((1ull << 40) - 1));
}
__global__ __launch_bounds__(FSC_T) void k_fe_scan(FeScanArgs a, long n, unsigned long long *__restrict__ state, uint32_t *__restrict__ ticket) {
    __shared__ uint32_t s_tile;
    __shared__ uint32_t s_wave[4][FSC_T / 64];
    __shared__ uint32_t s_prefix[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_tile = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t tile = s_tile;
    const long tiles = (n + FSC_TILE - 1) / FSC_TILE;
    const long i0 = (long)tile * FSC_TILE + 4 * tid;
    uint32_t v[4][4], mine[4], incl[4];
    for (int c = 0; c < a.n_ch; ++c) {
        const bool mx = a.ch[c].is_max != 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[c][k] = i0 + k < n ? fsc_load(a.ch[c], i0 + k) : 0u;
        mine[c] = mx ? max(max(v[c][0], v[c][1]), max(v[c][2], v[c][3])) : v[c][0] + v[c][1] + v[c][2] + v[c][3];
        uint32_t x = mine[c];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(x, d, 64);
            if (lane >= d) x = mx ? max(x, up) : x + up;
        }
        incl[c] = x;
        if (lane == 63) s_wave[c][wv] = x;
    }
    __syncthreads();
    if (wv == 0) {
        for (int c = 0; c < a.n_ch; ++c) {
            const bool mx = a.ch[c].is_max != 0;
            const uint32_t w = lane < FSC_T / 64 ? s_wave[c][lane] : 0u;
            uint32_t wi = w;
#pragma unroll
            for (int d = 1; d < FSC_T / 64; d <<= 1) {
                const uint32_t up = __shfl_up(wi, d, 64);
                if (lane >= d) wi = mx ? max(wi, up) : wi + up;
            }
            // exclusive aggregate of the wavefronts before mine; the tile's aggregate sits in lane 15
            const uint32_t prev = __shfl_up(wi, 1, 64);
            if (lane < FSC_T / 64) s_wave[c][lane] = lane ? prev : 0u;
            const uint32_t tile_total = __shfl(wi, FSC_T / 64 - 1, 64);
            if (lane == c) mine[0] = tile_total;                       // (lane c keeps channel c's aggregate for the look-back below)
        }
        if (lane < a.n_ch) {
            const int c = lane;
            const bool mx = a.ch[c].is_max != 0;
            const unsigned long long total = mine[0];
            unsigned long long *st = state + (size_t)c * tiles;
            unsigned long long before = 0;
            if (tile == 0) __hip_atomic_store(&st[0], (2ull << 62) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else {
                __hip_atomic_store(&st[tile], (1ull << 62) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (long j = (long)tile - 1;; --j) {
                    unsigned long long sv;
                    do sv = __hip_atomic_load(&st[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); while ((sv >> 62) == 0);
                    const unsigned long long val = sv & ((1ull << 62) - 1);
                    before = mx ? (val > before ? val : before) : before + val;
                    if ((sv >> 62) == 2) break;
                }
                const unsigned long long inc = mx ? (total > before ? total : before) : before + total;
                __hip_atomic_store(&st[tile], (2ull << 62) | inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            s_prefix[c] = (uint32_t)before;
            if ((long)(tile + 1) * FSC_TILE >= n && a.totals) a.totals[c] = (uint32_t)(mx ? (total > before ? total : before) : before + total);
        }
    }
    __syncthreads();
    for (int c = 0; c < a.n_ch; ++c) {
        const bool mx = a.ch[c].is_max != 0;
        // everything before this thread's four items: the tiles before, the wavefronts before, the lanes before
        const uint32_t lanes_before = __shfl_up(incl[c], 1, 64);
        uint32_t run = mx ? max(s_prefix[c], s_wave[c][wv]) : s_prefix[c] + s_wave[c][wv];
        if (lane) run = mx ? max(run, lanes_before) : run + lanes_before;
        uint32_t *out = a.ch[c].out;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i0 + k < n) out[i0 + k] = run;
            run = mx ? max(run, v[c][k]) : run + v[c][k];
        }
    }
}
static size_t fe_scan_scratch_bytes(long n) { return (4 * (size_t)((n + FSC_TILE - 1) / FSC_TILE) + 2) * 8; }
// `scratch` (fe_scan_scratch_bytes(n), ZEROED by the caller -- one memset may cover the scratch of several scans) ; n > 0
static int fe_scan(const FeScanArgs &a, long n, void *scratch, hipStream_t st) {
    const long tiles = (n + FSC_TILE - 1) / FSC_TILE;
    k_fe_scan<<<(unsigned)tiles, FSC_T, 0, st>>>(a, n, (unsigned long long *)scratch, (uint32_t *)((unsigned long long *)scratch + 4 * tiles));
    HIPCHK(hipGetLastError());
    return HGX_OK;
}

// ---- the record stage (row 8a-1): fields, filters, key grouping -----------------------------------------------------------------
typedef FeLine LineRef;

__global__ void __launch_bounds__(256) k_fe_records(const char *__restrict__ text, size_t text_bytes, const LineRef *__restrict__ lines, uint32_t n, int binary,
                                                    int simulation, FeRec *__restrict__ recs, FeCtl *ctl) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    FeRec r;
    const int rc = binary ? fe_parse_bam_record(text, lines[i].off, lines[i].len, simulation != 0, lines[i].task, r)
                          : fe_parse_text_record(text, text_bytes, lines[i].off, lines[i].len, simulation != 0, lines[i].task, r);
    if (rc < 0) {
        // a record the kernels cannot take: the call declines (checked after the record stage); until then the record must be
        // inert -- no stale offset or length for the filters, the key table or the byte-for-byte compares to follow
        fe_decline(ctl, rc);
        r = FeRec{};
        r.bits = FE_R_FAILED;
        r.flag = 4;
        r.task = (uint16_t)lines[i].task;
    }
    recs[i] = r;
}
__global__ void k_fe_rec_heads(const FeRec *__restrict__ recs, uint32_t n, const char *__restrict__ text, uint8_t *__restrict__ head) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) head[i] = (i == 0 || !fe_same_read_id(recs[i - 1], recs[i], text)) ? 1 : 0;
}
// filters + insertion into the key table: slot claimed by the 64-bit key, representative = the FIRST record (atomicMin)
__global__ void __launch_bounds__(256) k_fe_rec_filter_insert(const FeRec *__restrict__ recs, const uint8_t *__restrict__ head, uint32_t n, FeFilter flt,
                                                              unsigned long long *__restrict__ tkeys, uint32_t *__restrict__ rep,
                                                              uint32_t *__restrict__ pile, uint32_t *__restrict__ anyk, uint32_t mask,
                                                              uint8_t *__restrict__ kept, uint32_t *__restrict__ slot_of, FeCtl *ctl) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int k = fe_rec_kept(recs, head, i, flt);
    if (k < 0) { fe_decline(ctl, k); k = 0; }
    kept[i] = (uint8_t)k;
    const bool pm = fe_rec_in_pileup(recs[i], flt);
    const unsigned long long h = recs[i].key;
    uint32_t slot = (uint32_t)h & mask;
    for (uint32_t probe = 0; probe <= mask; ++probe) {
        const unsigned long long cur = atomicCAS(&tkeys[slot], ~0ull, h);
        if (cur == ~0ull || cur == h) break;
        slot = (slot + 1) & mask;
    }
    atomicMin(&rep[slot], i);
    if (pm) atomicAdd(&pile[slot], 1u);
    if (k) atomicOr(&anyk[slot], 1u);
    slot_of[i] = slot;
}
// every record against its key's representative, byte for byte; and the flags the numbering is scanned from
__global__ void __launch_bounds__(256) k_fe_group_flags(const FeRec *__restrict__ recs, uint32_t n, const char *__restrict__ text,
                                                        const uint32_t *__restrict__ rep, const uint32_t *__restrict__ pile,
                                                        const uint32_t *__restrict__ anyk, const uint32_t *__restrict__ slot_of,
                                                        const uint8_t *__restrict__ kept, uint32_t *__restrict__ is_key, uint32_t *__restrict__ is_dec,
                                                        uint32_t *__restrict__ kept32, uint32_t *__restrict__ prev_in, FeCtl *ctl) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = slot_of[i], r = rep[s];
    if (r != i && !fe_rec_same_key(recs[r], recs[i], text)) fe_decline(ctl, -HGX_FE_DECLINE_COLLISION);
    const bool key = r == i && (pile[s] > 0 || anyk[s]);
    is_key[i] = key ? 1u : 0u;
    is_dec[i] = (key && anyk[s]) ? 1u : 0u;
    kept32[i] = kept[i];
    prev_in[i] = kept[i] ? i + 1 : 0u;
}
__global__ void __launch_bounds__(256) k_fe_build_keys(const FeRec *__restrict__ recs, uint32_t n, const uint32_t *__restrict__ is_key,
                                                       const uint32_t *__restrict__ is_dec, const uint32_t *__restrict__ key_idx,
                                                       const uint32_t *__restrict__ dec_idx, const uint32_t *__restrict__ kept32,
                                                       const uint32_t *__restrict__ rec_idx, const uint32_t *__restrict__ slot_of,
                                                       const uint32_t *__restrict__ pile, int base_locus, FeKey *__restrict__ keys,
                                                       uint32_t *__restrict__ dslot, FeCtl *ctl) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (i == n - 1) { ctl->n_keys = key_idx[i] + is_key[i]; ctl->n_slots = dec_idx[i] + is_dec[i]; ctl->n_rec = rec_idx[i] + kept32[i]; }
    if (!is_key[i]) return;
    const FeRec f = recs[i];
    FeKey K;
    K.pos = f.pos - (base_locus + 1);
    K.n_pile = pile[slot_of[i]];
    K.slot = is_dec[i] ? dec_idx[i] : FE_NO_SLOT;
    K.cigar_off = f.cigar_off; K.seq_off = f.seq_off; K.zs_off = f.zs_off; K.md_off = f.md_off;
    K.seq_len = f.seq_len; K.cigar_len = f.cigar_len; K.zs_len = f.zs_len; K.md_len = f.md_len;
    K.flags = (uint16_t)(((f.bits & FE_R_HAS_ZS) ? FE_K_HAS_ZS : 0) | ((f.bits & FE_R_HAS_MD) ? FE_K_HAS_MD : 0) |
                         ((f.bits & FE_R_BIN) ? (FE_K_BIN_CIGAR | FE_K_PACKED_SEQ) : 0));
    K.task = f.task;
    keys[key_idx[i]] = K;
    dslot[slot_of[i]] = K.slot;
}
__global__ void __launch_bounds__(256) k_fe_build_recinfo(const FeRec *__restrict__ recs, uint32_t n, const char *__restrict__ text,
                                                          const uint8_t *__restrict__ kept, const uint32_t *__restrict__ rec_idx,
                                                          const uint32_t *__restrict__ prev_kept, const uint32_t *__restrict__ slot_of,
                                                          const uint32_t *__restrict__ dslot, uint32_t *__restrict__ rec_info) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !kept[i]) return;
    const uint32_t pv = prev_kept[i];                     // index + 1 of the previous record that passed the filters, 0 = none
    const bool hd = pv == 0 || !fe_same_read_id(recs[pv - 1], recs[i], text);
    rec_info[rec_idx[i]] = dslot[slot_of[i]] | ((recs[i].flag & 0x40) ? 1u << 30 : 0u) | (hd ? 1u << 31 : 0u);
}

// ---- get_pair_interdist (typing_common.py:1187-1265) on the device: CODIS D18S51's expected inner distance (round 6) ----------
// The records that count (aligned, NH <= 1, YT:Z:CP) are numbered by a scan and compacted; lane j of the compacted list owns the run
// that starts at j: a distance iff the run holds exactly two records and a third counted record follows it (fe_interdist_* of
// hgx_front_core.hpp).  hist = FE_INTERDIST_BINS counters, zeroed by the caller; *m = the number of counted records (the scan's total).
static_assert(FE_INTERDIST_HALF == HGX_INTERDIST_HALF && FE_INTERDIST_BINS == HGX_INTERDIST_BINS, "the kernels' histogram is the one the shards exchange");
__global__ void k_fe_interdist_flag(const FeRec *__restrict__ recs, uint32_t n, uint32_t *__restrict__ flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = fe_rec_in_interdist(recs[i]) ? 1u : 0u;
}
__global__ void k_fe_interdist_compact(const uint32_t *__restrict__ flag, const uint32_t *__restrict__ idx, uint32_t n, uint32_t *__restrict__ comp) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flag[i]) comp[idx[i]] = i;
}
__global__ void __launch_bounds__(256) k_fe_interdist_hist(const FeRec *__restrict__ recs, const char *__restrict__ text, const uint32_t *__restrict__ comp,
                                                           const uint32_t *__restrict__ m_ptr, uint32_t *__restrict__ hist, FeCtl *ctl) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x, m = *m_ptr;
    if (j + 2 >= m) return;
    const FeRec a = recs[comp[j]], b = recs[comp[j + 1]];
    if (j > 0 && fe_same_read_id(recs[comp[j - 1]], a, text)) return;
    if (!fe_same_read_id(a, b, text) || fe_same_read_id(a, recs[comp[j + 2]], text)) return;
    long long d;
    const int rc = fe_interdist_of(a, b, text, d);
    if (rc < 0) { fe_decline(ctl, rc); return; }
    atomicAdd(&hist[fe_interdist_bin(d)], 1u);
}

// ---- BAM records straight from the inflated stream (round 4): chain walk, region filter, name sort as kernels -----------------
// What hgx_bam.cpp does on the host's threads when the records are not left to the device: the records of a BAM form a chain (each
// block_size leads to the next), so the stream is cut into ranges; every range but the first GUESSES a record start (a header that
// is plausible and leads to three more plausible headers) and walks from there past its end; the guesses are then CHECKED -- a
// range's walk must end exactly where the next one's begins -- and anything that does not link up declines the call.
struct BamCtl { int32_t decline; uint32_t n_rec, n_kept, max_klen, unsorted; uint32_t tot[4]; };      // tot: k_fe_scan's aggregates (records of the walk ranges)
__device__ __forceinline__ void bam_decline(BamCtl *c, int code) { atomicCAS(&c->decline, 0, code); }
__device__ __forceinline__ uint32_t bam_u32(const unsigned char *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ int32_t bam_i32(const unsigned char *p) { int32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint32_t bam_u16(const unsigned char *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
__device__ bool bam_plausible(const unsigned char *raw, size_t n, size_t o, int n_ref) {
    if (o + 36 > n) return false;
    const uint32_t bs = bam_u32(raw + o);
    if (bs < 32 || o + 4 + (size_t)bs > n) return false;
    const unsigned char *r = raw + o + 4;
    const int32_t rid = bam_i32(r), pos = bam_i32(r + 4), nrid = bam_i32(r + 20), npos = bam_i32(r + 24), l_seq = bam_i32(r + 16);
    const uint32_t l_rn = r[8], n_cig = bam_u16(r + 12);
    if (rid < -1 || rid >= n_ref || nrid < -1 || nrid >= n_ref) return false;
    if (pos < -1 || npos < -1 || l_seq < 0 || l_rn == 0) return false;
    if (32 + (size_t)l_rn + 4 * (size_t)n_cig + (size_t)(l_seq + 1) / 2 + (size_t)l_seq > bs) return false;
    if (r[32 + l_rn - 1] != 0) return false;
    for (uint32_t k = 0; k + 1 < l_rn; ++k) if (r[32 + k] < 33 || r[32 + k] > 126) return false;
    return true;
}
// the chain's cheaper test: the fields that tie a header to its block_size and the name's terminator -- every load independent of the
// others (the character loop of bam_plausible is a chain of dependent loads; it runs once, on the candidate whose chain held)
__device__ bool bam_header_fits(const unsigned char *raw, size_t n, size_t o, int n_ref) {
    if (o + 36 > n) return false;
    const uint32_t bs = bam_u32(raw + o);
    const unsigned char *r = raw + o + 4;
    const int32_t rid = bam_i32(r), l_seq = bam_i32(r + 16);
    const uint32_t l_rn = r[8], n_cig = bam_u16(r + 12);
    if (bs < 32 || o + 4 + (size_t)bs > n) return false;
    if (rid < -1 || rid >= n_ref || l_seq < 0 || l_rn == 0) return false;
    if (32 + (size_t)l_rn + 4 * (size_t)n_cig + (size_t)(l_seq + 1) / 2 + (size_t)l_seq > bs) return false;
    return r[32 + l_rn - 1] == 0;
}
struct BamRange { uint32_t first, stop, count, state; };     // (offsets inside the task's stream) state: 1 = walked, 2 = no record start in the range, 0 = a broken record
// a task's stream inside the device text (one task: the whole text), its header's verdicts and its share of the walk ranges
struct BamSeg {
    uint32_t base, n, body0;           // where the stream starts in the text, its bytes, its first record
    int32_t n_ref;
    uint32_t act_off;                  // its references' actions in the action table (0 drop, 1 keep, 2 keep where the span overlaps)
    uint32_t filtered;
    long long left0, right0;
    uint32_t first_range, n_ranges;
};
__device__ __forceinline__ int bam_seg_of(const BamSeg *__restrict__ segs, int n_seg, uint32_t range) {
    int lo = 0, hi = n_seg - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (segs[mid].first_range <= range) lo = mid; else hi = mid - 1; }
    return lo;
}
// PASS 0: find the range's first record and count; PASS 1: the same walk again, writing (offset after block_size, length, task)
template <int PASS>
__global__ void __launch_bounds__(64) k_bam_walk(const unsigned char *__restrict__ text, const BamSeg *__restrict__ segs, int n_seg, int W,
                                                 BamRange *__restrict__ rng, const uint32_t *__restrict__ base, uint32_t *__restrict__ rec_off,
                                                 uint32_t *__restrict__ rec_len, uint16_t *__restrict__ rec_task) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= W) return;
    const int sg = bam_seg_of(segs, n_seg, (uint32_t)t);
    const BamSeg G = segs[sg];
    const unsigned char *raw = text + G.base;
    const size_t n = G.n, body0 = G.body0;
    const size_t k = (size_t)t - G.first_range, Wg = G.n_ranges;
    const size_t lo = body0 + (n - body0) * k / Wg, hi = body0 + (n - body0) * (k + 1) / Wg;
    size_t o = lo;
    if (PASS == 0) {
        if (k > 0) {
            // The scan for a record start, in two alternating loops so that the lanes of a wavefront stay together: (1) slide an
            // 8-byte window (block_size, refID) to the next offset whose block_size is in [32, 2^24) and whose refID names a
            // reference or none -- cheap, lanes leave it at different trip counts and WAIT for each other at its exit; (2) the
            // full header check and the chain of three more headers -- a dozen dependent loads from cold lines, which all lanes
            // now run at the same time.  (In one loop every lane met its candidate in a different iteration and the wavefront
            // ran the 64 chains one after the other: 0.6 of this kernel's 0.7 ms.)
            // The window is fed eight bytes at a time from 8-byte-aligned loads (the lanes of a wavefront scan 64 different
            // lines: a byte load per step made 8x the gathers), the next eight requested a round ahead.
            bool found = false;
            const size_t n8 = n & ~(size_t)7;                         // (the text buffer is padded by 64 bytes beyond the last stream)
            auto load8 = [&](size_t at) -> uint64_t {                 // bytes [at, at + 8) of the stream, `at` 8-aligned in the text
                uint64_t v = 0;
                if (at + 8 <= n8 + 8) v = *reinterpret_cast<const uint64_t *>(raw + at);
                return v;
            };
            // bring o to where (raw + o) is 8-aligned, byte by byte (at most seven candidates go through the full check directly)
            uint64_t lo8 = 0, hi8 = 0, nxt8 = 0;
            int fed = 0;                                              // bytes of hi8 not yet shifted into lo8
            {
                const size_t mis = (size_t)(reinterpret_cast<uintptr_t>(raw + o) & 7u);
                const size_t o_al = o - mis;                          // (>= 0: the stream's base is 64-aligned and o >= body0 > mis)
                lo8 = load8(o_al) >> (8 * mis);
                hi8 = load8(o_al + 8);
                if (mis) { lo8 |= hi8 << (64 - 8 * mis); hi8 >>= 8 * mis; }
                fed = 8 - (int)mis;
                nxt8 = load8(o_al + 16);
            }
            size_t next_at = o + 8 + (size_t)fed;                     // stream offset of the first byte of nxt8
            auto step = [&]() {                                       // the window moves on by one byte
                lo8 = (lo8 >> 8) | (hi8 << 56);
                hi8 >>= 8;
                if (--fed == 0) { hi8 = nxt8; fed = 8; next_at += 8; nxt8 = load8(next_at); }
                ++o;
            };
            while (!found && o < hi) {
                for (;;) {
                    const uint32_t bs0 = (uint32_t)lo8;
                    const int32_t rid0 = (int32_t)(lo8 >> 32);
                    if (o >= hi || (bs0 >= 32u && bs0 < (1u << 24) && rid0 >= -1 && rid0 < G.n_ref)) break;
                    step();
                }
                if (o >= hi) break;
                {
                    size_t q = o;
                    int good = 0;
                    while (good < 4 && q < n && bam_header_fits(raw, n, q, G.n_ref)) { q += 4 + (size_t)bam_u32(raw + q); ++good; }
                    found = good > 0 && (good == 4 || q == n) && bam_plausible(raw, n, o, G.n_ref);
                }
                if (!found) step();
            }
            if (!found) { rng[t] = BamRange{(uint32_t)hi, (uint32_t)hi, 0u, 2u}; return; }
        }
    } else {
        if (rng[t].state != 1u) return;
        o = rng[t].first;
    }
    size_t q = o;
    uint32_t cnt = 0;
    bool ok = true;
    uint32_t at = PASS == 1 ? base[t] : 0u;
    while (q < hi && q < n) {
        if (q + 4 > n) { ok = false; break; }
        const uint32_t bs = bam_u32(raw + q);
        if (bs < 32 || q + 4 + (size_t)bs > n) { ok = false; break; }
        if (PASS == 1) { rec_off[at] = (uint32_t)(G.base + q + 4); rec_len[at] = bs; rec_task[at] = (uint16_t)sg; ++at; }
        ++cnt;
        q += 4 + (size_t)bs;
    }
    if (PASS == 0) rng[t] = BamRange{(uint32_t)o, (uint32_t)q, cnt, ok ? 1u : 0u};
}
// the ranges of a task must link up: every walked range begins where the walked range before it stopped (ranges without a record
// start -- a record longer than a range -- are passed over), the first at the first record, the last ends with the stream
__global__ void __launch_bounds__(256) k_bam_link(const BamRange *__restrict__ rng, const BamSeg *__restrict__ segs, int n_seg, int W,
                                                  uint32_t *__restrict__ cnt, BamCtl *ctl) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= W) return;
    const BamRange r = rng[t];
    cnt[t] = r.state == 1u ? r.count : 0u;
    if (r.state == 2u) return;
    if (r.state != 1u) { bam_decline(ctl, HGX_FE_DECLINE_RECORD); return; }
    const BamSeg G = segs[bam_seg_of(segs, n_seg, (uint32_t)t)];
    const int r0 = (int)G.first_range, r1 = r0 + (int)G.n_ranges;
    int p = t - 1;
    while (p >= r0 && rng[p].state == 2u) --p;
    const uint32_t expect = p < r0 ? G.body0 : rng[p].stop;
    if (r.first != expect) { bam_decline(ctl, HGX_FE_DECLINE_RECORD); return; }
    int q = t + 1;
    while (q < r1 && rng[q].state == 2u) ++q;
    if (q == r1 && r.stop != G.n) bam_decline(ctl, HGX_FE_DECLINE_RECORD);
}
// region filter (hgx_bam.cpp: reference span from the CIGAR, overlap with the one region) + the checks the host makes on a record
__global__ void __launch_bounds__(256) k_bam_filter(const unsigned char *__restrict__ text, const uint32_t *__restrict__ rec_off, const uint32_t *__restrict__ rec_len,
                                                    const uint16_t *__restrict__ rec_task, uint32_t n_rec, const BamSeg *__restrict__ segs,
                                                    const uint8_t *__restrict__ ref_action, uint32_t *__restrict__ keep, BamCtl *ctl) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rec) return;
    const unsigned char *r = text + rec_off[i];
    const uint32_t bs = rec_len[i], l_rn = r[8];
    const BamSeg G = segs[rec_task[i]];
    uint32_t k = 1;
    if (G.filtered) {
        const int32_t rid = bam_i32(r), pos = bam_i32(r + 4);
        if (rid < 0 || rid >= G.n_ref) k = 0;
        else {
            const uint32_t n_cig = bam_u16(r + 12), flag = bam_u16(r + 14);
            long long reflen = 0;
            if (!(flag & 4) && 32 + (size_t)l_rn + 4 * (size_t)n_cig <= bs) {
                const unsigned char *c = r + 32 + l_rn;
                for (uint32_t x = 0; x < n_cig; ++x) {
                    const uint32_t v = bam_u32(c + 4 * x), op = v & 15;
                    if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) reflen += v >> 4;
                }
            }
            const long long end0 = (long long)pos + (reflen > 0 ? reflen : 1) - 1;
            const uint8_t act = ref_action[G.act_off + (uint32_t)rid];
            k = act == 1 ? 1u : (act == 2 ? ((end0 >= G.left0 && (long long)pos <= G.right0) ? 1u : 0u) : 0u);
        }
    }
    if (k && (l_rn == 0 || 32 + (size_t)l_rn > bs || r[32 + l_rn - 1] != 0)) { bam_decline(ctl, HGX_FE_DECLINE_RECORD); k = 0; }   // "malformed BAM record": the host's to report
    keep[i] = k;
    if (k) atomicMax(&ctl->max_klen, l_rn - 1);
}
__global__ void k_bam_compact(const uint32_t *__restrict__ keep, const uint32_t *__restrict__ pos, uint32_t n_rec, uint32_t *__restrict__ idx, BamCtl *ctl) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rec) return;
    if (keep[i]) idx[pos[i]] = i;
    if (i == n_rec - 1) ctl->n_kept = pos[i] + keep[i];
}
// QNAME order as hgx_bam.cpp's line_less: bytes, a name that is a prefix of another first
__device__ __forceinline__ int bam_name_cmp(const unsigned char *a, uint32_t la, const unsigned char *b, uint32_t lb) {
    const uint32_t m = la < lb ? la : lb;
    for (uint32_t k = 0; k < m; ++k) if (a[k] != b[k]) return a[k] < b[k] ? -1 : 1;
    return la < lb ? -1 : (la > lb ? 1 : 0);
}
// bytes [8 c, 8 c + 8) of a name, big endian, zero beyond its end (one 8-byte load: it reaches at most 7 bytes beyond the name, into the
// record's next fields or the buffer's padding)
__device__ __forceinline__ unsigned long long name_chunk(const unsigned char *name, uint32_t klen, uint32_t c) {
    if (8 * c >= klen) return 0ull;
    unsigned long long w;
    __builtin_memcpy(&w, name + 8 * c, 8);
    unsigned long long v = __builtin_bswap64(w);
    const uint32_t have = klen - 8 * c;
    if (have < 8) v &= ~0ull << (8 * (8 - have));
    return v;
}
// The name sort on the bits that VARY (round 5, late).  An LSD radix sort over 8-byte chunks of the names costs ~25 library
// launches per chunk; read names differ in a few dozen bit positions (the digits of a counter).  The order check that runs anyway
// also ORs together, per chunk, the XOR of every two neighbouring names -- exactly the bit positions in which not all names
// agree -- and the sort keys are those bits alone, most significant first, packed into 64-bit words (one word, one sort over
// ~30 bits, for the usual names).  Dropping positions in which all keys agree changes no comparison.
#define NAME_DIFF_CHUNKS 32
struct NameDiff { unsigned long long m[NAME_DIFF_CHUNKS]; };
__device__ __forceinline__ void name_diff_add(const unsigned char *a, uint32_t la, const unsigned char *b, uint32_t lb, bool valid, unsigned long long *diff) {
    const uint32_t l = la > lb ? la : lb;
    const uint32_t nc_mine = valid ? (l + 7) / 8 : 0u;
    uint32_t nc = nc_mine;                                                  // chunks this wavefront walks: the longest pair's
#pragma unroll
    for (int d = 32; d; d >>= 1) nc = max(nc, (uint32_t)__shfl_xor((int)nc, d, 64));
    if (nc > NAME_DIFF_CHUNKS) nc = NAME_DIFF_CHUNKS;                       // (longer names: the caller sorts chunk by chunk)
    for (uint32_t c = 0; c < nc; ++c) {
        unsigned long long x = c < nc_mine ? name_chunk(a, la, c) ^ name_chunk(b, lb, c) : 0ull;
#pragma unroll
        for (int d = 32; d; d >>= 1) x |= (unsigned long long)__shfl_xor((long long)x, d, 64);
        if ((threadIdx.x & 63) == 0 && x) atomicOr(&diff[c], x);
    }
}
// word `word` (0 = least significant) of the packed key: the varying bits of all chunks, most significant first
__device__ __forceinline__ unsigned long long name_packed_word(const unsigned char *name, uint32_t klen, const NameDiff &D, int n_chunks, int n_bits, int word) {
    unsigned long long out = 0;
    int p = n_bits;                                                         // bits still to place (the next goes to position p - 1)
    for (int c = 0; c < n_chunks && p > 64 * word; ++c) {
        unsigned long long m = D.m[c];
        if (!m) continue;
        const unsigned long long v = name_chunk(name, klen, (uint32_t)c);
        while (m) {
            const int b = 63 - __builtin_clzll(m);
            m &= ~(1ull << b);
            --p;
            if ((p >> 6) == word) out |= ((v >> b) & 1ull) << (p & 63);
        }
    }
    return out;
}
__global__ void __launch_bounds__(256) k_bam_sorted(const unsigned char *__restrict__ text, const uint32_t *__restrict__ rec_off, const uint16_t *__restrict__ rec_task,
                                                    const uint32_t *__restrict__ idx, uint32_t n, BamCtl *ctl, unsigned long long *__restrict__ diff) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i > 0 && i < n;
    const unsigned char *a = text + rec_off[idx[valid ? i - 1 : 0]], *b = text + rec_off[idx[valid ? i : 0]];
    const uint32_t la = (uint32_t)a[8] - 1, lb = (uint32_t)b[8] - 1;
    if (diff) name_diff_add(a + 32, la, b + 32, lb, valid, diff);          // (names of more than one 8-byte chunk: the sort will want it)
    if (!valid) return;
    if (rec_task[idx[i - 1]] != rec_task[idx[i]]) return;                  // (records are in task order before the sort: only names inside a task matter)
    if (bam_name_cmp(b + 32, lb, a + 32, la) < 0) ctl->unsorted = 1;
}
__global__ void __launch_bounds__(256) k_bam_name_key_packed(const unsigned char *__restrict__ text, const uint32_t *__restrict__ rec_off, const uint32_t *__restrict__ idx,
                                                             uint32_t n, NameDiff D, int n_chunks, int n_bits, int word, unsigned long long *__restrict__ key) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned char *r = text + rec_off[idx[i]];
    key[i] = name_packed_word(r + 32, (uint32_t)r[8] - 1, D, n_chunks, n_bits, word);
}
// bytes [8 c, 8 c + 8) of every name, big endian, zero beyond its end: LSD radix passes over these give the byte order above
__global__ void __launch_bounds__(256) k_bam_name_key(const unsigned char *__restrict__ text, const uint32_t *__restrict__ rec_off, const uint32_t *__restrict__ idx,
                                                      uint32_t n, uint32_t chunk, unsigned long long *__restrict__ key) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned char *r = text + rec_off[idx[i]];
    key[i] = name_chunk(r + 32, (uint32_t)r[8] - 1, chunk);
}
__global__ void k_bam_task_key(const uint16_t *__restrict__ rec_task, const uint32_t *__restrict__ idx, uint32_t n, unsigned long long *__restrict__ key) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) key[i] = rec_task[idx[i]];
}
__global__ void k_bam_lines(const uint32_t *__restrict__ rec_off, const uint32_t *__restrict__ rec_len, const uint16_t *__restrict__ rec_task,
                            const uint32_t *__restrict__ idx, uint32_t n, FeLine *__restrict__ lines) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const uint32_t r = idx[i]; lines[i] = FeLine{rec_off[r], rec_len[r], (uint32_t)rec_task[r]}; }
}

thread_local int g_last_device = 0, g_last_decline = 0, g_last_route = 0, g_last_parts = 0;
thread_local long long g_last_bytes = 0;      // bytes the last call sent to the device (text / inflated stream / key table)

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---- the device stages of one call ------------------------------------------------------------------------------------------------
struct DevInput {                     // keys, their text and the kept records, resident in HBM; ctl zeroed (or carrying a decline code)
    const FeKey *keys; uint32_t n_keys;
    const char *text;
    const uint32_t *rec_info; uint32_t n_rec;
    uint32_t n_slots;
    FeCtl *ctl;
    // a many-task batch (the samples of one locus in one pass): a pileup per task, and per task the reads, pairs and refs it
    // contributed (device arrays, zeroed by the caller; NULL for one task)
    int n_tasks = 1;
    uint32_t *task_reads = nullptr, *task_pairs = nullptr, *task_pieces = nullptr;
    unsigned long long *task_refs = nullptr;
    const hgx_locus *host_locus = nullptr;     // keep_trace: the variant names the trace lines spell
    // CODIS D18S51 (codis_choose_pairs / interdist_exchange): the histogram of this stream's inner distances -- counted on the device
    // by the record route (d_hist: FE_INTERDIST_BINS u32 counters), made by the host stages on the key route (h_hist)
    bool want_interdist = false;
    const uint32_t *d_hist = nullptr;
    const std::vector<int64_t> *h_hist = nullptr;
};
struct Lap {
    bool prof; hipStream_t st; double t_prev;
    explicit Lap(hipStream_t s) : prof(getenv("HGX_PARSE_PROFILE") != nullptr), st(s), t_prev(now_ms()) {}
    void operator()(const char *what) {
        if (!prof) return;
        (void)hipStreamSynchronize(st);
        const double t = now_ms();
        fprintf(stderr, "[hgx_front]     %-28s %8.2f ms\n", what, t - t_prev);
        t_prev = t;
    }
};

int front_stages(const FeLocus &F, const DevInput &di, const hgx_parse_opts &o, hipStream_t st, hgx_dbatch **out, int *declined) {
    *out = nullptr;
    *declined = 0;
    Lap lap(st);
    int rc = HGX_OK;
    const int n_ref = F.n_ref;
    const uint32_t n_keys = di.n_keys, n_rec = di.n_rec, S = di.n_slots;
    // every buffer of the call is declared here, the guard after them: on ANY way out the stream is drained first, then the
    // buffers go back to the pool
    DevBuf b_state, b_koff, b_knht, b_ht, b_clo, b_cnw, b_ckey, b_cmoff, b_mpool, b_slot_task, b_kpos, b_kpos2, b_kord, b_kord2, b_ktmp;
    DevBuf b_cnt, b_off, b_tmp, b_key_s, b_idx, b_flag, b_rank, b_head_of, b_head_cand;
    DevBuf b_wh, b_wh_s, b_hidx, b_ord, b_nw2, b_moff, b_new_id, b_cand_piece, b_tbits, b_trace, b_troff, b_scan;
    hgx_dbatch *d = new hgx_dbatch();
    struct Guard { hgx_dbatch *&d; hipStream_t st; ~Guard() { (void)hipStreamSynchronize(st); if (d) hgx_dbatch_destroy(d); } } guard{d, st};
    FeCtl *ctl = di.ctl;
    const FeKey *keys = di.keys;
    const char *text = di.text;
    const uint32_t *rec_info = di.rec_info;
    (void)rc;

    // pileup
    const int n_tasks = std::max(1, di.n_tasks);
    if (n_tasks > 1 && (o.pileup_exchange || o.pileup_exchange_dev)) { *declined = HGX_FE_DECLINE_OPTS; return HGX_OK; }
    if (n_tasks > 65535) { *declined = HGX_FE_DECLINE_SIZE; return HGX_OK; }
    d->n_ref = n_ref;
    const size_t n_cells = (size_t)n_tasks * n_ref * 6;
    d->d_counts = (uint32_t *)hgx_pool_alloc(std::max<size_t>((n_cells + 1) * 4, 16));      // (+ 1: the spare element of pileup_exchange_dev)
    d->d_nt_set = (uint8_t *)hgx_pool_alloc(std::max<size_t>((size_t)n_tasks * n_ref, 16));
    if (!d->d_counts || !d->d_nt_set) { hgx_set_error("device allocation of the pileup tables failed"); return HGX_ENOMEM; }
    HIPCHK(hipMemsetAsync(d->d_counts, 0, (n_cells + 1) * 4, st));
    if (n_keys && n_ref > 0) {
        const int tile = std::min(n_ref, 6000);                       // 6 counters x 4 bytes x 6000 positions = 144 KB of LDS
        const size_t lds = (size_t)tile * 6 * 4;
        HGX_ONCE_PER_DEVICE(HIPCHK(hipFuncSetAttribute((const void *)k_fe_pileup, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)));
        const unsigned n_tiles = (unsigned)((n_ref + tile - 1) / tile);
        // one task: up to 256 blocks share the keys; many: a task's few thousand keys go to 1-4 blocks (256 CUs / tasks)
        const unsigned per_task = (unsigned)std::max(1, std::min(4, 256 / n_tasks));
        const unsigned nb = n_tasks > 1 ? per_task : std::max(1u, std::min(256u, (n_keys + 63) / 64));
        k_fe_pileup<<<dim3(nb, n_tiles, (unsigned)n_tasks), 1024, lds, st>>>(keys, n_keys, text, n_ref, tile, d->d_counts, ctl);
    }
    // intra-locus read sharding (8e): this shard's counters -> the sum over all shards.  On the device where the caller gave the
    // device form (hgx_allreduce_sum_u32 over RCCL on this very buffer: no host bounce); the summed table is kept for the host
    // stages in case a later stage declines -- a rank communicates once per parse (hgx_pileup_share)
    hgx_pileup_share *share = o.pileup_exchange == &hgx_pileup_share::trampoline ? (hgx_pileup_share *)o.pileup_ctx : nullptr;
    if (o.pileup_exchange_dev && n_ref > 0 && !(share && share->have_sum)) {
        if (o.pileup_exchange_dev(o.pileup_dev_ctx, d->d_counts, (int64_t)n_cells + 1, (void *)st) != 0) {
            hgx_set_error("pileup exchange between the ranks of a sharded locus failed");
            return HGX_EINVAL;
        }
        if (share) {
            share->sum.resize(n_cells);
            HIPCHK(hipMemcpyAsync(share->sum.data(), d->d_counts, n_cells * 4, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            share->have_sum = true;
        }
    } else if (o.pileup_exchange && n_ref > 0) {
        std::vector<uint32_t> h((size_t)n_ref * 6);
        HIPCHK(hipMemcpyAsync(h.data(), d->d_counts, h.size() * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (o.pileup_exchange(o.pileup_ctx, h.data(), (int64_t)h.size()) != 0) {
            hgx_set_error("pileup exchange between the ranks of a sharded locus failed");
            return HGX_EINVAL;
        }
        HIPCHK(hipMemcpyAsync(d->d_counts, h.data(), h.size() * 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    if (n_ref > 0) k_fe_nt_set<<<nblk((long)n_tasks * n_ref, 256), 256, 0, st>>>(d->d_counts, n_tasks * n_ref, d->d_nt_set);
    lap("pileup");
    // CODIS D18S51: the sample's expected inner distance (get_pair_interdist) -- the histogram of this stream's distances, summed
    // over the shards of a sharded locus (AFTER the pileup exchange: the order of the host stages' exchanges), its middle element
    long long expected = -1;
    const bool choose = di.want_interdist && o.codis_choose_pairs;
    if (di.want_interdist) {
        if (n_tasks > 1 || (!di.d_hist && (!di.h_hist || di.h_hist->size() != (size_t)HGX_INTERDIST_BINS))) { *declined = HGX_FE_DECLINE_OPTS; return HGX_OK; }
        std::vector<int64_t> hist((size_t)HGX_INTERDIST_BINS);
        if (di.d_hist) {
            std::vector<uint32_t> h32((size_t)HGX_INTERDIST_BINS);
            HIPCHK(hipMemcpyAsync(h32.data(), di.d_hist, h32.size() * 4, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            for (size_t b = 0; b < h32.size(); ++b) hist[b] = (int64_t)h32[b];
        } else hist = *di.h_hist;
        if (o.interdist_exchange && o.interdist_exchange(o.interdist_ctx, hist.data(), (int64_t)hist.size()) != 0) {
            hgx_set_error("inter-distance exchange between the ranks of a sharded locus failed");
            return HGX_EINVAL;
        }
        if (hgx_interdist_median(hist.data(), &expected)) { *declined = HGX_FE_DECLINE_SIZE; return HGX_OK; }     // (outside the histogram: the host's sort)
        lap("expected inner distance");
    }

    // decode
    const size_t ht_cap = (size_t)S * 48 + 4096, cand_cap = (size_t)S * 12 + 4096, mask_cap = cand_cap * 16;
    if (ht_cap >= (1ull << 32) || mask_cap >= (1ull << 32)) { *declined = HGX_FE_DECLINE_SIZE; return HGX_OK; }
    ALLOC(b_state, std::max<size_t>(S, 16));
    ALLOC(b_koff, std::max<size_t>(S, 4) * 4);
    ALLOC(b_knht, std::max<size_t>(S, 4) * 4);
    ALLOC(b_ht, ht_cap * 4);
    ALLOC(b_clo, cand_cap * 2);
    ALLOC(b_cnw, cand_cap * 2);
    ALLOC(b_ckey, cand_cap * 8);
    ALLOC(b_cmoff, cand_cap * 4);
    ALLOC(b_mpool, mask_cap * 4);
    if (n_tasks > 1) ALLOC(b_slot_task, std::max<size_t>(S, 8) * 2);
    FePools pools;
    pools.ht_pool = b_ht.as<int32_t>(); pools.ht_cap = (uint32_t)ht_cap; pools.ht_cursor = &ctl->ht_cursor;
    pools.cand_lo = b_clo.as<uint16_t>(); pools.cand_nw = b_cnw.as<uint16_t>(); pools.cand_key = b_ckey.as<uint64_t>();
    pools.cand_mask_off = b_cmoff.as<uint32_t>(); pools.cand_cap = (uint32_t)cand_cap; pools.cand_cursor = &ctl->cand_cursor;
    pools.mask_pool = b_mpool.as<uint32_t>(); pools.mask_cap = (uint32_t)mask_cap; pools.mask_cursor = &ctl->mask_cursor;
    pools.trace_pool = nullptr; pools.trace_cap = 0; pools.trace_cursor = &ctl->trace_cursor; pools.key_trace_off = nullptr;
    const size_t trace_cap = (size_t)S * 96 + 4096;
    const bool tracing = o.keep_trace && n_tasks == 1 && di.host_locus && trace_cap < (1ull << 31);
    if (tracing) {                                                    // (tests: the intermediates of every decoded key, see FePools)
        ALLOC(b_trace, trace_cap * 4);
        ALLOC(b_troff, std::max<size_t>(S, 4) * 4);
        HIPCHK(hipMemsetAsync(b_troff.p, 0xFF, std::max<size_t>(S, 4) * 4, st));
        pools.trace_pool = b_trace.as<int32_t>(); pools.trace_cap = (uint32_t)trace_cap; pools.key_trace_off = b_troff.as<uint32_t>();
    }
    const FeParse po{o.num_editdist, o.error_correction};
    const FePile pile{d->d_nt_set, d->d_counts};
    if (n_keys) {
        // decode in backbone order of the keys: neighbours in a wavefront then walk the same stretch of the variant list and take the
        // same branches (the result does not depend on who decodes what: pools are filled through cursors, the piece table is
        // ordered by content)
        const uint32_t *order = nullptr;
        if (n_keys >= 4096 && n_tasks == 1 && n_ref > 0) {
            const uint32_t bins = (uint32_t)n_ref + 4096u + 1u;
            const size_t bin_bytes = ((size_t)bins * 4 + 255) & ~(size_t)255, sc = fe_scan_scratch_bytes(bins);
            ALLOC(b_ktmp, 2 * bin_bytes + sc);                         // histogram | offsets | the scan's tile states
            ALLOC(b_kord2, (size_t)n_keys * 4);
            HIPCHK(hipMemsetAsync(b_ktmp.p, 0, 2 * bin_bytes + sc, st));
            uint32_t *const hist = b_ktmp.as<uint32_t>(), *const offs = (uint32_t *)((char *)b_ktmp.p + bin_bytes);
            k_fe_pos_hist<<<nblk(n_keys, 256), 256, 0, st>>>(keys, n_keys, bins, hist);
            FeScanArgs sa{};
            sa.n_ch = 1;
            sa.ch[0] = FeScanCh{hist, offs, 0, FSC_U32};
            rc = fe_scan(sa, (long)bins, (char *)b_ktmp.p + 2 * bin_bytes, st);
            if (rc) return rc;
            k_fe_pos_scatter<<<nblk(n_keys, 256), 256, 0, st>>>(keys, n_keys, bins, offs, b_kord2.as<uint32_t>());
            order = b_kord2.as<uint32_t>();
        } else if (n_keys >= 4096) {
            ALLOC(b_kpos, (size_t)n_keys * 4); ALLOC(b_kpos2, (size_t)n_keys * 4); ALLOC(b_kord, (size_t)n_keys * 4); ALLOC(b_kord2, (size_t)n_keys * 4);
            k_fe_key_pos<<<nblk(n_keys, 256), 256, 0, st>>>(keys, n_keys, b_kpos.as<uint32_t>(), b_kord.as<uint32_t>());
            size_t tb = 0;
            (void)hipcub::DeviceRadixSort::SortPairs((void *)nullptr, tb, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (int)n_keys, 0, 32, st);
            ALLOC(b_ktmp, std::max<size_t>(tb, 256));
            int key_bits = 21;                                     // (task << 20) ^ (position + 4096): only the bits in use are sorted on
            while (key_bits < 32 && (1 << (key_bits - 20)) < n_tasks + 1) ++key_bits;
            HIPCHK(hipcub::DeviceRadixSort::SortPairs(b_ktmp.p, tb, b_kpos.as<uint32_t>(), b_kpos2.as<uint32_t>(), b_kord.as<uint32_t>(), b_kord2.as<uint32_t>(), (int)n_keys, 0, key_bits, st));
            order = b_kord2.as<uint32_t>();
        }
        k_fe_decode<<<nblk(n_keys, FE_DECODE_BLOCK), FE_DECODE_BLOCK, 0, st>>>(F, po, pile, keys, n_keys, text, pools, b_state.as<uint8_t>(), b_koff.as<uint32_t>(),
                                                       b_knht.as<uint32_t>(), n_tasks > 1 ? b_slot_task.as<uint16_t>() : (uint16_t *)nullptr, order, ctl);
    }
    // the pair counts need nothing but the decode results
    ALLOC(b_cnt, std::max<size_t>(n_rec, 1) * 8);
    ALLOC(b_off, std::max<size_t>(n_rec, 1) * 8);
    if (n_rec) k_fe_pair_count<<<nblk(n_rec, 256), 256, 0, st>>>(rec_info, n_rec, b_state.as<uint8_t>(), b_koff.as<uint32_t>(), b_knht.as<uint32_t>(),
                                                                 b_ht.as<int32_t>(), b_cnt.as<unsigned long long>(), ctl,
                                                                 n_tasks > 1 ? b_slot_task.as<uint16_t>() : (const uint16_t *)nullptr,
                                                                 n_tasks > 1 ? di.task_reads : (uint32_t *)nullptr, di.task_pairs, di.task_refs, choose ? 1 : 0);
    if (n_rec && choose) k_fe_pair_choose<<<1, 64, 0, st>>>(rec_info, n_rec, b_state.as<uint8_t>(), b_koff.as<uint32_t>(), b_knht.as<uint32_t>(),
                                                            b_ht.as<int32_t>(), b_cnt.as<unsigned long long>(), ctl, expected);
    FeCtl h;
    { const int rc_d = hgx_d2h(&h, ctl, sizeof(FeCtl), st); if (rc_d) { (void)hgx_sync(st); return rc_d; } }
    { const int rc_s = hgx_sync(st); if (rc_s) return rc_s; }
    lap("decode + pair counts");
    if (h.decline) { *declined = -h.decline; return HGX_OK; }
    const uint32_t n_cand = h.cand_cursor, choose_head = h.last_head;
    d->n_reads = (int32_t)h.n_reads;
    d->n_gene_refs = (int64_t)h.n_gene_refs;

    // temp storage for the sorts (one block, the largest request)
    size_t tmp_bytes = 0;
    {
        size_t b = 0;
        (void)hipcub::DeviceRadixSort::SortPairs((void *)nullptr, b, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                           (int)std::max<uint32_t>(n_cand, 1), 0, 64, st);
        tmp_bytes = std::max(tmp_bytes, b);
    }
    ALLOC(b_tmp, std::max<size_t>(tmp_bytes, 256));
    // one zeroed block for the tile states of the three scans of this function
    const size_t sc_pair = fe_scan_scratch_bytes(std::max<uint32_t>(n_rec, 1)), sc_cand = fe_scan_scratch_bytes(std::max<uint32_t>(n_cand, 1));
    ALLOC(b_scan, sc_pair + 2 * sc_cand);
    HIPCHK(hipMemsetAsync(b_scan.p, 0, sc_pair + 2 * sc_cand, st));
    if (n_rec) {
        // pair index and first ref of every run that yields a pair: two channels of the packed counts in one pass
        FeScanArgs sa{};
        sa.n_ch = 2;
        sa.ch[0] = FeScanCh{b_cnt.p, b_off.as<uint32_t>(), 0, FSC_PAIR_FLAG};
        sa.ch[1] = FeScanCh{b_cnt.p, b_off.as<uint32_t>() + n_rec, 0, FSC_PAIR_REFS};
        rc = fe_scan(sa, (long)n_rec, b_scan.p, st);
        if (rc) return rc;
    }
    // distinct pieces
    uint32_t tab_cap = 1024;
    while (tab_cap < 2 * (uint64_t)n_cand) tab_cap <<= 1;
    ALLOC(b_key_s, (size_t)tab_cap * 12);                          // the table: keys [cap] u64 | representatives [cap] u32, all ones
    ALLOC(b_idx, std::max<size_t>(n_cand, 1) * 4);                  // slot of every candidate
    ALLOC(b_flag, std::max<size_t>(n_cand, 1) * 4);
    ALLOC(b_rank, std::max<size_t>(n_cand, 1) * 4);
    ALLOC(b_head_of, std::max<size_t>(n_cand, 1) * 4);
    ALLOC(b_head_cand, std::max<size_t>(n_cand, 1) * 4);
    if (n_cand) {
        unsigned long long *const tkeys = b_key_s.as<unsigned long long>();
        uint32_t *const rep = (uint32_t *)((char *)b_key_s.p + (size_t)tab_cap * 8);
        HIPCHK(hipMemsetAsync(b_key_s.p, 0xFF, (size_t)tab_cap * 12, st));
        k_fe_cand_insert<<<nblk(n_cand, 256), 256, 0, st>>>(b_ckey.as<uint64_t>(), n_cand, tkeys, rep, tab_cap - 1, b_idx.as<uint32_t>());
        k_fe_cand_flags<<<nblk(n_cand, 256), 256, 0, st>>>(b_idx.as<uint32_t>(), rep, n_cand, pools.cand_lo, pools.cand_nw, pools.cand_mask_off, pools.mask_pool,
                                                           b_flag.as<uint32_t>(), ctl);
        {
            FeScanArgs sa{};
            sa.n_ch = 1;
            sa.ch[0] = FeScanCh{b_flag.p, b_rank.as<uint32_t>(), 0, FSC_U32};
            rc = fe_scan(sa, (long)n_cand, (char *)b_scan.p + sc_pair, st);
            if (rc) return rc;
        }
        k_fe_cand_assign<<<nblk(n_cand, 256), 256, 0, st>>>(b_idx.as<uint32_t>(), rep, b_flag.as<uint32_t>(), b_rank.as<uint32_t>(), n_cand,
                                                            b_head_of.as<uint32_t>(), b_head_cand.as<uint32_t>(), ctl);
    }
    { const int rc_d = hgx_d2h(&h, ctl, sizeof(FeCtl), st); if (rc_d) { (void)hgx_sync(st); return rc_d; } }
    { const int rc_s = hgx_sync(st); if (rc_s) return rc_s; }
    lap("distinct pieces");
    if (h.decline) { *declined = -h.decline; return HGX_OK; }
    const uint32_t n_heads = n_cand ? h.n_heads : 0;
    const uint64_t n_pairs64 = h.pair_total >> 40, n_refs64 = h.pair_total & ((1ull << 40) - 1);        // (summed by k_fe_pair_count in 64 bits)
    if (n_refs64 >= (1ull << 31) || n_pairs64 >= (1ull << 31)) { *declined = HGX_FE_DECLINE_SIZE; return HGX_OK; }
    const uint32_t n_pairs = (uint32_t)n_pairs64, n_refs = (uint32_t)n_refs64;

    // canonical order of the distinct pieces, the piece table, the refs
    ALLOC(b_wh, std::max<size_t>(n_heads, 1) * 8);
    ALLOC(b_wh_s, std::max<size_t>(n_heads, 1) * 8);
    ALLOC(b_hidx, std::max<size_t>(n_heads, 1) * 4);
    ALLOC(b_ord, std::max<size_t>(n_heads, 1) * 4);
    ALLOC(b_nw2, std::max<size_t>(n_heads, 1) * 4);
    ALLOC(b_moff, std::max<size_t>(n_heads, 1) * 4);
    ALLOC(b_new_id, std::max<size_t>(n_heads, 1) * 4);
    ALLOC(b_cand_piece, std::max<size_t>(n_cand, 1) * 4);
    d->n_pieces = (int32_t)n_heads;
    d->n_pairs = (int32_t)n_pairs;
    d->n_refs = (int64_t)n_refs;
    d->d_pieces = (hgx_piece *)hgx_pool_alloc(std::max<size_t>((size_t)n_heads * sizeof(hgx_piece), 16));
    d->d_masks = (uint32_t *)hgx_pool_alloc(std::max<size_t>((size_t)h.mask_cursor * 4, 16));
    d->d_pair_off = (int32_t *)hgx_pool_alloc(((size_t)n_pairs + 1) * 4);
    d->d_pair_ref = (uint32_t *)hgx_pool_alloc(std::max<size_t>((size_t)n_refs * 4, 16));
    if (!d->d_pieces || !d->d_masks || !d->d_pair_off || !d->d_pair_ref) { hgx_set_error("device allocation of the piece batch failed"); return HGX_ENOMEM; }
    if (n_heads) {
        const int hash_shift = hgx_switch_has("front", "tie_test") ? 56 : 32;
        k_fe_head_keys<<<nblk(n_heads, 256), 256, 0, st>>>(b_head_cand.as<uint32_t>(), n_heads, pools.cand_lo, pools.cand_nw, pools.cand_mask_off,
                                                          pools.mask_pool, b_wh.as<uint64_t>(), b_hidx.as<uint32_t>(), hash_shift);
        size_t b = tmp_bytes;                                      // (n_heads <= n_cand: the block is large enough)
        HIPCHK(hipcub::DeviceRadixSort::SortPairs(b_tmp.p, b, b_wh.as<uint64_t>(), b_wh_s.as<uint64_t>(), b_hidx.as<uint32_t>(), b_ord.as<uint32_t>(),
                                                  (int)n_heads, 0, 64, st));
        k_fe_tie_fix<<<nblk(n_heads, 256), 256, 0, st>>>(b_ord.as<uint32_t>(), n_heads, b_head_cand.as<uint32_t>(), pools.cand_lo, pools.cand_nw,
                                                        pools.cand_mask_off, pools.mask_pool, hash_shift);
        k_fe_piece_sizes<<<nblk(n_heads, 256), 256, 0, st>>>(b_ord.as<uint32_t>(), b_head_cand.as<uint32_t>(), n_heads, pools.cand_nw, b_nw2.as<uint32_t>());
        {
            FeScanArgs sa{};
            sa.n_ch = 1;
            sa.ch[0] = FeScanCh{b_nw2.p, b_moff.as<uint32_t>(), 0, FSC_U32};
            rc = fe_scan(sa, (long)n_heads, (char *)b_scan.p + sc_pair + sc_cand, st);
            if (rc) return rc;
        }
        k_fe_write_pieces<<<nblk(n_heads, 256), 256, 0, st>>>(b_ord.as<uint32_t>(), b_head_cand.as<uint32_t>(), n_heads, pools.cand_lo, pools.cand_nw,
                                                             pools.cand_mask_off, pools.mask_pool, b_moff.as<uint32_t>(), d->d_pieces, d->d_masks,
                                                             b_new_id.as<uint32_t>(), ctl);
        k_fe_cand_piece<<<nblk(n_cand, 256), 256, 0, st>>>(b_head_of.as<uint32_t>(), b_new_id.as<uint32_t>(), n_cand, b_cand_piece.as<uint32_t>());
        if (n_tasks > 1 && di.task_pieces && S) {
            const uint32_t wpt = (n_heads + 31) / 32;
            ALLOC(b_tbits, (size_t)n_tasks * wpt * 4);
            HIPCHK(hipMemsetAsync(b_tbits.p, 0, (size_t)n_tasks * wpt * 4, st));
            k_fe_task_piece_bits<<<nblk(S, 256), 256, 0, st>>>(b_state.as<uint8_t>(), b_koff.as<uint32_t>(), b_knht.as<uint32_t>(), b_ht.as<int32_t>(),
                                                               b_cand_piece.as<uint32_t>(), b_slot_task.as<uint16_t>(), S, wpt, b_tbits.as<uint32_t>());
            k_fe_task_piece_count<<<n_tasks, 256, 0, st>>>(b_tbits.as<uint32_t>(), wpt, di.task_pieces);
        }
    }
    if (n_rec) k_fe_pair_emit<<<nblk(n_rec, 256), 256, 0, st>>>(rec_info, n_rec, b_state.as<uint8_t>(), b_koff.as<uint32_t>(), b_knht.as<uint32_t>(),
                                                                b_ht.as<int32_t>(), b_cnt.as<unsigned long long>(), b_off.as<uint32_t>(), b_off.as<uint32_t>() + n_rec,
                                                                b_cand_piece.as<uint32_t>(), d->d_pair_off, d->d_pair_ref, n_pairs, n_refs,
                                                                choose ? choose_head : 0u, expected);
    else HIPCHK(hipMemsetAsync(d->d_pair_off, 0, 4, st));
    { const int rc_d = hgx_d2h(&h, ctl, sizeof(FeCtl), st); if (rc_d) { (void)hgx_sync(st); return rc_d; } }
    { const int rc_s = hgx_sync(st); if (rc_s) return rc_s; }
    lap("piece table + refs");
    d->n_mask_u32 = n_heads ? (int64_t)h.n_masks : 0;
    d->sum_piece_words = d->n_mask_u32 / 2;
    if (tracing && n_rec) {
        std::vector<uint8_t> h_state(std::max<size_t>(S, 1));
        std::vector<uint32_t> h_troff(std::max<size_t>(S, 1)), h_rec(n_rec);
        std::vector<int32_t> h_pool(std::max<size_t>(h.trace_cursor, 1));
        if (S) HIPCHK(hipMemcpyAsync(h_state.data(), b_state.p, S, hipMemcpyDeviceToHost, st));
        if (S) HIPCHK(hipMemcpyAsync(h_troff.data(), b_troff.p, (size_t)S * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(h_rec.data(), rec_info, (size_t)n_rec * 4, hipMemcpyDeviceToHost, st));
        if (h.trace_cursor) HIPCHK(hipMemcpyAsync(h_pool.data(), b_trace.p, (size_t)h.trace_cursor * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        hgx_front_trace_lines(*di.host_locus, h_rec.data(), n_rec, h_state.data(), h_troff.data(), h_pool.data(), d->trace);
    }
    *out = d;
    d = nullptr;                       // (the guard keeps its hands off)
    return HGX_OK;
}

// the key route: the host stages made the key table; upload it, then the stages above
int front_run(hgx_locus &L, const hgx_front_input &in, const hgx_parse_opts &o, hipStream_t st, hgx_dbatch **out, int *declined) {
    *out = nullptr;
    *declined = 0;
    Lap lap(st);
    const FeLocus *Fp = nullptr;
    bool usable = false;
    int rc = dev_locus(L, st, &Fp, &usable);
    if (rc) return rc;
    if (!usable) { *declined = HGX_FE_DECLINE_LOCUS; return HGX_OK; }
    if (in.n_keys >= (1ull << 31) || in.n_rec >= (1ull << 31)) { *declined = HGX_FE_DECLINE_SIZE; return HGX_OK; }
    DevBuf b_keys, b_text, b_rec, b_ctl;
    struct Drain { hipStream_t s; ~Drain() { (void)hipStreamSynchronize(s); } } drain{st};
    ALLOC(b_keys, std::max<size_t>(in.n_keys, 1) * sizeof(FeKey));
    ALLOC(b_text, in.n_text + 64);
    ALLOC(b_rec, std::max<size_t>(in.n_rec, 1) * 4);
    ALLOC(b_ctl, sizeof(FeCtl));
    if (in.n_keys) HIPCHK(hipMemcpyAsync(b_keys.p, in.keys, in.n_keys * sizeof(FeKey), hipMemcpyHostToDevice, st));
    if (in.n_text) HIPCHK(hipMemcpyAsync(b_text.p, in.text, in.n_text, hipMemcpyHostToDevice, st));
    if (in.n_rec) HIPCHK(hipMemcpyAsync(b_rec.p, in.rec_info, in.n_rec * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(b_ctl.p, 0, sizeof(FeCtl), st));
    g_last_bytes += (long long)(in.n_keys * sizeof(FeKey) + in.n_text + in.n_rec * 4);
    lap("upload");
    DevInput di{b_keys.as<FeKey>(), (uint32_t)in.n_keys, b_text.as<char>(), b_rec.as<uint32_t>(), (uint32_t)in.n_rec, (uint32_t)in.n_slots,
                b_ctl.as<FeCtl>()};
    di.host_locus = &L;
    di.want_interdist = in.want_interdist;
    di.h_hist = &in.interdist_hist;
    return front_stages(*Fp, di, o, st, out, declined);
}

// The record route: the text (SAM) or the inflated stream (BAM) is already on its way to `d_text`; the line table follows, and
// fields, filters and key grouping run as kernels before the stages above.
typedef hgx_front_totals ManyTotals;       // what a many-task pass reports per task (host side)

// the line table of a stream as (offset, length, task), in pinned staging (made by the host's workers)
LineRef *line_refs(const char *raw, const hgx_line *lines, size_t n_lines, bool binary, uint32_t base, uint32_t task, LineRef *dst) {
    const size_t skip = binary ? 32 : 0;
    hgx_par_ranges(n_lines > 50000 ? hgx_default_threads() : 1, n_lines, [&](int, size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            dst[i].off = base + (uint32_t)((size_t)(lines[i].p - raw) - skip);
            dst[i].len = lines[i].len;
            dst[i].task = task;
        }
    });
    return dst;
}

// the line table of unwalked BAM streams (resident at d_text: one per task, at `bases`), made on the device: (offset, length, task) of
// the records the tasks' regions keep, task after task, in QNAME order inside a task (stable: file order among equal names) -- what
// hgx_bam.cpp's walk + filter + sort_lines give per file
int bam_lines_dev(const char *d_text, const std::vector<const hgx_bam_deferred *> &defs, const std::vector<size_t> &bases, const std::vector<size_t> &sizes,
                  hipStream_t st, DevBuf &b_lines, uint32_t *n_lines, int *declined) {
    *declined = 0;
    *n_lines = 0;
    Lap lap(st);
    const unsigned char *text = (const unsigned char *)d_text;
    const int n_seg = (int)defs.size();
    if (n_seg < 1 || n_seg > 65535) { *declined = HGX_FE_DECLINE_SIZE; return HGX_OK; }
    std::vector<BamSeg> segs((size_t)n_seg);
    std::vector<uint8_t> acts;
    size_t total_body = 0;
    for (int t = 0; t < n_seg; ++t) {
        if (defs[t]->body0 > sizes[t] || bases[t] + sizes[t] >= (1ull << 32) - 64) { *declined = HGX_FE_DECLINE_SIZE; return HGX_OK; }
        total_body += sizes[t] - defs[t]->body0;
    }
    // ranges: ~16 KB of records each, at most 8192 for a lone stream, shared out by size among many
    // Ranges of ~8 KB (about twenty records).  Measured at 1 M records (tools/walk_ranges.sh): the first walk takes 0.41 / 0.34 /
    // 0.32 / 0.32 / 0.35 ms with ranges of 48 / 16 / 8 / 4 / 2 KB -- most of it the scan for each range's first record, the walk
    // itself (the second pass: 0.04 ms) is a short chain of dependent loads per range.
    const size_t W_all = std::max<size_t>((size_t)n_seg, std::min<size_t>(((size_t)1 << 20) + (size_t)n_seg, total_body / 8192 + 1));
    uint32_t w_at = 0;
    for (int t = 0; t < n_seg; ++t) {
        const hgx_bam_deferred &d = *defs[t];
        BamSeg &G = segs[t];
        G.base = (uint32_t)bases[t]; G.n = (uint32_t)sizes[t]; G.body0 = (uint32_t)d.body0;
        G.n_ref = (int32_t)d.ref_action.size();
        G.act_off = (uint32_t)acts.size();
        acts.insert(acts.end(), d.ref_action.begin(), d.ref_action.end());
        G.filtered = d.filtered ? 1u : 0u; G.left0 = (long long)d.left0; G.right0 = (long long)d.right0;
        const size_t body = sizes[t] - d.body0;
        G.first_range = w_at;
        G.n_ranges = (uint32_t)std::max<size_t>(1, total_body ? (W_all * body + total_body - 1) / total_body : 1);
        w_at += G.n_ranges;
    }
    const int W = (int)w_at;
    DevBuf b_seg, b_rng, b_base, b_cnt, b_tmpw, b_ctl, b_act, b_off, b_len, b_task, b_keep, b_pos, b_idx, b_idx2, b_key, b_key2, b_tmp;
    struct Drain { hipStream_t s; ~Drain() { (void)hipStreamSynchronize(s); } } drain{st};
    ALLOC(b_seg, (size_t)n_seg * sizeof(BamSeg));
    ALLOC(b_rng, (size_t)W * sizeof(BamRange));
    ALLOC(b_base, (size_t)W * 4);
    ALLOC(b_cnt, (size_t)W * 4);
    ALLOC(b_ctl, sizeof(BamCtl));
    ALLOC(b_act, std::max<size_t>(acts.size(), 16));
    HIPCHK(hipMemsetAsync(b_ctl.p, 0, sizeof(BamCtl), st));
    HIPCHK(hipMemcpyAsync(b_seg.p, segs.data(), (size_t)n_seg * sizeof(BamSeg), hipMemcpyHostToDevice, st));
    if (!acts.empty()) HIPCHK(hipMemcpyAsync(b_act.p, acts.data(), acts.size(), hipMemcpyHostToDevice, st));
    BamCtl *ctl = b_ctl.as<BamCtl>();
    const BamSeg *d_seg = b_seg.as<BamSeg>();
    k_bam_walk<0><<<nblk(W, 64), 64, 0, st>>>(text, d_seg, n_seg, W, b_rng.as<BamRange>(), nullptr, nullptr, nullptr, nullptr);
    k_bam_link<<<nblk(W, 256), 256, 0, st>>>(b_rng.as<BamRange>(), d_seg, n_seg, W, b_cnt.as<uint32_t>(), ctl);
    {
        ALLOC(b_tmpw, fe_scan_scratch_bytes(W));
        HIPCHK(hipMemsetAsync(b_tmpw.p, 0, fe_scan_scratch_bytes(W), st));
        FeScanArgs sa{};
        sa.n_ch = 1;
        sa.ch[0] = FeScanCh{b_cnt.p, b_base.as<uint32_t>(), 0, FSC_U32};
        sa.totals = ctl->tot;                                                  // (the records of all ranges)
        const int rcs = fe_scan(sa, W, b_tmpw.p, st);
        if (rcs) return rcs;
    }
    BamCtl h;
    { const int rc_d = hgx_d2h(&h, ctl, sizeof(BamCtl), st); if (rc_d) { (void)hgx_sync(st); return rc_d; } }
    { const int rc_s = hgx_sync(st); if (rc_s) return rc_s; }
    lap("BAM record walk (ranges)");
    if (h.decline) { *declined = h.decline; return HGX_OK; }
    const uint32_t n_rec = h.tot[0];
    if (n_rec >= (1u << 30)) { *declined = HGX_FE_DECLINE_SIZE; return HGX_OK; }
    ALLOC(b_lines, std::max<size_t>(n_rec, 1) * sizeof(LineRef));
    if (n_rec == 0) return HGX_OK;
    ALLOC(b_off, (size_t)n_rec * 4); ALLOC(b_len, (size_t)n_rec * 4); ALLOC(b_task, (size_t)n_rec * 2 + 16); ALLOC(b_keep, (size_t)n_rec * 4); ALLOC(b_pos, (size_t)n_rec * 4);
    ALLOC(b_idx, (size_t)n_rec * 4); ALLOC(b_idx2, (size_t)n_rec * 4); ALLOC(b_key, (size_t)n_rec * 8); ALLOC(b_key2, (size_t)n_rec * 8);
    size_t tb2 = 0;
    (void)hipcub::DeviceRadixSort::SortPairs((void *)nullptr, tb2, (unsigned long long *)nullptr, (unsigned long long *)nullptr, (uint32_t *)nullptr,
                                             (uint32_t *)nullptr, (int)n_rec, 0, 64, st);
    const size_t tmp_bytes = std::max(tb2, fe_scan_scratch_bytes(n_rec));
    ALLOC(b_tmp, std::max<size_t>(tmp_bytes, 256));
    k_bam_walk<1><<<nblk(W, 64), 64, 0, st>>>(text, d_seg, n_seg, W, b_rng.as<BamRange>(), b_base.as<uint32_t>(), b_off.as<uint32_t>(), b_len.as<uint32_t>(),
                                             b_task.as<uint16_t>());
    k_bam_filter<<<nblk(n_rec, 256), 256, 0, st>>>(text, b_off.as<uint32_t>(), b_len.as<uint32_t>(), b_task.as<uint16_t>(), n_rec, d_seg, b_act.as<uint8_t>(),
                                                   b_keep.as<uint32_t>(), ctl);
    {
        HIPCHK(hipMemsetAsync(b_tmp.p, 0, fe_scan_scratch_bytes(n_rec), st));
        FeScanArgs sa{};
        sa.n_ch = 1;
        sa.ch[0] = FeScanCh{b_keep.p, b_pos.as<uint32_t>(), 0, FSC_U32};
        const int rcs = fe_scan(sa, (long)n_rec, b_tmp.p, st);
        if (rcs) return rcs;
    }
    k_bam_compact<<<nblk(n_rec, 256), 256, 0, st>>>(b_keep.as<uint32_t>(), b_pos.as<uint32_t>(), n_rec, b_idx.as<uint32_t>(), ctl);
    { const int rc_d = hgx_d2h(&h, ctl, sizeof(BamCtl), st); if (rc_d) { (void)hgx_sync(st); return rc_d; } }
    { const int rc_s = hgx_sync(st); if (rc_s) return rc_s; }
    lap("BAM records + region filter");
    if (h.decline) { *declined = h.decline; return HGX_OK; }
    const uint32_t n_kept = h.n_kept;
    uint32_t *idx = b_idx.as<uint32_t>(), *idx_alt = b_idx2.as<uint32_t>();
    if (n_kept > 1) {
        // an aligner writes its records grouped by read already: a stable sort would not move anything
        DevBuf b_diff;
        ALLOC(b_diff, sizeof(NameDiff));
        HIPCHK(hipMemsetAsync(b_diff.p, 0, sizeof(NameDiff), st));
        // names of one 8-byte chunk are sorted as they are (one sort either way); longer ones on their varying bits
        const int n_chunks = (int)(h.max_klen + 7) / 8;
        const bool packed = n_chunks >= 2 && n_chunks <= NAME_DIFF_CHUNKS && !hgx_switch_has("front", "name_chunks");
        k_bam_sorted<<<nblk(n_kept, 256), 256, 0, st>>>(text, b_off.as<uint32_t>(), b_task.as<uint16_t>(), idx, n_kept, ctl,
                                                        packed ? b_diff.as<unsigned long long>() : (unsigned long long *)nullptr);
        NameDiff nd;
        { const int rc_d = hgx_d2h(&h, ctl, sizeof(BamCtl), st); if (rc_d) { (void)hgx_sync(st); return rc_d; } }
        { const int rc_d = hgx_d2h(&nd, b_diff.p, sizeof(NameDiff), st); if (rc_d) { (void)hgx_sync(st); return rc_d; } }
        { const int rc_s = hgx_sync(st); if (rc_s) return rc_s; }
        if (h.unsorted) {
            unsigned long long *key = b_key.as<unsigned long long>(), *key_alt = b_key2.as<unsigned long long>();
            int n_bits = 0;
            for (int c = 0; c < std::min(n_chunks, NAME_DIFF_CHUNKS); ++c) n_bits += __builtin_popcountll(nd.m[c]);
            if (packed) {
                // the varying bits alone, 64 to a word, least significant word first; every pass stable
                for (int word = 0; word < (n_bits + 63) / 64; ++word) {
                    k_bam_name_key_packed<<<nblk(n_kept, 256), 256, 0, st>>>(text, b_off.as<uint32_t>(), idx, n_kept, nd, n_chunks, n_bits, word, key);
                    size_t b = tmp_bytes;
                    HIPCHK(hipcub::DeviceRadixSort::SortPairs(b_tmp.p, b, key, key_alt, idx, idx_alt, (int)n_kept, 0, std::min(64, n_bits - 64 * word), st));
                    std::swap(idx, idx_alt);
                }
            } else
            for (int chunk = n_chunks - 1; chunk >= 0; --chunk) {      // (names beyond 256 bytes; test switch front=name_chunks) eight bytes at a time, least significant first
                k_bam_name_key<<<nblk(n_kept, 256), 256, 0, st>>>(text, b_off.as<uint32_t>(), idx, n_kept, (uint32_t)chunk, key);
                size_t b = tmp_bytes;
                HIPCHK(hipcub::DeviceRadixSort::SortPairs(b_tmp.p, b, key, key_alt, idx, idx_alt, (int)n_kept, 0, 64, st));
                std::swap(idx, idx_alt);
            }
            if (n_seg > 1) {                                                           // ... and the tasks apart again, names in order inside
                k_bam_task_key<<<nblk(n_kept, 256), 256, 0, st>>>(b_task.as<uint16_t>(), idx, n_kept, key);
                size_t b = tmp_bytes;
                HIPCHK(hipcub::DeviceRadixSort::SortPairs(b_tmp.p, b, key, key_alt, idx, idx_alt, (int)n_kept, 0, 16, st));
                std::swap(idx, idx_alt);
            }
        }
        lap(h.unsorted ? "BAM name sort" : "BAM name order check");
    }
    if (n_kept) k_bam_lines<<<nblk(n_kept, 256), 256, 0, st>>>(b_off.as<uint32_t>(), b_len.as<uint32_t>(), b_task.as<uint16_t>(), idx, n_kept, b_lines.as<LineRef>());
    HIPCHK(hipGetLastError());
    *n_lines = n_kept;
    return HGX_OK;
}

// ---- the line table of SAM TEXT on the device (round 5) --------------------------------------------------------------
// What hgx_bam.cpp's readers do per line on the host's threads (0.2-0.3 CPU-seconds per 1 M-read call: memchr for the newlines,
// six fields of every line for the region test, the QNAME order check): here the text that went up is scanned where it lies.
//   k_sam_nl<0/1>    newlines per 4 KB tile (a wavefront reads 1 KB per step, 16 bytes per lane, a SWAR zero-byte test per dword),
//                    a scan of the tile counts, then the same walk again writing every line's start
//   k_sam_line_info  a lane per line: '\r' stripped, blank and '@' lines dropped, the region test of `samtools view` on RNAME, POS and
//                    the CIGAR's reference span (one region at most: hgx_bam_deferred), QNAME length
//   scan + compact, k_sam_sorted (QNAME order check), LSD radix passes over 8-byte QNAME chunks where the text is not in name order
//   (k_sam_name_key + hipcub sort, as for BAM), k_sam_lines -> FeLine
struct SamRegion { int filtered, whole_len, name_len; long long left0, right0; char whole[96], name[96]; };
constexpr int SAM_TILE = 4096;
__device__ __forceinline__ uint32_t sam_nl_mask(uint32_t w) {          // bit 8 k + 7 set where byte k of w is '\n'
    const uint32_t x = w ^ 0x0A0A0A0Au;
    return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;     // exact zero-byte test (no borrow across bytes)
}
template <int PASS>
__global__ void __launch_bounds__(256) k_sam_nl(const unsigned char *__restrict__ text, size_t n, uint32_t n_tiles, uint32_t *__restrict__ cnt,
                                                const uint32_t *__restrict__ base, uint32_t *__restrict__ starts) {
    const uint32_t tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (tile >= n_tiles) return;
    const size_t t0 = (size_t)tile * SAM_TILE;
    uint32_t run = PASS == 1 ? base[tile] : 0u;
    for (int k = 0; k < SAM_TILE / 1024; ++k) {
        const size_t at = t0 + (size_t)k * 1024 + 16u * lane;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (at + 16 <= n) v = *reinterpret_cast<const uint4 *>(text + at);            // (the buffer is padded by 64 bytes; `n` cuts the count)
        else if (at < n) { unsigned char tmp[16] = {0}; for (size_t j = 0; at + j < n; ++j) tmp[j] = text[at + j]; __builtin_memcpy(&v, tmp, 16); }
        const uint32_t m0 = sam_nl_mask(v.x), m1 = sam_nl_mask(v.y), m2 = sam_nl_mask(v.z), m3 = sam_nl_mask(v.w);
        const uint32_t c = (uint32_t)(__popc(m0) + __popc(m1) + __popc(m2) + __popc(m3));
        if (PASS == 0) run += c;
        else {
            const uint32_t incl = wave_incl_scan_u32_front(c);
            uint32_t at_line = run + incl - c;
            const uint32_t ms[4] = {m0, m1, m2, m3};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t m = ms[q];
                while (m) {
                    const int b = __ffs((int)m) - 1;                          // bit 8 j + 7
                    m &= m - 1;
                    starts[at_line + 1] = (uint32_t)(at + 4u * q + (uint32_t)(b >> 3) + 1u);
                    ++at_line;
                }
            }
            run += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
    }
    if (PASS == 0) {
        const uint32_t tot = (uint32_t)wave_sum_u64(run);
        if (lane == 0) cnt[tile] = tot;
    }
}
__device__ __forceinline__ bool sam_bytes_eq(const unsigned char *a, const char *b, int n) {
    for (int i = 0; i < n; ++i) if (a[i] != (unsigned char)b[i]) return false;
    return true;
}
__global__ void __launch_bounds__(256) k_sam_line_info(const unsigned char *__restrict__ text, size_t n, const uint32_t *__restrict__ starts, uint32_t n_all,
                                                       SamRegion R, uint32_t *__restrict__ keep, uint32_t *__restrict__ l_len, uint32_t *__restrict__ l_klen,
                                                       BamCtl *ctl) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_all) return;
    const uint32_t off = starts[i];
    uint32_t end = i + 1 < n_all ? starts[i + 1] - 1u : (uint32_t)n;             // the newline's place (or the text's end: a last line without one)
    if (end < off) end = off;
    if (end > off && text[end - 1] == '\r') --end;
    const uint32_t len = end - off;
    uint32_t k = 0, klen = 0;
    if (len && text[off] != '@') {
        // QNAME and, for the region test, FLAG / RNAME / POS / CIGAR (hgx_bam.cpp take_line)
        const unsigned char *p = text + off;
        uint32_t tab[6];
        int nf = 0;
        const int want = R.filtered ? 6 : 1;
        // the first `want` tabs, eight bytes per look (an exact zero-byte test of word ^ tabs gives their places)
        for (uint32_t q = 0; q < len && nf < want;) {
            if (q + 8 <= len) {
                unsigned long long w;
                __builtin_memcpy(&w, p + q, 8);
                const unsigned long long x = w ^ 0x0909090909090909ULL;
                unsigned long long m = ~(((x & 0x7F7F7F7F7F7F7F7FULL) + 0x7F7F7F7F7F7F7F7FULL) | x) & 0x8080808080808080ULL;
                while (m && nf < want) { tab[nf++] = q + (uint32_t)(__builtin_ctzll(m) >> 3); m &= m - 1; }
                q += 8;
            } else {
                if (p[q] == '\t') tab[nf++] = q;
                ++q;
            }
        }
        klen = nf ? tab[0] : len;
        if (!R.filtered) k = 1;
        else if (nf == 6) {
            long long flag = 0, pos = 0;
            bool fneg = false, pneg = false;
            {   // strtol(.., 10): optional blanks and sign, digits (what follows is ignored)
                uint32_t q = tab[0] + 1;
                while (q < tab[1] && (p[q] == ' ' || (p[q] >= 9 && p[q] <= 13))) ++q;
                if (q < tab[1] && (p[q] == '+' || p[q] == '-')) { fneg = p[q] == '-'; ++q; }
                for (; q < tab[1] && p[q] >= '0' && p[q] <= '9'; ++q) if (flag < (1ll << 40)) flag = flag * 10 + (p[q] - '0');
                if (fneg) flag = -flag;
                q = tab[2] + 1;
                while (q < tab[3] && (p[q] == ' ' || (p[q] >= 9 && p[q] <= 13))) ++q;
                if (q < tab[3] && (p[q] == '+' || p[q] == '-')) { pneg = p[q] == '-'; ++q; }
                for (; q < tab[3] && p[q] >= '0' && p[q] <= '9'; ++q) if (pos < (1ll << 40)) pos = pos * 10 + (p[q] - '0');
                if (pneg) pos = -pos;
            }
            const long long pos0 = pos - 1;
            long long reflen = 0;
            if (!(flag & 4)) {
                long long num = 0;
                for (uint32_t q = tab[4] + 1; q < tab[5]; ++q) {
                    const unsigned char c = p[q];
                    if (c >= '0' && c <= '9') { num = num * 10 + (c - '0'); continue; }
                    if (c == 'M' || c == 'D' || c == 'N' || c == '=' || c == 'X') reflen += num;
                    num = 0;
                }
            }
            const long long end0 = pos0 + (reflen > 0 ? reflen : 1) - 1;
            const unsigned char *rn = p + tab[1] + 1;
            const int rl = (int)(tab[2] - tab[1] - 1);
            if (rl == R.whole_len && sam_bytes_eq(rn, R.whole, rl)) k = 1;
            else if (R.name_len > 0 && rl == R.name_len && sam_bytes_eq(rn, R.name, rl)) k = (end0 >= R.left0 && pos0 <= R.right0) ? 1u : 0u;
        }
    }
    keep[i] = k;
    l_len[i] = len;
    l_klen[i] = klen;
    if (k) atomicMax(&ctl->max_klen, klen);
}
__global__ void k_sam_compact(const uint32_t *__restrict__ keep, const uint32_t *__restrict__ pos, const uint32_t *__restrict__ starts, const uint32_t *__restrict__ l_len,
                              const uint32_t *__restrict__ l_klen, uint32_t n_all, uint32_t *__restrict__ k_off, uint32_t *__restrict__ k_len,
                              uint32_t *__restrict__ k_klen, uint32_t *__restrict__ idx, BamCtl *ctl) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_all) return;
    if (keep[i]) { const uint32_t q = pos[i]; k_off[q] = starts[i]; k_len[q] = l_len[i]; k_klen[q] = l_klen[i]; idx[q] = q; }
    if (i == n_all - 1) ctl->n_kept = pos[i] + keep[i];
}
__global__ void __launch_bounds__(256) k_sam_sorted(const unsigned char *__restrict__ text, const uint32_t *__restrict__ k_off, const uint32_t *__restrict__ k_klen,
                                                    uint32_t n, BamCtl *ctl, unsigned long long *__restrict__ diff) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i > 0 && i < n;
    const unsigned char *a = text + k_off[valid ? i - 1 : 0], *b = text + k_off[valid ? i : 0];
    const uint32_t la = k_klen[valid ? i - 1 : 0], lb = k_klen[valid ? i : 0];
    if (diff) name_diff_add(a, la, b, lb, valid, diff);
    if (valid && bam_name_cmp(b, lb, a, la) < 0) ctl->unsorted = 1;
}
__global__ void __launch_bounds__(256) k_sam_name_key_packed(const unsigned char *__restrict__ text, const uint32_t *__restrict__ k_off, const uint32_t *__restrict__ k_klen,
                                                             const uint32_t *__restrict__ idx, uint32_t n, NameDiff D, int n_chunks, int n_bits, int word,
                                                             unsigned long long *__restrict__ key) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = idx[i];
    key[i] = name_packed_word(text + k_off[r], k_klen[r], D, n_chunks, n_bits, word);
}
__global__ void __launch_bounds__(256) k_sam_name_key(const unsigned char *__restrict__ text, const uint32_t *__restrict__ k_off, const uint32_t *__restrict__ k_klen,
                                                      const uint32_t *__restrict__ idx, uint32_t n, uint32_t chunk, unsigned long long *__restrict__ key) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = idx[i], klen = k_klen[r];
    unsigned long long v = 0;
    if (8 * chunk < klen) {
        const uint32_t have = min(8u, klen - 8 * chunk);
        const unsigned char *q = text + k_off[r] + 8 * chunk;
        for (uint32_t j = 0; j < have; ++j) v |= (unsigned long long)q[j] << (8 * (7 - j));      // big endian: byte order = key order
    }
    key[i] = v;
}
__global__ void k_sam_lines(const uint32_t *__restrict__ k_off, const uint32_t *__restrict__ k_len, const uint32_t *__restrict__ idx, uint32_t n,
                            uint32_t off_base, FeLine *__restrict__ lines) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const uint32_t r = idx[i]; lines[i] = FeLine{off_base + k_off[r], k_len[r], 0u}; }
}

// `off_base`: d_text is a PART of a larger text that starts off_base bytes into it (the lines' offsets count from the whole text's
// first byte).  `unsorted_out` != NULL: a text that is not in name order is not sorted here -- *unsorted_out = 1, no lines.
int sam_lines_dev(const char *d_text, size_t n_bytes, const hgx_bam_deferred &def, hipStream_t st, DevBuf &b_lines, uint32_t *n_lines, int *declined,
                  uint32_t off_base = 0, int *unsorted_out = nullptr) {
    *declined = 0;
    *n_lines = 0;
    if (unsorted_out) *unsorted_out = 0;
    Lap lap(st);
    const unsigned char *text = (const unsigned char *)d_text;
    if (n_bytes >= (1ull << 32) - 64) { *declined = HGX_FE_DECLINE_SIZE; return HGX_OK; }
    SamRegion R;
    memset(&R, 0, sizeof(R));
    R.filtered = def.filtered ? 1 : 0;
    if (def.filtered) {
        if (def.region_whole.size() >= sizeof(R.whole) || def.region_name.size() >= sizeof(R.name)) { *declined = HGX_FE_DECLINE_SIZE; return HGX_OK; }
        R.whole_len = (int)def.region_whole.size(); memcpy(R.whole, def.region_whole.data(), def.region_whole.size());
        R.name_len = (int)def.region_name.size(); memcpy(R.name, def.region_name.data(), def.region_name.size());
        R.left0 = (long long)def.left0; R.right0 = (long long)def.right0;
    }
    const uint32_t n_tiles = (uint32_t)((n_bytes + SAM_TILE - 1) / SAM_TILE);
    DevBuf b_cnt, b_base, b_ctl, b_starts, b_keep, b_pos, b_len, b_klen, b_koff, b_klen2, b_klen3, b_idx, b_idx2, b_key, b_key2, b_tmp;
    struct Drain { hipStream_t s; ~Drain() { (void)hipStreamSynchronize(s); } } drain{st};
    const size_t sc_t = fe_scan_scratch_bytes(std::max<uint32_t>(n_tiles, 1));
    ALLOC(b_cnt, std::max<size_t>(n_tiles, 1) * 4);
    ALLOC(b_base, std::max<size_t>(n_tiles, 1) * 4);
    ALLOC(b_ctl, 256 + sc_t);
    HIPCHK(hipMemsetAsync(b_ctl.p, 0, 256 + sc_t, st));
    BamCtl *ctl = b_ctl.as<BamCtl>();
    BamCtl h;
    memset(&h, 0, sizeof(h));
    if (n_tiles) {
        k_sam_nl<0><<<nblk(n_tiles, 4), 256, 0, st>>>(text, n_bytes, n_tiles, b_cnt.as<uint32_t>(), nullptr, nullptr);
        FeScanArgs sa{};
        sa.n_ch = 1;
        sa.ch[0] = FeScanCh{b_cnt.p, b_base.as<uint32_t>(), 0, FSC_U32};
        sa.totals = ctl->tot;
        const int rcs = fe_scan(sa, (long)n_tiles, (char *)b_ctl.p + 256, st);
        if (rcs) return rcs;
        { const int rc_d = hgx_d2h(&h, ctl, sizeof(BamCtl), st); if (rc_d) { (void)hgx_sync(st); return rc_d; } }
        { const int rc_s = hgx_sync(st); if (rc_s) return rc_s; }
    }
    // lines = newlines (+ a last line without one); one more entry than lines for "the next line's start"
    const uint32_t n_nl = h.tot[0];
    ALLOC(b_starts, ((size_t)n_nl + 2) * 4);
    HIPCHK(hipMemsetAsync(b_starts.p, 0, 4, st));                            // the first line starts at 0
    if (n_tiles) k_sam_nl<1><<<nblk(n_tiles, 4), 256, 0, st>>>(text, n_bytes, n_tiles, nullptr, b_base.as<uint32_t>(), b_starts.as<uint32_t>());
    // (a text that ends with its last newline has no line behind it; k_sam_line_info drops the empty one)
    const uint32_t n_all = n_nl + 1;
    lap("SAM newline scan");
    if (n_all >= (1u << 30)) { *declined = HGX_FE_DECLINE_SIZE; return HGX_OK; }
    ALLOC(b_keep, (size_t)n_all * 4); ALLOC(b_pos, (size_t)n_all * 4); ALLOC(b_len, (size_t)n_all * 4); ALLOC(b_klen, (size_t)n_all * 4);
    ALLOC(b_koff, (size_t)n_all * 4); ALLOC(b_klen2, (size_t)n_all * 4); ALLOC(b_klen3, (size_t)n_all * 4); ALLOC(b_idx, (size_t)n_all * 4);
    size_t tb2 = 0;
    (void)hipcub::DeviceRadixSort::SortPairs((void *)nullptr, tb2, (unsigned long long *)nullptr, (unsigned long long *)nullptr, (uint32_t *)nullptr,
                                             (uint32_t *)nullptr, (int)n_all, 0, 64, st);
    const size_t tmp_bytes = std::max(tb2, fe_scan_scratch_bytes(n_all));
    ALLOC(b_tmp, std::max<size_t>(tmp_bytes, 256));
    HIPCHK(hipMemsetAsync(b_tmp.p, 0, fe_scan_scratch_bytes(n_all), st));
    k_sam_line_info<<<nblk(n_all, 256), 256, 0, st>>>(text, n_bytes, b_starts.as<uint32_t>(), n_all, R, b_keep.as<uint32_t>(), b_len.as<uint32_t>(),
                                                      b_klen.as<uint32_t>(), ctl);
    {
        FeScanArgs sa{};
        sa.n_ch = 1;
        sa.ch[0] = FeScanCh{b_keep.p, b_pos.as<uint32_t>(), 0, FSC_U32};
        const int rcs = fe_scan(sa, (long)n_all, b_tmp.p, st);
        if (rcs) return rcs;
    }
    k_sam_compact<<<nblk(n_all, 256), 256, 0, st>>>(b_keep.as<uint32_t>(), b_pos.as<uint32_t>(), b_starts.as<uint32_t>(), b_len.as<uint32_t>(), b_klen.as<uint32_t>(),
                                                    n_all, b_koff.as<uint32_t>(), b_klen2.as<uint32_t>(), b_klen3.as<uint32_t>(), b_idx.as<uint32_t>(), ctl);
    { const int rc_d = hgx_d2h(&h, ctl, sizeof(BamCtl), st); if (rc_d) { (void)hgx_sync(st); return rc_d; } }
    { const int rc_s = hgx_sync(st); if (rc_s) return rc_s; }
    lap("SAM lines + region filter");
    const uint32_t n_kept = h.n_kept;
    // QNAME is at most 254 characters (SAM specification 1.4; a BAM's l_read_name caps it at 255 with the NUL): a longer "name" is a
    // malformed or tab-less line -- its length would become the number of 8-byte sort passes below (a 1 MB line: ~130 000 of them)
    // before the record stage declined it anyway.  The host stages take the call and word the error.
    if (h.max_klen > 254) { *declined = HGX_FE_DECLINE_RECORD; return HGX_OK; }
    ALLOC(b_lines, std::max<size_t>(n_kept, 1) * sizeof(LineRef));
    uint32_t *idx = b_idx.as<uint32_t>();
    if (n_kept > 1) {
        DevBuf b_diff;
        ALLOC(b_diff, sizeof(NameDiff));
        HIPCHK(hipMemsetAsync(b_diff.p, 0, sizeof(NameDiff), st));
        const int n_chunks = (int)(h.max_klen + 7) / 8;
        const bool packed = n_chunks >= 2 && n_chunks <= NAME_DIFF_CHUNKS && !hgx_switch_has("front", "name_chunks");
        k_sam_sorted<<<nblk(n_kept, 256), 256, 0, st>>>(text, b_koff.as<uint32_t>(), b_klen3.as<uint32_t>(), n_kept, ctl,
                                                        packed ? b_diff.as<unsigned long long>() : (unsigned long long *)nullptr);
        NameDiff nd;
        { const int rc_d = hgx_d2h(&h, ctl, sizeof(BamCtl), st); if (rc_d) { (void)hgx_sync(st); return rc_d; } }
        { const int rc_d = hgx_d2h(&nd, b_diff.p, sizeof(NameDiff), st); if (rc_d) { (void)hgx_sync(st); return rc_d; } }
        { const int rc_s = hgx_sync(st); if (rc_s) return rc_s; }
        if (h.unsorted && unsorted_out) { *unsorted_out = 1; return HGX_OK; }
        if (h.unsorted) {
            ALLOC(b_idx2, (size_t)n_kept * 4); ALLOC(b_key, (size_t)n_kept * 8); ALLOC(b_key2, (size_t)n_kept * 8);
            uint32_t *idx_alt = b_idx2.as<uint32_t>();
            unsigned long long *key = b_key.as<unsigned long long>(), *key_alt = b_key2.as<unsigned long long>();
            int n_bits = 0;
            for (int c = 0; c < std::min(n_chunks, NAME_DIFF_CHUNKS); ++c) n_bits += __builtin_popcountll(nd.m[c]);
            if (packed) {
                for (int word = 0; word < (n_bits + 63) / 64; ++word) {                // the varying bits alone (see k_bam_sorted)
                    k_sam_name_key_packed<<<nblk(n_kept, 256), 256, 0, st>>>(text, b_koff.as<uint32_t>(), b_klen3.as<uint32_t>(), idx, n_kept, nd, n_chunks, n_bits,
                                                                               word, key);
                    size_t b = tmp_bytes;
                    HIPCHK(hipcub::DeviceRadixSort::SortPairs(b_tmp.p, b, key, key_alt, idx, idx_alt, (int)n_kept, 0, std::min(64, n_bits - 64 * word), st));
                    std::swap(idx, idx_alt);
                }
            } else
            for (int chunk = n_chunks - 1; chunk >= 0; --chunk) {      // (long names; front=name_chunks) least significant eight bytes first; every pass stable
                k_sam_name_key<<<nblk(n_kept, 256), 256, 0, st>>>(text, b_koff.as<uint32_t>(), b_klen3.as<uint32_t>(), idx, n_kept, (uint32_t)chunk, key);
                size_t b = tmp_bytes;
                HIPCHK(hipcub::DeviceRadixSort::SortPairs(b_tmp.p, b, key, key_alt, idx, idx_alt, (int)n_kept, 0, 64, st));
                std::swap(idx, idx_alt);
            }
        }
        lap(h.unsorted ? "SAM name sort" : "SAM name order check");
    }
    if (n_kept) k_sam_lines<<<nblk(n_kept, 256), 256, 0, st>>>(b_koff.as<uint32_t>(), b_klen2.as<uint32_t>(), idx, n_kept, off_base, b_lines.as<LineRef>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));                                           // (idx may live in a buffer of this function)
    *n_lines = n_kept;
    return HGX_OK;
}

int records_run(hgx_locus &L, const char *d_text, size_t raw_bytes, const LineRef *h_lines, size_t n_lines, bool binary, int n_tasks,
                const hgx_parse_opts &o, hipStream_t st, hgx_dbatch **out, ManyTotals *many, int *declined, const LineRef *d_lines = nullptr,
                const FeRec *pre_recs = nullptr) {
    // (pre_recs: the records of d_lines taken apart already -- k_fe_records ran beside the upload's tail, records_split below)
    *out = nullptr;
    *declined = 0;
    Lap lap(st);
    const FeLocus *Fp = nullptr;
    bool usable = false;
    int rc = dev_locus(L, st, &Fp, &usable);
    if (rc) return rc;
    if (!usable) { *declined = HGX_FE_DECLINE_LOCUS; return HGX_OK; }
    if (raw_bytes >= (1ull << 32) - 64 || n_lines >= (1ull << 30)) { *declined = HGX_FE_DECLINE_SIZE; return HGX_OK; }
    const uint32_t n = (uint32_t)n_lines;
    uint32_t cap = 1024;
    while (cap < 2 * (uint64_t)n) cap <<= 1;
    DevBuf b_lines, b_recs, b_head, b_kept, b_slot_of, b_tkeys, b_rep, b_pile, b_anyk, b_dslot, b_is_key, b_is_dec, b_kept32, b_prev_in;
    DevBuf b_key_idx, b_dec_idx, b_rec_idx, b_prev, b_tmp, b_keys, b_rec, b_ctl, b_treads, b_tpairs, b_trefs, b_tpieces;
    DevBuf b_ihist, b_iflag, b_iidx, b_icomp;
    const bool want_interdist = o.codis_choose_pairs || o.interdist_exchange;
    if (want_interdist && n_tasks > 1) { *declined = HGX_FE_DECLINE_OPTS; return HGX_OK; }
    struct Drain { hipStream_t s; ~Drain() { (void)hipStreamSynchronize(s); } } drain{st};
    if (!d_lines) ALLOC(b_lines, std::max<size_t>(n_lines, 1) * sizeof(LineRef));
    if (!pre_recs) ALLOC(b_recs, std::max<size_t>(n_lines, 1) * sizeof(FeRec));
    ALLOC(b_head, std::max<size_t>(n_lines, 16));
    ALLOC(b_kept, std::max<size_t>(n_lines, 16));
    ALLOC(b_slot_of, std::max<size_t>(n_lines, 4) * 4);
    // (one block of ones -- key table, representatives -- and one of zeros -- pile / any-kept counters, the control block, the scan's
    // tile states: two memsets instead of six)
    const size_t sc_bytes = fe_scan_scratch_bytes(std::max<uint32_t>(n, 1));
    const size_t ctl_at = ((size_t)cap * 8 + 255) & ~(size_t)255, scan_at = ctl_at + 256;
    ALLOC(b_tkeys, (size_t)cap * 12);                              // tkeys [cap] u64 | rep [cap] u32
    ALLOC(b_pile, scan_at + sc_bytes);                             // pile [cap] u32 | anyk [cap] u32 | FeCtl | scan state
    ALLOC(b_dslot, (size_t)cap * 4);
    ALLOC(b_is_key, std::max<size_t>(n_lines, 4) * 4);
    ALLOC(b_is_dec, std::max<size_t>(n_lines, 4) * 4);
    ALLOC(b_kept32, std::max<size_t>(n_lines, 4) * 4);
    ALLOC(b_prev_in, std::max<size_t>(n_lines, 4) * 4);
    ALLOC(b_key_idx, std::max<size_t>(n_lines, 4) * 4);
    ALLOC(b_dec_idx, std::max<size_t>(n_lines, 4) * 4);
    ALLOC(b_rec_idx, std::max<size_t>(n_lines, 4) * 4);
    ALLOC(b_prev, std::max<size_t>(n_lines, 4) * 4);
    static_assert(sizeof(FeCtl) <= 256, "FeCtl outgrew its slot");
    HIPCHK(hipMemsetAsync(b_tkeys.p, 0xFF, (size_t)cap * 12, st));
    HIPCHK(hipMemsetAsync(b_pile.p, 0, scan_at + sc_bytes, st));
    uint32_t *const d_rep = (uint32_t *)((char *)b_tkeys.p + (size_t)cap * 8);
    uint32_t *const d_pile = b_pile.as<uint32_t>(), *const d_anyk = b_pile.as<uint32_t>() + cap;
    if (n && !d_lines) {
        HIPCHK(hipMemcpyAsync(b_lines.p, h_lines, (size_t)n * sizeof(LineRef), hipMemcpyHostToDevice, st));
        g_last_bytes += (long long)((size_t)n * sizeof(LineRef));
    }
    const LineRef *lines_dev = d_lines ? d_lines : b_lines.as<LineRef>();
    lap("text + line table on the device");
    FeCtl *ctl = (FeCtl *)((char *)b_pile.p + ctl_at);
    FeCtl h;
    memset(&h, 0, sizeof(h));
    if (n) {
        const FeFilter flt{o.num_editdist, o.allow_discordant, o.base_locus};
        FeRec *recs = pre_recs ? const_cast<FeRec *>(pre_recs) : b_recs.as<FeRec>();
        if (!pre_recs) k_fe_records<<<nblk(n, 256), 256, 0, st>>>(d_text, raw_bytes + 64, lines_dev, n, binary ? 1 : 0, o.simulation, recs, ctl);
        k_fe_rec_heads<<<nblk(n, 256), 256, 0, st>>>(recs, n, d_text, b_head.as<uint8_t>());
        k_fe_rec_filter_insert<<<nblk(n, 256), 256, 0, st>>>(recs, b_head.as<uint8_t>(), n, flt, b_tkeys.as<unsigned long long>(), d_rep,
                                                            d_pile, d_anyk, cap - 1, b_kept.as<uint8_t>(),
                                                            b_slot_of.as<uint32_t>(), ctl);
        k_fe_group_flags<<<nblk(n, 256), 256, 0, st>>>(recs, n, d_text, d_rep, d_pile, d_anyk,
                                                      b_slot_of.as<uint32_t>(), b_kept.as<uint8_t>(), b_is_key.as<uint32_t>(), b_is_dec.as<uint32_t>(),
                                                      b_kept32.as<uint32_t>(), b_prev_in.as<uint32_t>(), ctl);
        {   // key numbers, decode slots, kept-record numbers (sums) and "the kept record before me" (a running maximum): one pass
            FeScanArgs sa{};
            sa.n_ch = 4;
            sa.ch[0] = FeScanCh{b_is_key.p, b_key_idx.as<uint32_t>(), 0, FSC_U32};
            sa.ch[1] = FeScanCh{b_is_dec.p, b_dec_idx.as<uint32_t>(), 0, FSC_U32};
            sa.ch[2] = FeScanCh{b_kept32.p, b_rec_idx.as<uint32_t>(), 0, FSC_U32};
            sa.ch[3] = FeScanCh{b_prev_in.p, b_prev.as<uint32_t>(), 1, FSC_U32};
            rc = fe_scan(sa, (long)n, (char *)b_pile.p + scan_at, st);
            if (rc) return rc;
        }
        // (the key table is sized by the records: a key per record at most)
        ALLOC(b_keys, (size_t)n * sizeof(FeKey));
        ALLOC(b_rec, (size_t)n * 4);
        k_fe_build_keys<<<nblk(n, 256), 256, 0, st>>>(recs, n, b_is_key.as<uint32_t>(), b_is_dec.as<uint32_t>(), b_key_idx.as<uint32_t>(),
                                                     b_dec_idx.as<uint32_t>(), b_kept32.as<uint32_t>(), b_rec_idx.as<uint32_t>(), b_slot_of.as<uint32_t>(),
                                                     d_pile, o.base_locus, b_keys.as<FeKey>(), b_dslot.as<uint32_t>(), ctl);
        k_fe_build_recinfo<<<nblk(n, 256), 256, 0, st>>>(recs, n, d_text, b_kept.as<uint8_t>(), b_rec_idx.as<uint32_t>(), b_prev.as<uint32_t>(),
                                                        b_slot_of.as<uint32_t>(), b_dslot.as<uint32_t>(), b_rec.as<uint32_t>());
        if (want_interdist) {
            // CODIS D18S51: the inner distances of this stream's unique concordant pairs, as a histogram (front_stages takes the median)
            const size_t hist_bytes = ((size_t)HGX_INTERDIST_BINS * 4 + 255) & ~(size_t)255;      // hist | the scan's total | scan state
            ALLOC(b_ihist, hist_bytes + 256 + sc_bytes);
            ALLOC(b_iflag, (size_t)n * 4); ALLOC(b_iidx, (size_t)n * 4); ALLOC(b_icomp, (size_t)n * 4);
            HIPCHK(hipMemsetAsync(b_ihist.p, 0, hist_bytes + 256 + sc_bytes, st));
            uint32_t *const d_m = (uint32_t *)((char *)b_ihist.p + hist_bytes);
            k_fe_interdist_flag<<<nblk(n, 256), 256, 0, st>>>(recs, n, b_iflag.as<uint32_t>());
            FeScanArgs sa{};
            sa.n_ch = 1;
            sa.ch[0] = FeScanCh{b_iflag.p, b_iidx.as<uint32_t>(), 0, FSC_U32};
            sa.totals = d_m;
            rc = fe_scan(sa, (long)n, (char *)b_ihist.p + hist_bytes + 256, st);
            if (rc) return rc;
            k_fe_interdist_compact<<<nblk(n, 256), 256, 0, st>>>(b_iflag.as<uint32_t>(), b_iidx.as<uint32_t>(), n, b_icomp.as<uint32_t>());
            k_fe_interdist_hist<<<nblk(n, 256), 256, 0, st>>>(recs, d_text, b_icomp.as<uint32_t>(), d_m, b_ihist.as<uint32_t>(), ctl);
        }
        { const int rc_d = hgx_d2h(&h, ctl, sizeof(FeCtl), st); if (rc_d) { (void)hgx_sync(st); return rc_d; } }
        { const int rc_s = hgx_sync(st); if (rc_s) return rc_s; }
    } else {
        ALLOC(b_keys, sizeof(FeKey));
        ALLOC(b_rec, 16);
    }
    lap("records: fields, filters, keys");
    if (h.decline) { *declined = -h.decline; return HGX_OK; }
    DevInput di{b_keys.as<FeKey>(), h.n_keys, d_text, b_rec.as<uint32_t>(), h.n_rec, h.n_slots, ctl};
    di.n_tasks = std::max(1, n_tasks);
    di.host_locus = &L;
    di.want_interdist = want_interdist;
    di.d_hist = want_interdist && n ? b_ihist.as<uint32_t>() : nullptr;
    static const std::vector<int64_t> no_distances((size_t)HGX_INTERDIST_BINS, 0);
    di.h_hist = &no_distances;                                          // (an empty stream still takes part in the exchange)
    if (di.n_tasks > 1) {
        ALLOC(b_treads, (size_t)di.n_tasks * 4); ALLOC(b_tpairs, (size_t)di.n_tasks * 4); ALLOC(b_trefs, (size_t)di.n_tasks * 8);
        ALLOC(b_tpieces, (size_t)di.n_tasks * 4);
        HIPCHK(hipMemsetAsync(b_tpieces.p, 0, (size_t)di.n_tasks * 4, st));
        di.task_pieces = b_tpieces.as<uint32_t>();
        HIPCHK(hipMemsetAsync(b_treads.p, 0, (size_t)di.n_tasks * 4, st));
        HIPCHK(hipMemsetAsync(b_tpairs.p, 0, (size_t)di.n_tasks * 4, st));
        HIPCHK(hipMemsetAsync(b_trefs.p, 0, (size_t)di.n_tasks * 8, st));
        di.task_reads = b_treads.as<uint32_t>(); di.task_pairs = b_tpairs.as<uint32_t>(); di.task_refs = b_trefs.as<unsigned long long>();
    }
    rc = front_stages(*Fp, di, o, st, out, declined);
    if (rc || *declined || !many) return rc;
    many->reads.assign((size_t)di.n_tasks, 0); many->pairs.assign((size_t)di.n_tasks, 0); many->refs.assign((size_t)di.n_tasks, 0);
    many->pieces.assign((size_t)di.n_tasks, 0);
    if (di.n_tasks > 1) {
        HIPCHK(hipMemcpyAsync(many->pieces.data(), b_tpieces.p, (size_t)di.n_tasks * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(many->reads.data(), b_treads.p, (size_t)di.n_tasks * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(many->pairs.data(), b_tpairs.p, (size_t)di.n_tasks * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(many->refs.data(), b_trefs.p, (size_t)di.n_tasks * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    } else if (*out) {
        many->reads[0] = (uint32_t)(*out)->n_reads; many->pairs[0] = (uint32_t)(*out)->n_pairs; many->refs[0] = (uint64_t)(*out)->n_refs;
        many->pieces[0] = (uint32_t)(*out)->n_pieces;
    }
    return HGX_OK;
}

// SAM text whose last bytes are still on their way up (round 6).  The reader hands the text over in phases; the link carries it at
// ~47 GB/s (8.7 ms for the 410 MB of a 1 M-read file) and every kernel used to wait for the last byte.  Here the text is cut at a line
// end inside the phases that have LANDED but for the last one: part A's line table (k_sam_*) and record fields (k_fe_records, the
// largest kernel of a SAM call) run on a second stream beside the last phase's transfer, part B's behind it; the two line tables and
// record arrays are joined (the name order across the cut checked on the host's copy of the text) and the record stage goes on from
// the filters.  Anything unusual -- a part that is not in name order, a declined part -- returns *handled = 0: the whole text then goes
// the ordinary way.
struct SamPhase { size_t end; hipEvent_t landed; };
// the second stream of such a call: made once per device and caller in flight, then reused (hipStreamCreate takes 3-9 ms on this
// runtime -- more than the overlap wins -- so a call never makes one it could borrow)
struct SideStreams {
    std::mutex mu;
    std::vector<std::pair<int, hipStream_t>> idle;      // (device, stream)
    hipStream_t take(int dev) {
        {
            std::lock_guard<std::mutex> g(mu);
            for (size_t k = 0; k < idle.size(); ++k)
                if (idle[k].first == dev) { hipStream_t s = idle[k].second; idle.erase(idle.begin() + (long)k); return s; }
        }
        hipStream_t s = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
        return s;
    }
    void give(int dev, hipStream_t s) { std::lock_guard<std::mutex> g(mu); idle.push_back({dev, s}); }
};
SideStreams g_side;
int records_split(hgx_locus &L, const char *d_text, const char *raw, size_t raw_bytes, const hgx_bam_deferred &def, const std::vector<SamPhase> &phases,
                  const hgx_parse_opts &o, hipStream_t st, hgx_dbatch **out, int *declined, int *handled) {
    *handled = 0;
    if (phases.size() < 2 || phases.back().end != raw_bytes || raw_bytes >= (1ull << 32) - 64) return HGX_OK;
    // The parts: everything before the last phase (whole lines), with many phases cut once more so that the part that waits for the
    // last byte is small -- [0, end of phase n-3) behind phase n-3's event, [.., end of phase n-2) behind phase n-2's, the rest behind
    // the last copy.  The early parts run on a second stream in turn; each has finished before the next one's bytes are up.
    struct Part { size_t b0, b1; hipEvent_t after; DevBuf lines, recs; uint32_t n = 0; };
    std::vector<size_t> ends;
    if (phases.size() >= 6) ends.push_back(phases.size() - 3);
    ends.push_back(phases.size() - 2);
    std::vector<Part> parts;
    parts.reserve(4);                                               // (a Part holds buffers: the vector must not move them once they are made)
    size_t at = 0;
    for (size_t e : ends) {
        size_t cut = phases[e].end;
        while (cut > at && raw[cut - 1] != '\n') --cut;             // whole lines
        if (cut <= at) continue;
        parts.emplace_back();
        parts.back().b0 = at; parts.back().b1 = cut; parts.back().after = phases[e].landed;
        at = cut;
    }
    if (parts.empty() || at < raw_bytes / 4 || at >= raw_bytes) return HGX_OK;
    parts.emplace_back();
    parts.back().b0 = at; parts.back().b1 = raw_bytes; parts.back().after = nullptr;
    const size_t P = parts.size();
    Lap lap(st);
    const bool dbg = getenv("HGX_SPLIT_DEBUG") != nullptr;
    const double t_in = now_ms();
    auto mark = [&](const char *what, size_t k) { if (dbg) fprintf(stderr, "[records_split] part %zu of %zu: %-24s +%.3f ms\n", k + 1, P, what, now_ms() - t_in); };
    int dev = -1;
    HIPCHK(hipGetDevice(&dev));
    hipStream_t sb = g_side.take(dev);
    if (!sb) return HGX_OK;
    DevBuf b_lines, b_recs, b_pctl;
    hipEvent_t early_done = nullptr;
    std::vector<FeLine> edge(2 * P);                                 // (first and last line of every part: read back through the staging, so they outlive the guard)
    struct Guard {
        int dev; hipStream_t &sb, st; hipEvent_t &ev;
        ~Guard() { (void)hipStreamSynchronize(sb); (void)hgx_sync(st); if (ev) (void)hipEventDestroy(ev); g_side.give(dev, sb); }
    } guard{dev, sb, st, early_done};
    ALLOC(b_pctl, 256);
    HIPCHK(hipMemsetAsync(b_pctl.p, 0, 256, sb));
    size_t n = 0;
    for (size_t k = 0; k < P; ++k) {
        Part &pt = parts[k];
        const bool last = k + 1 == P;
        hipStream_t s = last ? st : sb;                              // (st: behind the last phase's copy)
        if (!last) HIPCHK(hipStreamWaitEvent(sb, pt.after, 0));
        int dec = 0, unsorted = 0;
        const int rc = sam_lines_dev(d_text + pt.b0, pt.b1 - pt.b0, def, s, pt.lines, &pt.n, &dec, (uint32_t)pt.b0, &unsorted);
        mark("lines back", k);
        if (rc) return rc;
        if (dec || unsorted) return HGX_OK;
        n += pt.n;
        if (last || pt.n == 0) continue;
        ALLOC(pt.recs, (size_t)pt.n * sizeof(FeRec));
        // (text_bytes = the part's end: no load of a record's scan reaches into bytes that are still landing)
        k_fe_records<<<nblk(pt.n, 256), 256, 0, sb>>>(d_text, pt.b1, pt.lines.as<LineRef>(), pt.n, 0, o.simulation, pt.recs.as<FeRec>(), b_pctl.as<FeCtl>());
        HIPCHK(hipGetLastError());
        mark("record fields queued", k);
    }
    if (dbg) fprintf(stderr, "[records_split] %zu lines in %zu parts (%u in the last)\n", n, P, parts.back().n);
    if (n == 0 || n >= (1ull << 30) || parts[0].n == 0) return HGX_OK;
    HIPCHK(hipEventCreateWithFlags(&early_done, hipEventDisableTiming));
    HIPCHK(hipEventRecord(early_done, sb));
    // joined: line tables and records of the parts, in file order; the last part's fields straight into their place
    ALLOC(b_lines, n * sizeof(LineRef));
    ALLOC(b_recs, n * sizeof(FeRec));
    size_t base = 0;
    for (size_t k = 0; k < P; ++k) {
        Part &pt = parts[k];
        if (pt.n == 0) continue;
        { const int rc_d = hgx_d2h(&edge[2 * k], pt.lines.as<LineRef>(), sizeof(FeLine), st); if (rc_d) { (void)hgx_sync(st); return rc_d; } }
        { const int rc_d = hgx_d2h(&edge[2 * k + 1], pt.lines.as<LineRef>() + (pt.n - 1), sizeof(FeLine), st); if (rc_d) { (void)hgx_sync(st); return rc_d; } }
        HIPCHK(hipMemcpyAsync(b_lines.as<LineRef>() + base, pt.lines.p, (size_t)pt.n * sizeof(LineRef), hipMemcpyDeviceToDevice, st));
        if (k + 1 == P) {
            k_fe_records<<<nblk(pt.n, 256), 256, 0, st>>>(d_text, raw_bytes + 64, pt.lines.as<LineRef>(), pt.n, 0, o.simulation, b_recs.as<FeRec>() + base, b_pctl.as<FeCtl>());
            HIPCHK(hipGetLastError());
        }
        base += pt.n;
    }
    HIPCHK(hipStreamWaitEvent(st, early_done, 0));
    base = 0;
    for (size_t k = 0; k + 1 < P; ++k) {
        if (parts[k].n) HIPCHK(hipMemcpyAsync(b_recs.as<FeRec>() + base, parts[k].recs.p, (size_t)parts[k].n * sizeof(FeRec), hipMemcpyDeviceToDevice, st));
        base += parts[k].n;
    }
    FeCtl h;
    { const int rc_d = hgx_d2h(&h, b_pctl.p, sizeof(FeCtl), st); if (rc_d) { (void)hgx_sync(st); return rc_d; } }
    { const int rc_s = hgx_sync(st); if (rc_s) return rc_s; }
    lap("parts: lines + record fields, joined");
    // name order across the cuts: line_less of hgx_bam.cpp on the host's copy (the last name of a part against the first of the next)
    const FeLine *prev = nullptr;
    for (size_t k = 0; k < P; ++k) {
        if (parts[k].n == 0) continue;
        if (prev) {
            const FeLine &ea = *prev, &eb = edge[2 * k];
            const char *a = raw + ea.off, *b = raw + eb.off;
            const void *ta = memchr(a, '\t', ea.len), *tb = memchr(b, '\t', eb.len);
            const size_t la = ta ? (size_t)((const char *)ta - a) : ea.len, lb = tb ? (size_t)((const char *)tb - b) : eb.len;
            const int c = memcmp(b, a, std::min(la, lb));
            if (c < 0 || (c == 0 && lb < la)) return HGX_OK;        // the next part's first name sorts before this one's last: the whole text is sorted the ordinary way
        }
        prev = &edge[2 * k + 1];
    }
    *handled = 1;
    g_last_parts = (int)P;
    if (h.decline) { *declined = -h.decline; return HGX_OK; }
    return records_run(L, d_text, raw_bytes, nullptr, n, false, 1, o, st, out, nullptr, declined, b_lines.as<LineRef>(), b_recs.as<FeRec>());
}

// Size gates of the device front end, from tools/front_gate.py on an MI355X box (round 6; 7 000- and 500-allele loci alike): the
// device stages cost a flat 1.0-1.5 ms for SAM text and 2.7-3.0 ms for a BAM file (inflate, walk and sorts on the device) up to
// 40 000 records (0.8-1.0 and 2.6-2.9 ms since the round's last changes: profiles/r06_final_front_gate.txt), the host stages 2.3 us
// per record on one thread -- they cross at ~700-800 records of SAM text and ~2 000 BAM records.  (Rounds 3-5 had the gate at 20 000 records / 8 MB: a real-depth sample of one locus -- 1 500 to 10 000 reads,
// devel/hg_test4_realbasic/*.report:11 -- never reached the kernels and paid 4-9 ms of host decode instead of 1.5.)
constexpr size_t FE_MIN_RECORDS = 1000;              // records a key- / record-route call must hold
constexpr size_t FE_MIN_BYTES = 300u << 10;          // ... and bytes of SAM text / inflated BAM stream (below: not even uploaded)
constexpr size_t FE_MIN_DEFER_BYTES = 512u << 10;    // bytes from which the line table / the BAM walk + inflate are the device's too

// host stages with the device stages hooked in; *out is always a device batch on success
template <class Parse>
int parse_dev(hgx_dbatch **out, hipStream_t st, const hgx_parse_opts *opts_in, Parse parse) {
    *out = nullptr;
    // a sharded locus: whichever route finishes the parse, this rank takes part in ONE pileup exchange (hgx_pileup_share)
    hgx_pileup_share share;
    hgx_parse_opts opts_own = *opts_in;
    if (opts_in->pileup_exchange_dev && !opts_in->pileup_exchange) {
        hgx_set_error("pileup_exchange_dev needs the host form (pileup_exchange) beside it: the device route may decline before its exchange");
        return HGX_EINVAL;
    }
    if (opts_in->pileup_exchange) {
        share.orig = opts_in->pileup_exchange;
        share.orig_ctx = opts_in->pileup_ctx;
        opts_own.pileup_exchange = &hgx_pileup_share::trampoline;
        opts_own.pileup_ctx = &share;
    }
    hgx_interdist_share id_share;          // ... and in ONE exchange of the inter-distance histogram (CODIS D18S51)
    if (opts_in->interdist_exchange) {
        id_share.orig = opts_in->interdist_exchange;
        id_share.orig_ctx = opts_in->interdist_ctx;
        opts_own.interdist_exchange = &hgx_interdist_share::trampoline;
        opts_own.interdist_ctx = &id_share;
    }
    const hgx_parse_opts *opts = &opts_own;
    hgx_dbatch *made = nullptr;
    hgx_front_hook hook;
    hook.mem = hgx_front_alloc{pinned_alloc, pinned_release};
    const bool force = hgx_switch_has("front", "device");
    const bool host_only = hgx_switch_has("front", "host");
    const bool no_records = hgx_switch_has("front", "keys");
    hook.min_records = force ? 0 : FE_MIN_RECORDS;
    int route = 0;                         // 2 = the record route produced the batch, 1 = the key route, 0 = the host stages
    hook.run = [&](hgx_locus &L, const hgx_front_input &in, const hgx_parse_opts &o, int *declined) {
        if (!force && in.n_rec < FE_MIN_RECORDS) { *declined = HGX_FE_DECLINE_SMALL; return (int)HGX_OK; }     // a dozen launches cost more than a small host decode
        hgx_dbatch_destroy(made);
        made = nullptr;
        const int rc = front_run(L, in, o, st, &made, declined);
        if (!rc && !*declined && made) route = 1;
        return rc;
    };
    // record route: the reader's bytes go up as soon as they are complete (SAM text read / BAM stream inflated), while the host
    // still walks and name-sorts the records
    DevBuf b_text, b_comp;
    size_t text_cap = 0;                  // bytes b_text holds (a second read of a file that changed meanwhile may ask for more)
    auto text_room = [&](size_t need) -> bool {        // true = b_text can take `need` bytes (drained and re-allocated if it could not)
        if (b_text.p && need <= text_cap) return true;
        if (b_text.p) { (void)hipStreamSynchronize(st); hgx_pool_free(b_text.p); b_text.p = nullptr; text_cap = 0; }
        if (b_text.alloc(need)) return false;
        text_cap = need;
        return true;
    };
    const char *up_raw = nullptr;
    const unsigned char *comp_from = nullptr;
    size_t up_bytes = 0, comp_n = 0;
    bool up_failed = false;
    // the phases of a SAM text on their way up, each with an event behind its copy (records_split)
    std::vector<SamPhase> sam_phases;
    auto drop_phases = [&]() { for (SamPhase &p : sam_phases) (void)hipEventDestroy(p.landed); sam_phases.clear(); };
    struct PhaseGuard { std::function<void()> f; ~PhaseGuard() { f(); } } phase_guard{drop_phases};
    const bool split_ok = !hgx_switch_has("front", "sam_whole");
    if (!host_only && !no_records) {
        hook.on_raw = [&](const char *raw, size_t n_bytes, size_t begin, size_t end) {
            if (n_bytes >= (1ull << 32) - 64 || up_failed) return;
            if (!force && n_bytes < FE_MIN_BYTES) return;            // (the record stage will decline it as small: no upload for nothing)
            if (!text_room(n_bytes + 64)) { up_failed = true; return; }
            if (end > begin && hipMemcpyAsync((char *)b_text.p + begin, raw + begin, end - begin, hipMemcpyHostToDevice, st) != hipSuccess) { up_failed = true; return; }
            if (up_raw != raw || up_bytes != n_bytes || begin == 0) drop_phases();      // (a file read a second time: the events of the first read say nothing)
            if (split_ok) {
                hipEvent_t ev = nullptr;
                if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess && hipEventRecord(ev, st) == hipSuccess) sam_phases.push_back(SamPhase{end, ev});
                else { if (ev) (void)hipEventDestroy(ev); drop_phases(); }
            }
            up_raw = raw;
            up_bytes = n_bytes;
            g_last_bytes += (long long)(end - begin);
        };
        // a BAM's records are walked, filtered and name-sorted on the device too when the stream is big enough to be worth the launches
        // -- and its BGZF blocks inflated there (hgx_inflate.hip): the deflated file goes up instead of the inflated stream
        hook.defer_walk = true;
        hook.defer_text = !hgx_switch_has("front", "host_lines");          // SAM text: the line table as kernels too (round 5)
        hook.defer_min_bytes = force ? 0 : FE_MIN_DEFER_BYTES;
        if (!hgx_switch_has("front", "host_inflate")) {
            // the file's bytes start their way up before the host has looked at the container (0.4 ms of transfer for a 20 MB BAM,
            // under the hop through 5 000 block headers and the inflate of the BAM header)
            hook.comp_early = [&](const unsigned char *data, size_t n) {
                if (up_failed || b_comp.p || b_comp.alloc(n + 2048)) return;
                if (hipMemcpyAsync(b_comp.p, data, n, hipMemcpyHostToDevice, st) != hipSuccess ||
                    hipMemsetAsync((char *)b_comp.p + n, 0, 2048, st) != hipSuccess) { up_failed = true; return; }
                comp_from = data; comp_n = n;
            };
            hook.comp_sync = [&]() { if (comp_from) (void)hipStreamSynchronize(st); };
            hook.inflate_dev = [&](const unsigned char *data, size_t n, const std::vector<hgx_bgzf_block> &blocks, size_t total) -> int {
                if (up_failed || total >= (1ull << 32) - 64) return 1;
                if (!text_room(total + 64)) return 1;
                struct DrainC { hipStream_t s; ~DrainC() { (void)hipStreamSynchronize(s); } } drain_c{st};
                if (comp_from != data || comp_n != n) {
                    if (b_comp.p) { (void)hipStreamSynchronize(st); hgx_pool_free(b_comp.p); b_comp.p = nullptr; }
                    if (b_comp.alloc(n + 2048)) return 1;
                    if (hipMemcpyAsync(b_comp.p, data, n, hipMemcpyHostToDevice, st) != hipSuccess) return 1;
                    if (hipMemsetAsync((char *)b_comp.p + n, 0, 2048, st) != hipSuccess) return 1;
                }
                int bad = 0;
                void *stg = pinned_alloc(hgx_bgzf_inflate_staging_bytes(blocks.size()));         // (registered: the block table queues behind the file)
                struct Unstage { void *p; ~Unstage() { pinned_release(p); } } unstage{stg};
                if (hgx_bgzf_inflate_dev(b_comp.as<unsigned char>(), blocks.data(), blocks.size(), b_text.as<unsigned char>(), st, &bad, stg) != HGX_OK || bad) return 1;
                up_raw = nullptr;
                up_bytes = total;
                g_last_bytes += (long long)n;
                return 0;
            };
        }
        hook.records = [&](hgx_locus &L, const char *raw, size_t raw_bytes, const hgx_line *lines, size_t n, bool binary, const hgx_parse_opts &o,
                           int *declined, const hgx_bam_deferred *def) {
            if (!def && !force && (n < FE_MIN_RECORDS || raw_bytes < FE_MIN_BYTES)) { *declined = HGX_FE_DECLINE_SMALL; return (int)HGX_OK; }
            if (up_failed || raw != up_raw || raw_bytes != up_bytes || !b_text.p) { *declined = HGX_FE_DECLINE_SIZE; return (int)HGX_OK; }
            if (def) {
                DevBuf b_dl;
                uint32_t n_dl = 0;
                struct DrainL { hipStream_t s; ~DrainL() { (void)hipStreamSynchronize(s); } } drain_l{st};
                if (def->text && raw && sam_phases.size() >= 2) {
                    int handled = 0;
                    const int rcs = records_split(L, b_text.as<char>(), raw, raw_bytes, *def, sam_phases, o, st, &made, declined, &handled);
                    if (rcs) return rcs;
                    if (handled) {
                        if (!*declined && made) route = 2;
                        return (int)HGX_OK;
                    }
                    hgx_dbatch_destroy(made);
                    made = nullptr;
                    *declined = 0;
                }
                int rc = def->text ? sam_lines_dev(b_text.as<char>(), raw_bytes, *def, st, b_dl, &n_dl, declined)
                                   : bam_lines_dev(b_text.as<char>(), {def}, {(size_t)0}, {raw_bytes}, st, b_dl, &n_dl, declined);
                if (rc || *declined) return rc;
                rc = records_run(L, b_text.as<char>(), raw_bytes, nullptr, n_dl, !def->text, 1, o, st, &made, nullptr, declined, b_dl.as<LineRef>());
                if (!rc && !*declined && made) route = 2;
                return rc;
            }
            LineRef *h_lines = (LineRef *)pinned_alloc(std::max<size_t>(n, 1) * sizeof(LineRef));
            if (!h_lines) { hgx_set_error("pinned allocation of the line table failed"); return (int)HGX_ENOMEM; }
            struct Unpin { void *p; ~Unpin() { pinned_release(p); } } unpin{h_lines};
            line_refs(raw, lines, n, binary, 0u, 0u, h_lines);
            const int rc = records_run(L, b_text.as<char>(), raw_bytes, h_lines, n, binary, 1, o, st, &made, nullptr, declined);
            if (!rc && !*declined && made) route = 2;
            return rc;
        };
    }
    struct Drain { hipStream_t s; ~Drain() { (void)hipStreamSynchronize(s); } } drain{st};     // (an upload may still read the reader's buffer)
    hgx_batch *b = nullptr;
    g_last_bytes = 0;
    g_last_parts = 0;
    const int rc = parse(&b, host_only ? nullptr : &hook, opts);
    (void)hipStreamSynchronize(st);
    g_last_route = rc ? 0 : route;
    g_last_decline = host_only ? -1 : (route == 2 ? 0 : route == 1 ? 0 : hook.declined ? hook.declined : hook.declined_records);
    g_last_device = (!rc && route > 0 && made) ? 1 : 0;
    if (rc) { hgx_dbatch_destroy(made); hgx_batch_destroy(b); return rc; }
    if (g_last_device) { hgx_batch_destroy(b); *out = made; return HGX_OK; }
    hgx_dbatch_destroy(made);
    if (!b) { hgx_set_error("front end produced no batch"); return HGX_EINVAL; }
    const int rc2 = hgx_dbatch_create(out, b, st);
    hgx_batch_destroy(b);
    return rc2;
}

}   // namespace

extern "C" int hgx_parse_sam_dev(hgx_dbatch **out, const hgx_locus *loc, const char *sam, size_t n_bytes, const hgx_parse_opts *opts, void *stream) {
    ARGCHK(out && loc && opts);
    return parse_dev(out, (hipStream_t)stream, opts,
                     [&](hgx_batch **b, hgx_front_hook *hook, const hgx_parse_opts *o) { return hgx_parse_sam_hook(b, loc, sam, n_bytes, o, hook); });
}

extern "C" int hgx_parse_alignment_file_dev(hgx_dbatch **out, const hgx_locus *loc, const char *path, const char *regions,
                                            const hgx_parse_opts *opts, void *stream) {
    ARGCHK(out && loc && path && opts);
    return parse_dev(out, (hipStream_t)stream, opts,
                     [&](hgx_batch **b, hgx_front_hook *hook, const hgx_parse_opts *o) { return hgx_parse_alignment_file_hook(b, loc, path, regions, o, hook); });
}

// ---- one alignment file, many loci (typing_core.py:370: `for test_Gene_names in locus_list` over ONE alignment file) ---------------
// The reference pipes `samtools view F ref_allele | sort` once per locus (core:436-468): the file is decompressed and scanned as
// many times as there are loci.  An hgx_alignment is the file read ONCE, its bytes resident in HBM -- the SAM text, or the BAM
// stream inflated by k_bgzf_inflate_w -- and hgx_alignment_parse_dev is the per-locus rest: region filter, name order, record
// fields, filters, key grouping, pileup, decode, piece table and pair protocol as kernels over those bytes (read-only: the loci of
// a panel run side by side on streams of their own).  Whatever the kernels decline for a locus, and files that do not defer
// (several regions per locus, a stream beyond 4 GB), go through hgx_parse_alignment_file_dev on the path: the same batch.
struct hgx_alignment {
    std::string path;
    int dev = 0;
    bool resident = false, text = false;
    void *d_text = nullptr;
    size_t raw_bytes = 0, body0 = 0;
    std::vector<std::string> refs;
    long long bytes_up = 0;
    ~hgx_alignment() { if (d_text) hgx_pool_free(d_text); }
};

extern "C" int hgx_alignment_open(hgx_alignment **out, const char *path, int32_t n_threads, void *stream) {
    ARGCHK(out && path);
    *out = nullptr;
    hipStream_t st = (hipStream_t)stream;
    std::unique_ptr<hgx_alignment> A(new hgx_alignment());
    A->path = path;
    HIPCHK(hipGetDevice(&A->dev));
    if (hgx_switch_has("front", "host")) { *out = A.release(); return HGX_OK; }          // (every locus through the host stages)
    if (!hgx_switch_has("front", "device")) {
        // below the device front end's size gate (FE_MIN_BYTES of stream; BGZF deflates ~1 : 4) the file is not even read here: the
        // per-locus call on the path decides
        struct stat sb;
        unsigned char magic[2] = {0, 0};
        FILE *f = fopen(path, "rb");
        if (!f || fstat(fileno(f), &sb) != 0) { if (f) fclose(f); hgx_set_error("cannot open %s", path); return HGX_EINVAL; }
        const bool gz = fread(magic, 1, 2, f) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
        fclose(f);
        if ((size_t)sb.st_size < (gz ? FE_MIN_BYTES / 4 : FE_MIN_BYTES)) { *out = A.release(); return HGX_OK; }
    }
    DevBuf b_text, b_comp;
    bool up_failed = false;
    const unsigned char *comp_from = nullptr;
    size_t comp_n = 0;
    struct Drain { hipStream_t s; ~Drain() { (void)hipStreamSynchronize(s); } } drain{st};     // (an upload may still read the reader's buffer)
    try {
        hgx_align_lines al;
        hgx_big_alloc_scope pinned(hgx_front_alloc{pinned_alloc, pinned_release}, 1u << 20);
        al.on_raw = [&](const char *raw, size_t n_bytes, size_t begin, size_t end) {
            if (n_bytes >= (1ull << 32) - 64 || up_failed) { up_failed = true; return; }
            if (!b_text.p && b_text.alloc(n_bytes + 64)) { up_failed = true; return; }
            if (end > begin && hipMemcpyAsync((char *)b_text.p + begin, raw + begin, end - begin, hipMemcpyHostToDevice, st) != hipSuccess) up_failed = true;
            A->bytes_up += (long long)(end - begin);
        };
        al.defer_walk = true;
        al.defer_text = true;
        al.defer_min_bytes = 0;
        if (!hgx_switch_has("front", "host_inflate")) {
            al.comp_early = [&](const unsigned char *data, size_t n) {
                if (up_failed || b_comp.p || b_comp.alloc(n + 2048)) return;
                if (hipMemcpyAsync(b_comp.p, data, n, hipMemcpyHostToDevice, st) != hipSuccess ||
                    hipMemsetAsync((char *)b_comp.p + n, 0, 2048, st) != hipSuccess) { up_failed = true; return; }
                comp_from = data; comp_n = n;
            };
            al.comp_sync = [&]() { if (comp_from) (void)hipStreamSynchronize(st); };
            al.inflate_dev = [&](const unsigned char *data, size_t n, const std::vector<hgx_bgzf_block> &blocks, size_t total) -> int {
                if (up_failed || total >= (1ull << 32) - 64 || b_text.p) return 1;
                if (b_text.alloc(total + 64)) return 1;
                struct DrainC { hipStream_t s; ~DrainC() { (void)hipStreamSynchronize(s); } } drain_c{st};
                if (comp_from != data || comp_n != n) {
                    if (b_comp.p) { (void)hipStreamSynchronize(st); hgx_pool_free(b_comp.p); b_comp.p = nullptr; }
                    if (b_comp.alloc(n + 2048)) return 1;
                    if (hipMemcpyAsync(b_comp.p, data, n, hipMemcpyHostToDevice, st) != hipSuccess) return 1;
                    if (hipMemsetAsync((char *)b_comp.p + n, 0, 2048, st) != hipSuccess) return 1;
                }
                int bad = 0;
                void *stg = pinned_alloc(hgx_bgzf_inflate_staging_bytes(blocks.size()));
                struct Unstage { void *p; ~Unstage() { pinned_release(p); } } unstage{stg};
                if (hgx_bgzf_inflate_dev(b_comp.as<unsigned char>(), blocks.data(), blocks.size(), b_text.as<unsigned char>(), st, &bad, stg) != HGX_OK || bad) {
                    (void)hipStreamSynchronize(st);
                    hgx_pool_free(b_text.p);
                    b_text.p = nullptr;
                    return 1;
                }
                A->bytes_up += (long long)n;
                return 0;
            };
        }
        const int rc = hgx_read_alignment_lines(path, nullptr, n_threads, al, /*keep_binary=*/true);
        if (rc) return rc;
        HIPCHK(hipStreamSynchronize(st));
        if (al.deferred.on && !up_failed && b_text.p) {
            A->resident = true;
            A->text = al.deferred.text;
            A->body0 = al.deferred.body0;
            A->refs = al.ref_names;
            A->raw_bytes = al.raw_bytes;
            A->d_text = b_text.p;
            b_text.p = nullptr;
        }
    } catch (const std::exception &e) {
        hgx_set_error("hgx_alignment_open: %s", e.what());
        return HGX_ENOMEM;
    }
    *out = A.release();
    return HGX_OK;
}

extern "C" int hgx_alignment_close(hgx_alignment *al) { delete al; return HGX_OK; }

extern "C" int hgx_alignment_dims(const hgx_alignment *al, int32_t *resident, int32_t *is_text, size_t *stream_bytes, long long *bytes_to_device) {
    ARGCHK(al);
    if (resident) *resident = al->resident ? 1 : 0;
    if (is_text) *is_text = al->text ? 1 : 0;
    if (stream_bytes) *stream_bytes = al->raw_bytes;
    if (bytes_to_device) *bytes_to_device = al->bytes_up;
    return HGX_OK;
}

extern "C" int hgx_alignment_parse_dev(hgx_dbatch **out, hgx_alignment *al, const hgx_locus *loc, const char *regions, const hgx_parse_opts *opts,
                                       void *stream) {
    ARGCHK(out && al && loc && opts);
    *out = nullptr;
    hipStream_t st = (hipStream_t)stream;
    int dev = -1;
    HIPCHK(hipGetDevice(&dev));
    hgx_bam_deferred def;
    const bool host_opts = opts->interdist_exchange || opts->pileup_exchange || opts->pileup_exchange_dev;      // (a shard's exchanges: the per-path call shares them out)
    if (al->resident && dev == al->dev && !host_opts && !hgx_switch_has("front", "host") &&
        hgx_deferred_for_regions(regions, al->text, al->body0, al->refs, def) == 0) {
        // (size gate: the stream is in HBM already -- the kernels take any locus the host would not finish faster; the record count is
        // only known after the region filter, so the gate is on what the filter keeps)
        DevBuf b_dl;
        uint32_t n_dl = 0;
        int declined = 0;
        hgx_dbatch *made = nullptr;
        int rc;
        {
            struct DrainL { hipStream_t s; ~DrainL() { (void)hipStreamSynchronize(s); } } drain_l{st};
            rc = def.text ? sam_lines_dev((const char *)al->d_text, al->raw_bytes, def, st, b_dl, &n_dl, &declined)
                          : bam_lines_dev((const char *)al->d_text, {&def}, {(size_t)0}, {al->raw_bytes}, st, b_dl, &n_dl, &declined);
            if (!rc && !declined)
                rc = records_run(*const_cast<hgx_locus *>(loc), (const char *)al->d_text, al->raw_bytes, nullptr, n_dl, !def.text, 1, *opts, st, &made, nullptr,
                                 &declined, b_dl.as<LineRef>());
        }
        if (rc) { hgx_dbatch_destroy(made); return rc; }
        if (!declined && made) {
            g_last_route = 2; g_last_decline = 0; g_last_device = 1; g_last_bytes = 0; g_last_parts = 0;
            *out = made;
            return HGX_OK;
        }
        hgx_dbatch_destroy(made);
    }
    return hgx_parse_alignment_file_dev(out, loc, al->path.c_str(), regions, opts, stream);
}

// MANY tasks of one locus (the samples of a panel) in ONE pass of the record route: the tasks' files are read side by side on the
// host's threads, their bytes land in one device buffer, and every record carries its task -- keys, read ids and pairs never
// cross tasks, each task has its own pileup, the piece table is shared (what hgx_batch_merge makes of the tasks' own batches).
// *declined != 0: nothing was made (the caller runs the host front end per task and merges).
int hgx_front_many_dev(hgx_dbatch **out, hgx_front_totals *tot, const hgx_locus *loc, const char *const *paths, const char *const *regions,
                       const char *const *sams, const size_t *sam_bytes, int n_tasks, const hgx_parse_opts *opts, void *stream, int *declined) {
    ARGCHK(out && tot && loc && opts && declined && n_tasks >= 0);
    *out = nullptr;
    *declined = 0;
    hipStream_t st = (hipStream_t)stream;
    g_last_bytes = 0; g_last_route = 0; g_last_device = 0; g_last_decline = 0; g_last_parts = 0;
    auto decline = [&](int code) { *declined = code; g_last_decline = code; return (int)HGX_OK; };
    if (hgx_switch_has("front", "host")) { *declined = -1; g_last_decline = -1; return HGX_OK; }
    if (opts->keep_trace || opts->codis_choose_pairs || opts->interdist_exchange || opts->pileup_exchange || opts->pileup_exchange_dev) return decline(HGX_FE_DECLINE_OPTS);
    if (n_tasks < 1 || n_tasks > 65535) return decline(HGX_FE_DECLINE_SIZE);
    hgx_many_streams ms;
    const hgx_front_alloc mem{pinned_alloc, pinned_release};
    const bool prof = getenv("HGX_PARSE_PROFILE") != nullptr;
    double t_prev = now_ms();
    auto lap = [&](const char *what) {
        if (!prof) return;
        const double t = now_ms();
        fprintf(stderr, "[hgx_front_many] %-28s %8.2f ms\n", what, t - t_prev);
        t_prev = t;
    };
    // BAM files: left deflated -- the files' bytes go up as they are read, ONE launch inflates the BGZF blocks of all tasks, the
    // records are walked / filtered / name-sorted per task on the device (k_bgzf_inflate, k_bam_*); the host reads, hops through
    // the containers and inflates the header blocks.  Anything that does not fit goes the way below.
    if (paths && !hgx_switch_has("front", "host_inflate")) {
        std::vector<size_t> cbase((size_t)n_tasks + 1, 0), csize((size_t)n_tasks, 0);
        bool stat_ok = true;
        for (int t = 0; t < n_tasks; ++t) {
            struct stat sb;
            if (!paths[t] || stat(paths[t], &sb) != 0) { stat_ok = false; break; }
            csize[t] = (size_t)sb.st_size;
            cbase[(size_t)t + 1] = (cbase[t] + csize[t] + 63) & ~(size_t)63;
        }
        const size_t ctotal = cbase[(size_t)n_tasks];
        if (stat_ok && ctotal < (1ull << 32) - 4096) {
            DevBuf b_comp, b_text2, b_dl;
            struct DrainB { hipStream_t s; ~DrainB() { (void)hipStreamSynchronize(s); } } drain_b{st};
            ALLOC(b_comp, ctotal + 4096);
            int dev = 0;
            HIPCHK(hipGetDevice(&dev));
            std::vector<hgx_bgzf_task> bt;
            std::atomic<bool> up_bad{false};
            auto on_bt = [&](int t) {
                if (bt[t].n != csize[t] || hipSetDevice(dev) != hipSuccess ||
                    hipMemcpyAsync((char *)b_comp.p + cbase[t], bt[t].data, bt[t].n, hipMemcpyHostToDevice, st) != hipSuccess) up_bad = true;
            };
            struct FreeTasks { std::vector<hgx_bgzf_task> &v; hipStream_t s; ~FreeTasks() { (void)hipStreamSynchronize(s); for (auto &x : v) hgx_host_free(x.data); } } free_tasks{bt, st};
            int rcb = hgx_bgzf_tasks_read(bt, paths, regions, n_tasks, opts->n_threads, &mem, on_bt);
            if (rcb) return rcb;
            lap("read (deflated; uploads issued)");
            bool all_ok = !up_bad.load();
            for (int t = 0; t < n_tasks && all_ok; ++t) all_ok = bt[t].ok;
            std::vector<size_t> pbase((size_t)n_tasks + 1, 0), psize((size_t)n_tasks, 0);
            size_t n_blocks = 0;
            for (int t = 0; t < n_tasks && all_ok; ++t) {
                psize[t] = bt[t].total;
                pbase[(size_t)t + 1] = (pbase[t] + bt[t].total + 63) & ~(size_t)63;
                n_blocks += bt[t].blocks.size();
            }
            const size_t ptotal = pbase[(size_t)n_tasks];
            if (all_ok && ptotal < (1ull << 32) - 64 && (hgx_switch_has("front", "device") || ptotal >= FE_MIN_DEFER_BYTES)) {
                std::vector<hgx_bgzf_block> all;
                all.reserve(n_blocks);
                for (int t = 0; t < n_tasks; ++t)
                    for (hgx_bgzf_block b : bt[t].blocks) { b.in_off += cbase[t]; b.out_off += pbase[t]; all.push_back(b); }
                ALLOC(b_text2, ptotal + 64);
                HIPCHK(hipMemsetAsync((char *)b_comp.p + ctotal, 0, 4096, st));
                int bad = 0;
                rcb = hgx_bgzf_inflate_dev(b_comp.as<unsigned char>(), all.data(), all.size(), b_text2.as<unsigned char>(), st, &bad, nullptr);
                if (rcb) return rcb;
                lap("BGZF inflate (device)");
                if (!bad) {
                    std::vector<const hgx_bam_deferred *> defs((size_t)n_tasks);
                    for (int t = 0; t < n_tasks; ++t) defs[t] = &bt[t].def;
                    pbase.resize((size_t)n_tasks);
                    uint32_t n_dl = 0;
                    int dec = 0;
                    rcb = bam_lines_dev(b_text2.as<char>(), defs, pbase, psize, st, b_dl, &n_dl, &dec);
                    if (rcb) return rcb;
                    if (!dec) {
                        hgx_dbatch *made = nullptr;
                        rcb = records_run(*const_cast<hgx_locus *>(loc), b_text2.as<char>(), ptotal, nullptr, n_dl, true, n_tasks, *opts, st, &made, tot, &dec,
                                          b_dl.as<LineRef>());
                        (void)hipStreamSynchronize(st);
                        lap("device stages");
                        if (rcb) { hgx_dbatch_destroy(made); return rcb; }
                        if (!dec && made) {
                            g_last_bytes += (long long)ctotal;
                            g_last_route = 2; g_last_device = 1;
                            *out = made;
                            return HGX_OK;
                        }
                        hgx_dbatch_destroy(made);
                    }
                }
            }
        }
    }
    // The device text buffer is reserved before the sizes are known (files: 16x their bytes on disk -- BGZF rarely inflates that
    // far -- at most what 32-bit offsets hold), and every reader thread takes the next free range for its task and starts the
    // upload the moment its bytes are complete: the transfers run under the other tasks' inflate / walk / sort.  A task that no
    // longer fits only takes its range: everything is sent again, to a buffer of the right size, after the reads.
    size_t cap = 0;
    for (int t = 0; t < n_tasks; ++t) {
        if (paths) {
            struct stat sb;
            cap += (paths[t] && stat(paths[t], &sb) == 0 ? (size_t)sb.st_size * 16 : 0) + (64u << 10);
        } else cap += (sam_bytes[t] + 63) & ~(size_t)63;
    }
    cap = std::min<size_t>(cap, (1ull << 32) - 128);
    DevBuf b_text;
    struct Drain { hipStream_t s; ~Drain() { (void)hipStreamSynchronize(s); } } drain{st};       // (uploads read the readers' buffers)
    ALLOC(b_text, cap + 64);
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    std::atomic<size_t> cursor{0};
    std::atomic<long long> sent{0};
    std::atomic<bool> late{false};
    auto on_task = [&](int t) -> int {
        const size_t n = (ms.raw_bytes[t] + 63) & ~(size_t)63;
        const size_t off = cursor.fetch_add(n);
        ms.base[t] = off;
        if (off + n > cap) { late = true; return HGX_OK; }
        if (!ms.raw_bytes[t]) return HGX_OK;
        HIPCHK(hipSetDevice(dev));                                         // (a reader thread of the host pool)
        HIPCHK(hipMemcpyAsync((char *)b_text.p + off, ms.raw[t], ms.raw_bytes[t], hipMemcpyHostToDevice, st));
        sent += (long long)ms.raw_bytes[t];
        return HGX_OK;
    };
    int rc = hgx_many_read(ms, paths, regions, sams, sam_bytes, n_tasks, opts->n_threads, &mem, on_task);
    if (rc) return rc;
    lap("read (+ uploads issued)");
    const size_t total = cursor.load(), n_lines = ms.line_base[(size_t)n_tasks];
    ms.base[(size_t)n_tasks] = total;
    if (ms.mixed) return decline(HGX_FE_DECLINE_OPTS);                   // SAM text and BAM records in one batch: per task on the host
    if (!hgx_switch_has("front", "device") && n_lines < FE_MIN_RECORDS) return decline(HGX_FE_DECLINE_SMALL);
    if (total >= (1ull << 32) - 64 || n_lines >= (1ull << 30)) return decline(HGX_FE_DECLINE_SIZE);
    if (late.load() || hgx_switch_has("front", "late")) {
        HIPCHK(hipStreamSynchronize(st));
        hgx_pool_free(b_text.p);
        b_text.p = nullptr;
        ALLOC(b_text, total + 64);
        for (int t = 0; t < n_tasks; ++t) {
            if (!ms.raw_bytes[t]) continue;
            HIPCHK(hipMemcpyAsync((char *)b_text.p + ms.base[t], ms.raw[t], ms.raw_bytes[t], hipMemcpyHostToDevice, st));
            sent += (long long)ms.raw_bytes[t];
        }
    }
    g_last_bytes += sent.load();
    LineRef *h_lines = (LineRef *)pinned_alloc(std::max<size_t>(n_lines, 1) * sizeof(LineRef));
    if (!h_lines) { hgx_set_error("pinned allocation of the line table failed"); return HGX_ENOMEM; }
    struct Unpin { void *p; ~Unpin() { pinned_release(p); } } unpin{h_lines};
    hgx_many_lines(ms, h_lines, opts->n_threads);
    lap("line table");
    if (prof) { (void)hipStreamSynchronize(st); lap("uploads done"); }
    hgx_dbatch *made = nullptr;
    int dec = 0;
    rc = records_run(*const_cast<hgx_locus *>(loc), b_text.as<char>(), total, h_lines, n_lines, ms.binary, n_tasks, *opts, st, &made, tot, &dec);
    (void)hipStreamSynchronize(st);
    lap("device stages");
    if (rc) { hgx_dbatch_destroy(made); return rc; }
    if (dec || !made) { hgx_dbatch_destroy(made); return decline(dec ? dec : HGX_FE_DECLINE_SIZE); }
    g_last_route = 2; g_last_device = 1;
    *out = made;
    return HGX_OK;
}

extern "C" int hgx_front_last_parts(int32_t *parts) {
    ARGCHK(parts);
    *parts = g_last_parts;
    return HGX_OK;
}
extern "C" int hgx_front_last(int32_t *route, int32_t *decline_code, int64_t *bytes_to_device) {
    if (route) *route = g_last_device ? g_last_route : 0;
    if (decline_code) *decline_code = g_last_decline;
    if (bytes_to_device) *bytes_to_device = g_last_bytes;
    return HGX_OK;
}

// a device batch back on the host (tests, tools): pieces, masks, refs -- and the pileup tables where the device front end made them
extern "C" int hgx_dbatch_to_host(const hgx_dbatch *d, hgx_batch **out) {
    ARGCHK(d && out);
    hgx_batch *b = new hgx_batch();
    b->pieces.resize((size_t)d->n_pieces);
    b->masks.resize((size_t)d->n_mask_u32);
    b->pair_off.assign((size_t)d->n_pairs + 1, 0);
    b->pair_ref.resize((size_t)d->n_refs);
    b->n_reads = d->n_reads;
    auto down = [&](void *dst, const void *src, size_t n) -> int {
        if (n) HIPCHK(hipMemcpy(dst, src, n, hipMemcpyDeviceToHost));
        return HGX_OK;
    };
    int rc = down(b->pieces.data(), d->d_pieces, b->pieces.size() * sizeof(hgx_piece));
    if (!rc) rc = down(b->masks.data(), d->d_masks, b->masks.size() * 4);
    if (!rc) rc = down(b->pair_off.data(), d->d_pair_off, b->pair_off.size() * 4);
    if (!rc) rc = down(b->pair_ref.data(), d->d_pair_ref, b->pair_ref.size() * 4);
    if (!rc && d->d_counts && d->n_ref > 0) {
        b->counts.resize((size_t)d->n_ref * 6);
        b->nt_set.resize((size_t)d->n_ref);
        rc = down(b->counts.data(), d->d_counts, b->counts.size() * 4);
        if (!rc) rc = down(b->nt_set.data(), d->d_nt_set, b->nt_set.size());
    }
    if (rc) { delete b; return rc; }
    for (const auto &t : d->trace) b->trace.push_back(TraceRec{t});
    *out = b;
    return HGX_OK;
}
