// hgx_inflate.hip -- BGZF inflate on the device (row 8f-3): ONE WAVEFRONT PER BGZF BLOCK.
// Two forms: k_bgzf_inflate_w (round 5, the default: the symbol loop lane-parallel, 64 bit offsets per window -- see its header
// below) and k_bgzf_inflate (round 4: one symbol per trip; test switch front=inflate_v1, the comparison form).  What follows
// describes the first form's structure, which the second shares but for the symbol loop and the input path.
//
// A BGZF file (SAM/BAM specification section 4.1) is a sequence of independent gzip members of at most 64 KB of payload each, with
// the compressed size in an extra field and CRC-32 / ISIZE behind the deflate stream: the host only hops from header to header
// (hgx_bam.cpp) and hands over (offset, length, CRC, ISIZE, output offset) per block.  Here a wavefront decodes its block's
// DEFLATE stream (RFC 1951: stored, fixed and dynamic Huffman blocks) with wave-UNIFORM control flow -- every lane runs the same
// bit reader on the same bits, so nothing diverges -- and the 64 lanes share the byte work:
//   input    256 bytes at a time in a register per lane (the next 256 already requested), a dword handed to the bit reader with
//            a v_readlane: no memory latency inside the symbol loop;
//   symbols  a 10-bit (literal/length) and an 8-bit (distance) table of {symbol, code length} in LDS, built per DEFLATE block by
//            the lanes; longer codes by the canonical first-code walk (counts per length, symbols sorted by (length, symbol));
//   output   an 8 KB ring in LDS holds the NEAR part of DEFLATE's 32 KB window: a literal is one byte store, a match up to ~7.9 KB
//            back is copied by the lanes inside the ring (source index modulo the distance: every source byte lies before the
//            match); a match farther back reads what has already left for memory (L1-bypassing loads behind a wait for the
//            wavefront's own stores).  A small ring is what buys occupancy: 13 KB of LDS per wavefront seats three per SIMD, and
//            with sequential symbol decoding the number of wavefronts in flight IS the throughput.  2 KB at a time leave for
//            memory, each lane taking the CRC-32 of its 32-byte piece on the way, the pieces joined with the "32 zero bytes"
//            operator of the CRC (crc32_combine's algebra, the matrix precomputed on the host).
// A block's verdict (0 = inflated, CRC-32 and ISIZE right) is written per block: anything else makes the caller fall back to the
// host reader, which reproduces the failure with its own message.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <type_traits>
#include <vector>

#include "hgx_common.hpp"
#include "hgx_internal.hpp"

namespace {

#ifndef HGX_INF_RING
#define HGX_INF_RING 8192
#define HGX_INF_FLUSH 2048
#endif
constexpr int RING = HGX_INF_RING, RMASK = RING - 1, FLUSH = HGX_INF_FLUSH, PIECE = FLUSH / 64;      // (see the header: the ring holds the near window only)
#ifndef HGX_INF_LITP
#define HGX_INF_LITP 9
#endif
#ifndef HGX_INF_DISTP
#define HGX_INF_DISTP 8
#endif
constexpr int LIT_P = HGX_INF_LITP, DIST_P = HGX_INF_DISTP;

struct HuffLds {
    uint16_t pt[1 << LIT_P];             // primary table: symbol << 4 | code length (0 = longer than the table's bits, or no such code)
    uint16_t sorted[288];                // symbols by (length, symbol)
    uint16_t count[16], first[16], offs[16];
};
struct DistLds {
    uint16_t pt[1 << DIST_P];
    uint16_t sorted[32];
    uint16_t count[16], first[16], offs[16];
};
struct InfLds {
    unsigned char ring[RING];
    HuffLds lit;
    DistLds dist;
    unsigned char lens[384];             // [0, 19): the code-length code; [32, 32 + n_lit + n_dist): the two codes' lengths
    uint32_t crc_piece[64];
};

__constant__ uint8_t c_cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct BlockDesc { uint32_t in_off, in_len, out_off, out_len, crc; };
struct CrcOp { uint32_t col[7][32]; };       // level k: crc -> crc advanced by PIECE << k zero bytes (k = 6: a whole FLUSH)

enum { INF_OK = 0, INF_BAD_BLOCK_TYPE = 1, INF_BAD_STORED = 2, INF_BAD_CODE = 3, INF_BAD_DIST = 4, INF_OVERRUN = 5, INF_BAD_SIZE = 6, INF_BAD_CRC = 7,
       INF_BAD_LENGTHS = 8, INF_INPUT_END = 9 };

// the wave's bit reader: bit 0 of `buf` is the next bit of the stream
struct Bits {
    const uint32_t *base;                // dword-aligned start of the input
    uint32_t cur, nxt;                   // this lane's dword of the current / the next 256-byte chunk
    uint32_t chunk, idx;                 // chunk number, next dword of the current chunk to hand out
    uint64_t buf;
    int cnt;
    uint32_t n_dwords;                   // dwords that belong to the block (a read beyond them is padding: INF_INPUT_END if it gets used)
    uint32_t taken;                      // dwords handed out
};
__device__ __forceinline__ void bits_init(Bits &b, const unsigned char *in, uint32_t in_len, int lane) {
    const uintptr_t a = (uintptr_t)in;
    b.base = (const uint32_t *)(a & ~(uintptr_t)3);
    const int skip = (int)(a & 3);
    b.n_dwords = (uint32_t)((skip + in_len + 3) / 4);
    b.chunk = 0; b.idx = 0; b.taken = 0;
    b.cur = b.base[lane];
    b.nxt = b.base[64 + lane];
    b.buf = 0; b.cnt = 0;
    // the first dword: drop the bytes in front of the stream
    const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)b.cur, 0);
    b.buf = (uint64_t)(w >> (8 * skip));
    b.cnt = 32 - 8 * skip;
    b.idx = 1; b.taken = 1;
}
__device__ __forceinline__ void bits_refill(Bits &b, int lane) {
    while (b.cnt <= 32) {
        if (b.idx == 64) {
            b.cur = b.nxt;
            b.chunk += 1;
            // (a damaged stream may ask for far more bits than its block holds: beyond the block's dwords + 3/4 KB -- inside the
            // padding the caller guarantees behind the last block -- the reader is fed zeros, and the decode ends in an error)
            const uint32_t at = (b.chunk + 1) * 64;
            b.nxt = at < b.n_dwords + 192u ? b.base[(size_t)at + lane] : 0u;
            b.idx = 0;
        }
        const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)b.cur, (int)b.idx);
        b.buf |= (uint64_t)w << b.cnt;
        b.cnt += 32;
        b.idx += 1;
        b.taken += 1;
    }
}
__device__ __forceinline__ uint32_t bits_peek(const Bits &b, int n) { return (uint32_t)(b.buf & ((1ull << n) - 1ull)); }
__device__ __forceinline__ void bits_drop(Bits &b, int n) { b.buf >>= n; b.cnt -= n; }
__device__ __forceinline__ uint32_t bits_take(Bits &b, int n) { const uint32_t v = bits_peek(b, n); bits_drop(b, n); return v; }

// Canonical Huffman tables from code lengths lens[0..n): false = over-subscribed (or, unless `allow_incomplete`, incomplete) set of
// lengths.  All 64 lanes work: a symbol per lane and 64 symbols per round; the number of codes of every length and a symbol's rank
// among the codes of its length come from one ballot per (round, length) -- the counters live in scalar registers under fully
// unrolled length loops.  (The first form walked the symbols on lane 0 with its per-length cursors in a register array: every
// dynamic index became a 16-way compare-and-select chain, ~150 instructions per symbol, a fifth of the kernel's scalar work.)
enum { K_LIT = 0, K_MATCH = 1, K_EOB = 2, K_SLOW = 3, K_BAD = 4, K_SUB = 5 };
// What a table entry says about symbol s with a code of L bits.  MODE 0: symbol << 4 | L (bit 15 marks a literal on request) -- the
// first form's tables and the code-length code.  MODE 1 / 2 (k_bgzf_inflate_w): everything the per-lane decode would otherwise work
// out from the symbol, computed once per symbol here instead of once per lane and window there --
//   literal/length: L | extra bits << 4 | kind << 8 | (literal value or base length) << 16
//   distance:       L | extra bits << 4 | (no such distance) << 8 | base distance << 16
template <int MODE>
__device__ __forceinline__ uint32_t huff_entry(uint32_t s, uint32_t L, bool mark_literals) {
    if (MODE == 1) {
        if (s < 256u) return L | ((uint32_t)K_LIT << 8) | (s << 16);
        if (s == 256u) return L | ((uint32_t)K_EOB << 8);
        if (s > 285u) return L | ((uint32_t)K_BAD << 8);
        const uint32_t li = s - 257u;
        const uint32_t le = li < 8u || li == 28u ? 0u : (li - 4u) >> 2;
        const uint32_t lbase = li < 8u ? 3u + li : li == 28u ? 258u : ((4u + (li & 3u)) << le) + 3u;
        return L | (le << 4) | ((uint32_t)K_MATCH << 8) | (lbase << 16);
    }
    if (MODE == 2) {
        if (s >= 30u) return L | 0x100u;
        const uint32_t de = s < 4u ? 0u : (s >> 1) - 1u;
        const uint32_t dbase = s < 4u ? s + 1u : ((2u + (s & 1u)) << de) + 1u;
        return L | (de << 4) | (dbase << 16);
    }
    return (s << 4) | L | ((mark_literals && s < 256u) ? 0x8000u : 0u);
}
template <class T, int P, int MODE = 0>
__device__ bool huff_build(T &H, const unsigned char *lens, int n, int lane, bool allow_incomplete, bool mark_literals = false) {
    for (int i = lane; i < (1 << P); i += 64) H.pt[i] = 0;
    __builtin_amdgcn_wave_barrier();
    uint32_t cnt[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) cnt[v] = 0;
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int s = c0 + lane;
        const int L = s < n ? (int)lens[s] : 0;
#pragma unroll
        for (int v = 1; v < 16; ++v) cnt[v] += (uint32_t)__popcll(__ballot(L == v));
    }
    int left = 1;
    bool ok = true;
#pragma unroll
    for (int v = 1; v < 16; ++v) { left = (left << 1) - (int)cnt[v]; ok = ok && left >= 0; }
    if (left > 0 && !allow_incomplete) ok = false;
    if (!ok) return false;                                            // (uniform)
    {
        uint32_t my_c = 0, my_f = 0, my_o = 0, c = 0, at = 0;
#pragma unroll
        for (int v = 1; v < 16; ++v) {
            c = (c + cnt[v - 1]) << 1;                                // first code of length v; `at` = codes shorter than v
            if (lane == v) { my_c = cnt[v]; my_f = c; my_o = at; }
            at += cnt[v];
        }
        if (lane < 16) { H.count[lane] = (uint16_t)my_c; H.first[lane] = (uint16_t)my_f; H.offs[lane] = (uint16_t)my_o; }
    }
    uint32_t seen[16];                                                // codes of every length in the rounds so far
#pragma unroll
    for (int v = 0; v < 16; ++v) seen[v] = 0;
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int s = c0 + lane;
        const int L = s < n ? (int)lens[s] : 0;
        uint32_t rank = 0, fst = 0, off = 0, c = 0, at = 0;
#pragma unroll
        for (int v = 1; v < 16; ++v) {
            c = (c + cnt[v - 1]) << 1;
            const unsigned long long m = __ballot(L == v);
            if (L == v) { rank = seen[v] + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); fst = c; off = at; }
            seen[v] += (uint32_t)__popcll(m);
            at += cnt[v];
        }
        if (L != 0) {
            H.sorted[off + rank] = (uint16_t)s;                       // symbols by (length, symbol)
            if (L <= P) {
                const uint32_t rev = __brev(fst + rank) >> (32 - L);
                const auto e = (typename std::remove_reference<decltype(H.pt[0])>::type)huff_entry<MODE>((uint32_t)s, (uint32_t)L, mark_literals);
                for (uint32_t k = rev; k < (1u << P); k += 1u << L) H.pt[k] = e;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    return true;
}
// an INCOMPLETE set of code lengths is legal only as zlib's inflate_table takes it: exactly one code, of one bit -- or, for the
// distance code of a block without matches, no code at all
__device__ bool huff_incomplete_ok(const unsigned char *lens, int n, int lane, bool allow_empty) {
    uint32_t n_codes = 0, n_one = 0;
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int s = c0 + lane;
        const int L = s < n ? (int)lens[s] : 0;
        n_codes += (uint32_t)__popcll(__ballot(L != 0));
        n_one += (uint32_t)__popcll(__ballot(L == 1));
    }
    return (n_codes == 1 && n_one == 1) || (allow_empty && n_codes == 0);
}
// one symbol: the primary table, or the walk over the longer lengths; -1 = no such code.  (At least 15 bits are in the buffer.)
template <class T, int P>
__device__ __forceinline__ int huff_decode(const T &H, Bits &b) {
    const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)H.pt[bits_peek(b, P)]);
    if (e & 15) { bits_drop(b, e & 15); return e >> 4; }
    const uint32_t rev15 = __brev(bits_peek(b, 15)) >> 17;           // the next 15 bits, first bit most significant
    for (int L = P + 1; L <= 15; ++L) {
        const uint32_t c = rev15 >> (15 - L);
        const uint32_t d = c - H.first[L];
        if (d < H.count[L]) { bits_drop(b, L); return H.sorted[H.offs[L] + d]; }
    }
    return -1;
}

__device__ __forceinline__ uint32_t crc_bytes(const unsigned char *ring, uint32_t pos, int n) {
    uint32_t c = 0xFFFFFFFFu;
    for (int i = 0; i < n; ++i) {
        c ^= ring[(pos + i) & RMASK];
#pragma unroll
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
    }
    return ~c;
}
__device__ __forceinline__ uint32_t crc_advance(const CrcOp &op, int level, uint32_t v) {
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) r ^= op.col[level][i] & (0u - ((v >> i) & 1u));
    return r;
}

__global__ void __launch_bounds__(64) k_bgzf_inflate(const unsigned char *__restrict__ in, const BlockDesc *__restrict__ blocks, int n_blocks,
                                                     unsigned char *__restrict__ out, CrcOp op, uint32_t *__restrict__ verdict) {
    extern __shared__ unsigned char inf_lds_raw[];
    InfLds &S = *reinterpret_cast<InfLds *>(inf_lds_raw);
    const int lane = threadIdx.x;
    const int bi = blockIdx.x;
    if (bi >= n_blocks) return;
    const BlockDesc B = blocks[bi];
    unsigned char *dst = out + B.out_off;
    uint32_t wpos = 0, fpos = 0, crc = 0;
    int err = INF_OK;
    auto flush = [&](uint32_t n) {             // the n oldest pending bytes leave the ring (n = FLUSH, or the rest at the end)
        for (uint32_t i = lane; i < n; i += 64) dst[fpos + i] = S.ring[(fpos + i) & RMASK];
        // CRC-32: a PIECE-byte piece per lane; a full FLUSH is joined pairwise in six rounds (piece 2j advanced by the length of
        // piece 2j + 1 and xor-ed with it, lengths doubling), the last, partial one piece by piece
        const uint32_t n_piece = (n + PIECE - 1) / PIECE;
        uint32_t mine = 0;
        if ((uint32_t)lane < n_piece) {
            const uint32_t p0 = fpos + (uint32_t)PIECE * lane;
            const int len = (int)min((uint32_t)PIECE, n - (uint32_t)PIECE * lane);
            mine = crc_bytes(S.ring, p0, len);
        }
        if (n == (uint32_t)FLUSH) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const uint32_t other = (uint32_t)__shfl_down((int)mine, 1 << k, 64);
                mine = crc_advance(op, k, mine) ^ other;           // (meaningful on the lanes that are multiples of 2 << k; lane 0 in the end)
            }
            crc = crc_advance(op, 6, crc) ^ (uint32_t)__builtin_amdgcn_readfirstlane((int)mine);
        } else {
            S.crc_piece[lane] = mine;
            __builtin_amdgcn_wave_barrier();
            for (uint32_t p = 0; p < n_piece; ++p) {
                const uint32_t len = min((uint32_t)PIECE, n - (uint32_t)PIECE * p);
                if (len == (uint32_t)PIECE) crc = crc_advance(op, 0, crc);
                else for (uint32_t k = 0; k < 8 * len; ++k) crc = (crc >> 1) ^ (0xEDB88320u & (0u - (crc & 1u)));
                crc ^= S.crc_piece[p];
            }
            __builtin_amdgcn_wave_barrier();
        }
        fpos += n;
    };
    if (B.out_len > 0) {
        Bits b;
        bits_init(b, in + B.in_off, B.in_len, lane);
        bool last = false;
        while (!last && !err) {
            bits_refill(b, lane);
            last = bits_take(b, 1) != 0;
            const uint32_t type = bits_take(b, 2);
            if (type == 0) {                                        // stored
                bits_drop(b, b.cnt & 7);
                bits_refill(b, lane);
                const uint32_t len = bits_take(b, 16), nlen = bits_take(b, 16);
                if ((len ^ 0xFFFFu) != nlen) { err = INF_BAD_STORED; break; }
                if (wpos + len > B.out_len) { err = INF_OVERRUN; break; }
                for (uint32_t i = 0; i < len; ++i) {
                    bits_refill(b, lane);
                    const uint32_t v = bits_take(b, 8);
                    if (lane == 0) S.ring[wpos & RMASK] = (unsigned char)v;
                    wpos += 1;
                    if (wpos - fpos >= (uint32_t)FLUSH) { __builtin_amdgcn_wave_barrier(); flush(FLUSH); }
                }
                continue;
            }
            if (type == 3) { err = INF_BAD_BLOCK_TYPE; break; }
            int n_lit, n_dist;
            if (type == 1) {                                        // fixed codes (RFC 1951, 3.2.6)
                for (int s = lane; s < 288; s += 64) S.lens[s] = (unsigned char)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
                if (lane < 32) S.lens[288 + lane] = 5;
                n_lit = 288; n_dist = 30;
                __builtin_amdgcn_wave_barrier();
                if (!huff_build<HuffLds, LIT_P>(S.lit, S.lens, 288, lane, false, true)) { err = INF_BAD_LENGTHS; break; }
                if (!huff_build<DistLds, DIST_P>(S.dist, S.lens + 288, 32, lane, true)) { err = INF_BAD_LENGTHS; break; }   // (32 five-bit codes, two of them never sent)
            } else {                                                // dynamic codes (3.2.7)
                n_lit = (int)bits_take(b, 5) + 257;
                n_dist = (int)bits_take(b, 5) + 1;
                const int n_cl = (int)bits_take(b, 4) + 4;
                if (n_lit > 286 || n_dist > 30) { err = INF_BAD_LENGTHS; break; }
                if (lane < 19) S.lens[lane] = 0;
                __builtin_amdgcn_wave_barrier();
                for (int i = 0; i < n_cl; ++i) {
                    bits_refill(b, lane);
                    const uint32_t v = bits_take(b, 3);
                    if (lane == 0) S.lens[c_cl_order[i]] = (unsigned char)v;
                }
                __builtin_amdgcn_wave_barrier();
                // the code-length code goes through the distance slot (19 symbols, at most 7 bits)
                if (!huff_build<DistLds, DIST_P>(S.dist, S.lens, 19, lane, false)) { err = INF_BAD_LENGTHS; break; }
                int at = 0, prev = 0;
                const int total = n_lit + n_dist;
                while (at < total && !err) {
                    bits_refill(b, lane);
                    const int sym = huff_decode<DistLds, DIST_P>(S.dist, b);
                    if (sym < 0) { err = INF_BAD_CODE; break; }
                    int rep = 1, val = sym;
                    if (sym == 16) { if (at == 0) { err = INF_BAD_LENGTHS; break; } rep = 3 + (int)bits_take(b, 2); val = prev; }
                    else if (sym == 17) { rep = 3 + (int)bits_take(b, 3); val = 0; }
                    else if (sym == 18) { rep = 11 + (int)bits_take(b, 7); val = 0; }
                    if (at + rep > total) { err = INF_BAD_LENGTHS; break; }
                    if (lane < rep) S.lens[32 + at + lane] = (unsigned char)val;         // (rep <= 138: three rounds at most)
                    if (lane + 64 < rep) S.lens[32 + at + lane + 64] = (unsigned char)val;
                    if (lane + 128 < rep) S.lens[32 + at + lane + 128] = (unsigned char)val;
                    at += rep;
                    prev = val;
                }
                if (err) break;
                __builtin_amdgcn_wave_barrier();
                if (S.lens[32 + 256] == 0) { err = INF_BAD_LENGTHS; break; }              // no end-of-block code
                if (!huff_build<HuffLds, LIT_P>(S.lit, S.lens + 32, n_lit, lane, false, true)) {
                    // (an incomplete literal/length code is legal only when it is ONE code of one bit: zlib's inflate_table; so here)
                    if (!huff_incomplete_ok(S.lens + 32, n_lit, lane, false) || !huff_build<HuffLds, LIT_P>(S.lit, S.lens + 32, n_lit, lane, true, true)) { err = INF_BAD_LENGTHS; break; }
                }
                if (!huff_build<DistLds, DIST_P>(S.dist, S.lens + 32 + n_lit, n_dist, lane, false)) {
                    if (!huff_incomplete_ok(S.lens + 32 + n_lit, n_dist, lane, true) || !huff_build<DistLds, DIST_P>(S.dist, S.lens + 32 + n_lit, n_dist, lane, true)) { err = INF_BAD_LENGTHS; break; }
                }
            }
            // ---- the symbols of the block -------------------------------------------------------------------------------
            // (literals are gathered eight at a time: one byte store by eight lanes instead of eight stores by one)
            uint64_t lit_acc = 0;
            uint32_t lit_n = 0;
            auto lit_out = [&]() {
                if (lit_n) {
                    if ((uint32_t)lane < lit_n) S.ring[(wpos + lane) & RMASK] = (unsigned char)(lit_acc >> (8 * lane));
                    wpos += lit_n;
                    lit_n = 0;
                    lit_acc = 0;
                }
            };
            for (;;) {
                // runs of literals: the shortest path through the table there is -- peek, look up, drop, gather
                uint32_t e;
                for (;;) {
                    bits_refill(b, lane);
                    e = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.lit.pt[bits_peek(b, LIT_P)]);
                    if (!(e & 0x8000u)) break;
                    bits_drop(b, (int)(e & 15u));
                    lit_acc |= (uint64_t)((e >> 4) & 255u) << (8 * lit_n);
                    lit_n += 1;
                    if (lit_n == 8) {
                        if (wpos + lit_n > B.out_len) { err = INF_OVERRUN; break; }
                        lit_out();
                        if (wpos - fpos >= (uint32_t)FLUSH) { __builtin_amdgcn_wave_barrier(); flush(FLUSH); }
                    }
                }
                if (err) break;
                int sym;
                if (e & 15u) { bits_drop(b, (int)(e & 15u)); sym = (int)(e >> 4); }
                else {                                              // a code longer than the table's bits (or none at all)
                    sym = -1;
                    const uint32_t rev15 = __brev(bits_peek(b, 15)) >> 17;
                    for (int L = LIT_P + 1; L <= 15; ++L) {
                        const uint32_t c = rev15 >> (15 - L);
                        const uint32_t d = c - S.lit.first[L];
                        if (d < S.lit.count[L]) { bits_drop(b, L); sym = S.lit.sorted[S.lit.offs[L] + d]; break; }
                    }
                    sym = __builtin_amdgcn_readfirstlane(sym);
                }
                if (sym < 0) { err = INF_BAD_CODE; break; }
                if (sym < 256) {
                    lit_acc |= (uint64_t)(uint32_t)sym << (8 * lit_n);
                    lit_n += 1;
                    if (lit_n < 8) continue;
                    if (wpos + lit_n > B.out_len) { err = INF_OVERRUN; break; }
                    lit_out();
                } else if (sym == 256) {
                    if (wpos + lit_n > B.out_len) { err = INF_OVERRUN; break; }
                    lit_out();
                    break;
                } else {
                    if (wpos + lit_n > B.out_len) { err = INF_OVERRUN; break; }
                    lit_out();
                    const int li = sym - 257;
                    if (li >= 29) { err = INF_BAD_CODE; break; }
                    // base and extra bits of the length / distance codes (RFC 1951, 3.2.5) by arithmetic: a table in memory would
                    // put two dependent loads on the path of every match
                    const int le = li < 8 || li == 28 ? 0 : (li - 4) >> 2;
                    const uint32_t lbase = li < 8 ? 3u + (uint32_t)li : li == 28 ? 258u : ((4u + ((uint32_t)li & 3u)) << le) + 3u;
                    const uint32_t len = lbase + bits_take(b, le);
                    bits_refill(b, lane);
                    const int ds = huff_decode<DistLds, DIST_P>(S.dist, b);
                    if (ds < 0 || ds >= 30) { err = INF_BAD_CODE; break; }
                    const int de = ds < 4 ? 0 : (ds >> 1) - 1;
                    const uint32_t dbase = ds < 4 ? (uint32_t)ds + 1u : ((2u + ((uint32_t)ds & 1u)) << de) + 1u;
                    const uint32_t dist = dbase + bits_take(b, de);
                    if (dist > wpos) { err = INF_BAD_DIST; break; }
                    if (wpos + len > B.out_len) { err = INF_OVERRUN; break; }
                    __builtin_amdgcn_wave_barrier();
                    // every source byte lies before the match: byte i comes from (i mod dist) bytes into the last `dist` bytes
                    if (dist >= len && dist <= (uint32_t)(RING - 258)) {                 // (the usual case: source and target apart)
                        for (uint32_t i = lane; i < len; i += 64) S.ring[(wpos + i) & RMASK] = S.ring[(wpos - dist + i) & RMASK];
                    } else if (dist == 1) {                                              // a run of one byte
                        const unsigned char v = S.ring[(wpos - 1) & RMASK];
                        for (uint32_t i = lane; i < len; i += 64) S.ring[(wpos + i) & RMASK] = v;
                    } else if (dist < len) {
                        // i mod dist without an integer division: i < 258 and dist < 258, and (i + 0.5) / dist is at least 0.5 / 258
                        // away from every integer -- far more than the error of the reciprocal
                        const float rcp = __frcp_rn((float)dist);
                        for (uint32_t i = lane; i < len; i += 64) {
                            const uint32_t q = (uint32_t)(((float)i + 0.5f) * rcp);
                            S.ring[(wpos + i) & RMASK] = S.ring[(wpos - dist + (i - q * dist)) & RMASK];
                        }
                    } else {
                        // beyond the ring: those bytes left for memory at least RING - FLUSH - 516 bytes ago (what is still pending
                        // is closer than that); the wavefront's own stores are waited for, the loads bypass its L1
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        for (uint32_t i = lane; i < len; i += 64)
                            S.ring[(wpos + i) & RMASK] = __builtin_nontemporal_load(dst + (wpos - dist + i));      // (dist > len here)
                    }
                    __builtin_amdgcn_wave_barrier();
                    wpos += len;
                }
                if (wpos - fpos >= (uint32_t)FLUSH) { __builtin_amdgcn_wave_barrier(); flush(FLUSH); }
            }
            if (!err && b.taken > b.n_dwords + 2) err = INF_INPUT_END;        // (the reader runs at most two dwords ahead of what it used)
        }
        __builtin_amdgcn_wave_barrier();
        if (!err) {
            if (wpos != B.out_len) err = INF_BAD_SIZE;
            else {
                while (wpos - fpos >= (uint32_t)FLUSH) flush(FLUSH);
                if (wpos > fpos) flush(wpos - fpos);
                if (crc != B.crc) err = INF_BAD_CRC;
            }
        }
    } else if (B.crc != 0) err = INF_BAD_CRC;
    if (lane == 0) verdict[bi] = (uint32_t)err;
}


// =====================================================================================================================================
// Round 5: the symbol loop LANE-PARALLEL.  The form above decodes one symbol per trip through a chain of dependent LDS round trips
// (table look-up -> readfirstlane -> extra bits -> distance look-up -> copy: ~800 clocks per symbol, 3 000 symbols per 64 KB block)
// with 63 lanes watching.  Here every lane decodes the symbol that WOULD start at its own bit offset -- lane l at P + l, 64 offsets
// per window -- in one pass of the tables (literal / length entry, extra bits, distance entry, extra bits: plain per-lane
// arithmetic on a 64-bit window of the stream, two LDS gathers for the whole wavefront); the true chain is then picked out by
// hopping from lane to lane with v_readlane (cur += consumed[cur]: ~6 symbols per 64 bits of a BAM stream), which also hands every
// symbol on the chain its output offset.  The literals of a window leave in ONE byte store, the matches are copied one after the
// other by the lanes (a match may read what an earlier symbol of its window wrote: LDS operations of a wavefront execute in order).
// Codes longer than the tables' index bits (rare symbols by construction) and everything else that is not the common case are
// marked SLOW by the lane and decoded by the wave-uniform path if -- and only if -- the chain lands on them.
// The compressed stream reaches the lanes through a 512-byte LDS ring (two 256-byte chunks; the next chunk waits in a register per
// lane); headers (block type, code lengths) are read with the same window function at lane offset 0.
// =====================================================================================================================================
#ifndef HGX_INF_TMAX
#define HGX_INF_TMAX 1024
#endif
constexpr int T_MAX = HGX_INF_TMAX;                              // most bytes one window may produce (a 64-bit window of 2-bit codes could ask for 8 KB)
constexpr uint32_t NEAR_MAX = RING - T_MAX - 320;        // matches up to this distance are copied inside the ring, farther ones from memory

#ifndef HGX_INF_SUBPOOL
#define HGX_INF_SUBPOOL 256
#endif
constexpr uint32_t SUB_CAP = HGX_INF_SUBPOOL;
static_assert((SUB_CAP & (SUB_CAP - 1u)) == 0u && SUB_CAP <= 32768u, "the pool is indexed under a mask; offsets live in 16 bits");
struct HuffLdsW {                        // (entries as huff_entry<1> / <2> make them)
    uint32_t pt[1 << LIT_P];
    uint16_t sorted[288];
    uint16_t count[16], first[16], offs[16];
};
struct DistLdsW {
    uint32_t pt[1 << DIST_P];
    uint16_t sorted[32];
    uint16_t count[16], first[16], offs[16];
};
struct InfLdsW {
    unsigned char ring[RING];
    HuffLdsW lit;
    DistLdsW dist;
    union {
        unsigned char lens[384];         // [0, 19): the code-length code; [32, 32 + n_lit + n_dist): the two codes' lengths
        uint32_t crc_piece[64];          // (only the last, partial flush of a block uses it: the lengths are history by then)
    };
    uint32_t in_ring[128];
    uint32_t sub[HGX_INF_SUBPOOL];       // second-level tables of both codes (huff_sub_tables)
};

__device__ uint32_t g_crc_tab[4][256];                   // slice-by-4 tables of CRC-32 (filled once per device by the host)
// [k][v][lane]: a CRC whose k-th nibble is v, advanced by the (63 - lane) pieces of zero bytes that follow this lane's piece inside a
// FLUSH -- eight loads and XORs put a lane's piece CRC where the end of the flush is, and the XOR over the lanes is the CRC of the
// 2 KB (the CRC's zero operator is linear: no tree of six matrix products as in the first form)
__device__ uint32_t g_crc_lane_nib[8][16][64];
static_assert(PIECE == 32, "crc32_words8 and g_crc_lane_nib are laid out for 32-byte pieces");

struct Win {
    const uint32_t *base;
    uint32_t n_dwords;
    uint32_t nxt;                    // this lane's dword of chunk staged_hi + 1
    int staged_hi;                   // highest 256-byte chunk present in the LDS ring (it holds staged_hi - 1 and staged_hi)
    uint32_t P;                      // bit offset of the next unread bit, counted from base[0]
};
__device__ __forceinline__ uint32_t win_chunk(const Win &w, int c, int lane) {
    const uint32_t at = (uint32_t)c * 64u;
    return at < w.n_dwords + 192u ? w.base[(size_t)at + lane] : 0u;       // (beyond the block + padding: zeros, the decode ends in an error)
}
__device__ __forceinline__ void win_init(Win &w, uint32_t *in_ring, const unsigned char *in, uint32_t in_len, int lane) {
    const uintptr_t a = (uintptr_t)in;
    w.base = (const uint32_t *)(a & ~(uintptr_t)3);
    const uint32_t skip = (uint32_t)(a & 3);
    w.n_dwords = (skip + in_len + 3) / 4;
    in_ring[lane] = win_chunk(w, 0, lane);
    in_ring[64 + lane] = win_chunk(w, 1, lane);
    w.nxt = win_chunk(w, 2, lane);
    w.staged_hi = 1;
    w.P = 8 * skip;
}
// chunks (P >> 11) and the one after it are in the ring (a window reaches at most 127 bits beyond P)
__device__ __forceinline__ void win_ensure(Win &w, uint32_t *in_ring, int lane) {
    const int c = (int)(w.P >> 11);
    while (w.staged_hi < c + 1) {
        w.staged_hi += 1;
        in_ring[(w.staged_hi & 1) * 64 + lane] = w.nxt;
        w.nxt = win_chunk(w, w.staged_hi + 1, lane);
    }
}
// 64 bits of the stream from bit offset `at` (bit 0 = the first)
__device__ __forceinline__ uint64_t win_bits(const uint32_t *in_ring, uint32_t at) {
    const uint32_t d = at >> 5;
    const uint32_t w0 = in_ring[d & 127u], w1 = in_ring[(d + 1) & 127u], w2 = in_ring[(d + 2) & 127u];
    // (v_alignbit_b32: the low dword of {hi, lo} >> (shift & 31) -- a shift of 0 hands back `lo`, no special case)
    return (uint64_t)__builtin_amdgcn_alignbit(w1, w0, at) | ((uint64_t)__builtin_amdgcn_alignbit(w2, w1, at) << 32);
}
__device__ __forceinline__ uint64_t win_bits_uniform(const uint32_t *in_ring, uint32_t at) {
    const uint64_t v = win_bits(in_ring, at);
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) |
           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}
// a wave-uniform reader on top of the window (headers, stored blocks, slow symbols): 64 bits at a time out of the ring
struct UBits { uint64_t buf; int cnt; };
__device__ __forceinline__ void ub_fill(UBits &u, const Win &w, const uint32_t *in_ring) { u.buf = win_bits_uniform(in_ring, w.P); u.cnt = 64; }
__device__ __forceinline__ uint32_t ub_take(UBits &u, Win &w, uint32_t *in_ring, int n, int lane) {     // n <= 16
    if (u.cnt < n) { win_ensure(w, in_ring, lane); ub_fill(u, w, in_ring); }
    const uint32_t v = (uint32_t)(u.buf & ((1ull << n) - 1ull));
    u.buf >>= n; u.cnt -= n; w.P += (uint32_t)n;
    return v;
}

// inclusive prefix sum over the 64 lanes through the register file: row_shr 1 / 2 / 4 / 8 inside the rows of 16, then lane 15 of a row to
// the next row (row_bcast:15, rows 1 and 3) and lane 31 to the upper half (row_bcast:31, rows 2 and 3)
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, true);
    return v;
}

// XOR over the 64 lanes, the same way: lane 63 ends up with it
__device__ __forceinline__ uint32_t wave_xor_to_lane63(uint32_t v) {
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, true);
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, true);
    return v;
}

// SECOND-LEVEL TABLES for the codes longer than the P index bits of H.pt, so that a lane can decode them by itself (round 5, late: on
// BAM streams a fifth of the windows ended on such a symbol and went through the wave-uniform path).  The codes that share their
// first P bits share a table of 2^b entries in `pool` (b = their longest length - P), indexed by the stream's next b bits; the root
// entry of such a prefix becomes K_SUB | b << 4 | offset << 16.  Canonical codes in (length, symbol) order -- H.sorted -- have the
// long codes at the end with their prefixes in runs: one pass marks every prefix with its longest length (LDS max), one hands out
// the offsets (a scan over the runs' heads), one fills.  A code whose table does not fit the pool keeps an empty root entry: the
// wave-uniform path takes it, as it takes every long code in the first form.
template <class T, int P, int MODE>
__device__ void huff_sub_tables(T &H, const unsigned char *lens, int lane, uint32_t *pool, uint32_t &used, uint32_t cap) {      // cap <= SUB_CAP entries of `pool` are handed out
    __builtin_amdgcn_wave_barrier();
    uint32_t j0 = 0, n_long = 0;
#pragma unroll
    for (int v = 1; v < 16; ++v) { if (v <= P) j0 += H.count[v]; else n_long += H.count[v]; }
    j0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)j0);
    n_long = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_long);
    if (n_long == 0) return;
    const uint32_t end = j0 + n_long;
    auto code_of = [&](uint32_t j, uint32_t &sym, uint32_t &L, uint32_t &rev) {       // the j-th code in canonical order, bit-reversed (stream order)
        sym = H.sorted[j];
        L = lens[sym];
        rev = __brev((uint32_t)H.first[L] + (j - (uint32_t)H.offs[L])) >> (32 - L);
    };
    for (uint32_t j = j0 + lane; j < end; j += 64) {
        uint32_t sym, L, rev;
        code_of(j, sym, L, rev);
        atomicMax(&H.pt[rev & ((1u << P) - 1u)], L - (uint32_t)P);
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t carry = used, r_last = 0xFFFFFFFFu;
    for (uint32_t jb = j0; jb < end; jb += 64) {
        const uint32_t j = jb + lane;
        const bool valid = j < end;
        uint32_t sym = 0, L = 0, rev = 0;
        if (valid) code_of(j, sym, L, rev);
        const uint32_t r = valid ? rev & ((1u << P) - 1u) : 0xFFFFFFFEu;
        const uint32_t up = (uint32_t)__shfl_up((int)r, 1, 64);
        const uint32_t r_prev = lane == 0 ? r_last : up;
        const bool head = valid && r != r_prev;
        const uint32_t b = head ? H.pt[r] : 0u;
        const uint32_t size = head ? 1u << b : 0u;
        const uint32_t incl = wave_incl_scan_u32(size);
        const uint32_t off = carry + incl - size;
        if (head) H.pt[r] = off + size <= cap ? (((uint32_t)K_SUB << 8) | (b << 4) | (off << 16)) : 0u;
        carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        r_last = (uint32_t)__builtin_amdgcn_readlane((int)r, 63);
    }
    const uint32_t used_new = carry < cap ? carry : cap;
    for (uint32_t i = used + lane; i < used_new; i += 64) pool[i] = 0u;
    __builtin_amdgcn_wave_barrier();
    for (uint32_t j = j0 + lane; j < end; j += 64) {
        uint32_t sym, L, rev;
        code_of(j, sym, L, rev);
        const uint32_t ptr = H.pt[rev & ((1u << P) - 1u)];
        if ((ptr & 0x70Fu) == ((uint32_t)K_SUB << 8)) {
            const uint32_t b = (ptr >> 4) & 15u, o = ptr >> 16;
            const uint32_t e = huff_entry<MODE>(sym, L, false);
            for (uint32_t k = rev >> P; k < (1u << b); k += 1u << (L - (uint32_t)P)) pool[o + k] = e;
        }
    }
    used = used_new;
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint32_t crc32_words8(const uint32_t (&wd)[8]) {
    uint32_t c = 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        c ^= wd[k];
        c = g_crc_tab[3][c & 255u] ^ g_crc_tab[2][(c >> 8) & 255u] ^ g_crc_tab[1][(c >> 16) & 255u] ^ g_crc_tab[0][c >> 24];
    }
    return ~c;
}

// PROF: clock64() laps per phase, summed per block into prof[block][8] (test switch front=inflate_prof: a measuring aid)
enum { PH_HEADER = 0, PH_DECODE, PH_WALK, PH_LIT, PH_NEAR, PH_FAR, PH_FLUSH, PH_SLOW };
template <bool PROF, bool SMALL_POOL = false>
__global__ void __launch_bounds__(64) k_bgzf_inflate_w(const unsigned char *__restrict__ in, const BlockDesc *__restrict__ blocks, int n_blocks,
                                                       unsigned char *__restrict__ out, CrcOp op, uint32_t *__restrict__ verdict,
                                                       unsigned long long *__restrict__ prof) {
    constexpr uint32_t sub_cap = SMALL_POOL ? 32u : SUB_CAP;        // (an instantiation of its own for the tests: a run-time cap cost the product kernel 2-4 %)
    extern __shared__ unsigned char inf_lds_raw[];
    InfLdsW &S = *reinterpret_cast<InfLdsW *>(inf_lds_raw);
    const int lane = threadIdx.x;
    const int bi = blockIdx.x;
    if (bi >= n_blocks) return;
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last = PROF ? clock64() : 0ull, n_win = 0, n_far = 0, n_near = 0, n_slow = 0;
    auto lap = [&](int k) { if (PROF) { const unsigned long long t = clock64(); ph[k] += t - t_last; t_last = t; } };
    const BlockDesc B = blocks[bi];
    unsigned char *dst = out + B.out_off;
    uint32_t wpos = 0, fpos = 0, crc = 0;
    int err = INF_OK;
    // a full FLUSH leaves the ring: every lane stores its 32-byte piece (two 16-byte stores) and takes the piece's CRC-32 from the
    // same eight registers (slice-by-4 tables, L1-resident); the pieces are joined through g_crc_lane_nib
    auto flush_full = [&]() {
        const uint32_t p0 = (fpos + (uint32_t)PIECE * lane) & RMASK;            // (32-byte aligned: fpos is a multiple of FLUSH)
        uint32_t wd[8];
        const uint4 a = *reinterpret_cast<const uint4 *>(&S.ring[p0]);
        const uint4 b2 = *reinterpret_cast<const uint4 *>(&S.ring[p0 + 16]);
        wd[0] = a.x; wd[1] = a.y; wd[2] = a.z; wd[3] = a.w; wd[4] = b2.x; wd[5] = b2.y; wd[6] = b2.z; wd[7] = b2.w;
        unsigned char *g = dst + fpos + (uint32_t)PIECE * lane;
        __builtin_memcpy(g, &a, 16);
        __builtin_memcpy(g + 16, &b2, 16);
        const uint32_t mine = crc32_words8(wd);
        uint32_t adv = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) adv ^= g_crc_lane_nib[k][(mine >> (4 * k)) & 15u][lane];
        adv = wave_xor_to_lane63(adv);
        crc = crc_advance(op, 6, crc) ^ (uint32_t)__builtin_amdgcn_readlane((int)adv, 63);
        fpos += (uint32_t)FLUSH;
    };
    auto flush_rest = [&](uint32_t n) {        // the last n < FLUSH bytes
        for (uint32_t i = lane; i < n; i += 64) dst[fpos + i] = S.ring[(fpos + i) & RMASK];
        const uint32_t n_piece = (n + PIECE - 1) / PIECE;
        uint32_t mine = 0;
        if ((uint32_t)lane < n_piece) {
            const uint32_t p0 = fpos + (uint32_t)PIECE * lane;
            const int len = (int)min((uint32_t)PIECE, n - (uint32_t)PIECE * lane);
            mine = crc_bytes(S.ring, p0, len);
        }
        S.crc_piece[lane] = mine;
        __builtin_amdgcn_wave_barrier();
        for (uint32_t p = 0; p < n_piece; ++p) {
            const uint32_t len = min((uint32_t)PIECE, n - (uint32_t)PIECE * p);
            if (len == (uint32_t)PIECE) crc = crc_advance(op, 0, crc);
            else for (uint32_t k = 0; k < 8 * len; ++k) crc = (crc >> 1) ^ (0xEDB88320u & (0u - (crc & 1u)));
            crc ^= S.crc_piece[p];
        }
        __builtin_amdgcn_wave_barrier();
        fpos += n;
    };
    // `len` bytes at ring position `t` copied from `dist` bytes back (t, len, dist wave-uniform; the source lies before the target)
    auto copy_match = [&](uint32_t t, uint32_t len, uint32_t dist) {
        if (dist >= len && dist <= NEAR_MAX) {
            for (uint32_t i = lane; i < len; i += 64) S.ring[(t + i) & RMASK] = S.ring[(t - dist + i) & RMASK];
        } else if (dist == 1) {
            const unsigned char v = S.ring[(t - 1) & RMASK];
            for (uint32_t i = lane; i < len; i += 64) S.ring[(t + i) & RMASK] = v;
        } else if (dist < len) {
            const float rcp = __frcp_rn((float)dist);           // i mod dist without a division (see the first form)
            for (uint32_t i = lane; i < len; i += 64) {
                const uint32_t q = (uint32_t)(((float)i + 0.5f) * rcp);
                S.ring[(t + i) & RMASK] = S.ring[(t - dist + (i - q * dist)) & RMASK];
            }
        } else {
            // beyond the near part of the ring: those bytes left for memory long ago (NEAR_MAX > FLUSH + T_MAX + 258)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (uint32_t i = lane; i < len; i += 64) S.ring[(t + i) & RMASK] = __builtin_nontemporal_load(dst + (t - dist + i));
        }
    };
    if (B.out_len > 0) {
        Win w;
        win_init(w, S.in_ring, in + B.in_off, B.in_len, lane);
        const uint32_t end_bits = (uint32_t)(((uintptr_t)(in + B.in_off) & 3) + B.in_len) * 8u;
        bool last = false;
        while (!last && !err) {
            UBits u;
            win_ensure(w, S.in_ring, lane);
            ub_fill(u, w, S.in_ring);
            last = ub_take(u, w, S.in_ring, 1, lane) != 0;
            const uint32_t type = ub_take(u, w, S.in_ring, 2, lane);
            if (type == 0) {                                        // stored: the bytes straight from the input
                w.P = (w.P + 7u) & ~7u;
                win_ensure(w, S.in_ring, lane);
                ub_fill(u, w, S.in_ring);
                const uint32_t len = ub_take(u, w, S.in_ring, 16, lane), nlen = ub_take(u, w, S.in_ring, 16, lane);
                if ((len ^ 0xFFFFu) != nlen) { err = INF_BAD_STORED; break; }
                if (wpos + len > B.out_len) { err = INF_OVERRUN; break; }
                if (w.P + 8u * len > end_bits) { err = INF_INPUT_END; break; }
                const unsigned char *src = (const unsigned char *)w.base + (w.P >> 3);
                for (uint32_t done = 0; done < len;) {
                    const uint32_t n = min(len - done, (uint32_t)T_MAX);
                    for (uint32_t i = lane; i < n; i += 64) S.ring[(wpos + i) & RMASK] = src[done + i];
                    wpos += n;
                    done += n;
                    __builtin_amdgcn_wave_barrier();
                    while (wpos - fpos >= (uint32_t)FLUSH) flush_full();
                }
                w.P += 8u * len;
                continue;
            }
            if (type == 3) { err = INF_BAD_BLOCK_TYPE; break; }
            if (type == 1) {                                        // fixed codes (RFC 1951, 3.2.6)
                for (int s2 = lane; s2 < 288; s2 += 64) S.lens[s2] = (unsigned char)(s2 < 144 ? 8 : s2 < 256 ? 9 : s2 < 280 ? 7 : 8);
                if (lane < 32) S.lens[288 + lane] = 5;
                __builtin_amdgcn_wave_barrier();
                if (!huff_build<HuffLdsW, LIT_P, 1>(S.lit, S.lens, 288, lane, false)) { err = INF_BAD_LENGTHS; break; }
                if (!huff_build<DistLdsW, DIST_P, 2>(S.dist, S.lens + 288, 32, lane, true)) { err = INF_BAD_LENGTHS; break; }
            } else {                                                // dynamic codes (3.2.7)
                const int n_lit = (int)ub_take(u, w, S.in_ring, 5, lane) + 257;
                const int n_dist = (int)ub_take(u, w, S.in_ring, 5, lane) + 1;
                const int n_cl = (int)ub_take(u, w, S.in_ring, 4, lane) + 4;
                if (n_lit > 286 || n_dist > 30) { err = INF_BAD_LENGTHS; break; }
                if (lane < 19) S.lens[lane] = 0;
                __builtin_amdgcn_wave_barrier();
                for (int i = 0; i < n_cl; ++i) {
                    const uint32_t v = ub_take(u, w, S.in_ring, 3, lane);
                    if (lane == 0) S.lens[c_cl_order[i]] = (unsigned char)v;
                }
                __builtin_amdgcn_wave_barrier();
                if (!huff_build<DistLdsW, DIST_P, 0>(S.dist, S.lens, 19, lane, false)) { err = INF_BAD_LENGTHS; break; }
                int at = 0, prev = 0;
                const int total = n_lit + n_dist;
                while (at < total && !err) {
                    if (u.cnt < 16) { win_ensure(w, S.in_ring, lane); ub_fill(u, w, S.in_ring); }
                    const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.dist.pt[(uint32_t)u.buf & ((1u << DIST_P) - 1u)]);
                    if (!(e & 15u)) { err = INF_BAD_CODE; break; }           // (the code-length code has at most 7 bits: always in the table)
                    u.buf >>= (e & 15u); u.cnt -= (int)(e & 15u); w.P += (e & 15u);
                    const int sym = (int)(e >> 4);
                    int rep = 1, val = sym;
                    if (sym == 16) { if (at == 0) { err = INF_BAD_LENGTHS; break; } rep = 3 + (int)ub_take(u, w, S.in_ring, 2, lane); val = prev; }
                    else if (sym == 17) { rep = 3 + (int)ub_take(u, w, S.in_ring, 3, lane); val = 0; }
                    else if (sym == 18) { rep = 11 + (int)ub_take(u, w, S.in_ring, 7, lane); val = 0; }
                    if (at + rep > total) { err = INF_BAD_LENGTHS; break; }
                    if (lane < rep) S.lens[32 + at + lane] = (unsigned char)val;
                    if (lane + 64 < rep) S.lens[32 + at + lane + 64] = (unsigned char)val;
                    if (lane + 128 < rep) S.lens[32 + at + lane + 128] = (unsigned char)val;
                    at += rep;
                    prev = val;
                }
                if (err) break;
                __builtin_amdgcn_wave_barrier();
                if (S.lens[32 + 256] == 0) { err = INF_BAD_LENGTHS; break; }
                if (!huff_build<HuffLdsW, LIT_P, 1>(S.lit, S.lens + 32, n_lit, lane, false)) {
                    // an incomplete literal/length code is legal only when it is ONE code of one bit (zlib's inflate_table)
                    if (!huff_incomplete_ok(S.lens + 32, n_lit, lane, false) || !huff_build<HuffLdsW, LIT_P, 1>(S.lit, S.lens + 32, n_lit, lane, true)) { err = INF_BAD_LENGTHS; break; }
                }
                if (!huff_build<DistLdsW, DIST_P, 2>(S.dist, S.lens + 32 + n_lit, n_dist, lane, false)) {
                    if (!huff_incomplete_ok(S.lens + 32 + n_lit, n_dist, lane, true) || !huff_build<DistLdsW, DIST_P, 2>(S.dist, S.lens + 32 + n_lit, n_dist, lane, true)) { err = INF_BAD_LENGTHS; break; }
                }
                uint32_t sub_used = 0;
                huff_sub_tables<HuffLdsW, LIT_P, 1>(S.lit, S.lens + 32, lane, S.sub, sub_used, sub_cap);
                huff_sub_tables<DistLdsW, DIST_P, 2>(S.dist, S.lens + 32 + n_lit, lane, S.sub, sub_used, sub_cap);
            }
            // ---- the symbols of the block, a window of 64 bit offsets at a time ----------------------------------------------
            bool eob = false;
            lap(PH_HEADER);
            while (!eob && !err) {
                if (PROF) n_win++;
                win_ensure(w, S.in_ring, lane);
                const uint64_t bits = win_bits(S.in_ring, w.P + (uint32_t)lane);
                // this lane's symbol, as if one started at its offset.  Straight-line code: the table entries carry what a symbol means
                // (huff_entry<1> / <2>), every lane looks a distance up whether it holds a match or not, and the three shifts of the
                // window are v_alignbit_b32 (a code has <= 15 bits, a length <= 5 extra bits: after them the distance code and its
                // <= 13 extra bits fit the low dword)
                uint32_t kind, used, olen, val, mdist;
                {
                    const uint32_t blo = (uint32_t)bits, bhi = (uint32_t)(bits >> 32);
                    uint32_t e = S.lit.pt[blo & ((1u << LIT_P) - 1u)];
                    {   // a code longer than the index bits: its second-level table (huff_sub_tables), by the bits that follow
                        const bool sub = (e & 0x70Fu) == ((uint32_t)K_SUB << 8);
                        if (__builtin_amdgcn_ballot_w64(sub)) {
                            const uint32_t e2 = S.sub[((e >> 16) + __builtin_amdgcn_ubfe(blo, (uint32_t)LIT_P, (e >> 4) & 15u)) & (SUB_CAP - 1u)];
                            e = sub ? e2 : e;
                        }
                    }
                    const uint32_t nb = e & 15u, le = (e >> 4) & 15u, k0 = (e >> 8) & 7u, base = e >> 16;
                    const uint32_t xlo = __builtin_amdgcn_alignbit(bhi, blo, nb), xhi = bhi >> nb;
                    const uint32_t len = base + __builtin_amdgcn_ubfe(xlo, 0u, le);
                    const uint32_t y = __builtin_amdgcn_alignbit(xhi, xlo, le);
                    uint32_t ed = S.dist.pt[y & ((1u << DIST_P) - 1u)];
                    {
                        const bool sub = (ed & 0x70Fu) == ((uint32_t)K_SUB << 8);
                        if (__builtin_amdgcn_ballot_w64(sub)) {
                            const uint32_t e2 = S.sub[((ed >> 16) + __builtin_amdgcn_ubfe(y, (uint32_t)DIST_P, (ed >> 4) & 15u)) & (SUB_CAP - 1u)];
                            ed = sub ? e2 : ed;
                        }
                    }
                    const uint32_t dn = ed & 15u, de = (ed >> 4) & 15u;
                    const bool is_m = k0 == (uint32_t)K_MATCH;                         // (an empty entry is all zeros: nb = 0, k0 = K_LIT)
                    mdist = (ed >> 16) + __builtin_amdgcn_ubfe(y >> dn, 0u, de);
                    kind = nb ? k0 : (uint32_t)K_SLOW;
                    if (is_m) kind = dn == 0u ? (uint32_t)K_SLOW : (ed & 0x100u) ? (uint32_t)K_BAD : (uint32_t)K_MATCH;
                    used = is_m ? nb + le + dn + de : nb;
                    olen = is_m ? len : (kind == (uint32_t)K_LIT ? 1u : 0u);
                    val = base & 255u;
                }
                const uint32_t r0 = used | (kind << 6) | (olen << 9) | (val << 18);     // used <= 48, kind < 8, olen <= 258, val < 256
                if (PROF) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (__builtin_amdgcn_readfirstlane((int)r0) == -1) err = INF_BAD_CODE; }
                lap(PH_DECODE);
                // ---- the chain: from offset 0, hop by the bits each symbol uses.  The scalar loop only collects the positions (a
                // v_readlane, a bit set and an add per symbol; it ends ON the first symbol that is not a plain literal or match: its hop
                // is 0); what the chain means -- output offsets, the T_MAX cut, which lanes are literals and which matches -- is worked
                // out by all lanes at once from the position mask ----
                const uint32_t hop = kind <= K_MATCH ? used : 0u;
                uint32_t cur = 0;
                uint64_t m_chain = 0;
                while (cur < 64u) {
                    m_chain |= 1ull << cur;
                    const uint32_t h = (uint32_t)__builtin_amdgcn_readlane((int)hop, (int)cur);
                    if (h == 0u) break;
                    cur += h;
                }
                // cur < 64: the chain stopped ON lane cur (end of block, a slow symbol, a bad code); else it left the window at bit cur
                uint64_t m_norm = cur < 64u ? m_chain & ~(1ull << cur) : m_chain;
                uint32_t incl = wave_incl_scan_u32(((m_norm >> lane) & 1ull) ? olen : 0u);
                const uint64_t over = __ballot(((m_norm >> lane) & 1ull) && incl > (uint32_t)T_MAX);
                bool slow = false;
                if (over) {                                                 // the window would make more than T_MAX bytes: the rest starts the next one
                    const uint32_t cut = (uint32_t)__builtin_ctzll(over);
                    m_norm &= (1ull << cut) - 1ull;
                    cur = cut;
                    incl = wave_incl_scan_u32(((m_norm >> lane) & 1ull) ? olen : 0u);
                } else if (cur < 64u) {
                    const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)r0, (int)cur);
                    const uint32_t k = (r >> 6) & 7u;
                    if (k == K_SLOW) slow = true;
                    else if (k == K_BAD) err = INF_BAD_CODE;
                    else { eob = true; cur += r & 63u; }
                }
                const bool on_chain = (m_norm >> lane) & 1ull;
                const uint32_t my_off = incl - (on_chain ? olen : 0u);
                const uint32_t run = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                uint64_t m_match = __ballot(on_chain && kind == K_MATCH);
                const uint64_t m_lit = m_norm & ~m_match;
                if (err) break;
                if (wpos + run > B.out_len) { err = INF_OVERRUN; break; }
                lap(PH_WALK);
                // ---- output: the literals at once, the matches in order ----
                if ((m_lit >> lane) & 1ull) S.ring[(wpos + my_off) & RMASK] = (unsigned char)val;
                lap(PH_LIT);
                // Matches whose whole source lies BEFORE this window's output (offset + length <= distance) and that fit one round of
                // the lanes do not depend on anything the window writes: up to four at a time, all their reads issued before the first
                // write -- ring reads for the near ones, memory reads for the far ones -- so that one round trip serves the batch.
                // (A bad distance shows on the first match of a block only through `d > t`; those go the ordinary way below.)
                {
                    uint64_t m_ind = __ballot(on_chain && kind == K_MATCH && olen <= 64u && my_off + olen <= mdist && mdist <= wpos + my_off);
                    m_match &= ~m_ind;
                    while (m_ind) {
                        uint32_t tt[4], ll[4], vv[4];
                        int nb = 0;
                        bool any_far = false;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            tt[k] = 0; ll[k] = 0; vv[k] = 0;
                            if (m_ind) {
                                const int ml = __builtin_ctzll(m_ind);
                                m_ind &= m_ind - 1;
                                const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)r0, ml);
                                const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)mdist, ml);
                                tt[k] = wpos + (uint32_t)__builtin_amdgcn_readlane((int)my_off, ml);
                                ll[k] = (r >> 9) & 511u;
                                vv[k] = d;
                                any_far = any_far || d > NEAR_MAX;
                                nb = k + 1;
                            }
                        }
                        if (any_far) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (this wavefront's own flushes have landed)
                        uint32_t by[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            by[k] = 0;
                            if (k < nb && (uint32_t)lane < ll[k])
                                by[k] = vv[k] <= NEAR_MAX ? (uint32_t)S.ring[(tt[k] - vv[k] + lane) & RMASK]
                                                          : (uint32_t)__builtin_nontemporal_load(dst + (tt[k] - vv[k] + lane));
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (k < nb && (uint32_t)lane < ll[k]) S.ring[(tt[k] + lane) & RMASK] = (unsigned char)by[k];
                        if (PROF) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); n_near += nb; lap(PH_NEAR); }
                    }
                }
                while (m_match) {
                    const int ml = __builtin_ctzll(m_match);
                    m_match &= m_match - 1;
                    const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)r0, ml);
                    const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)mdist, ml);
                    const uint32_t t = wpos + (uint32_t)__builtin_amdgcn_readlane((int)my_off, ml);
                    if (d > t) { err = INF_BAD_DIST; break; }
                    copy_match(t, (r >> 9) & 511u, d);
                    if (PROF) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        const bool far_ = !(d >= ((r >> 9) & 511u) && d <= NEAR_MAX) && d != 1 && !(d < ((r >> 9) & 511u));
                        if (far_) n_far++; else n_near++;
                        lap(far_ ? PH_FAR : PH_NEAR);
                    }
                }
                if (err) break;
                wpos += run;
                w.P += cur;
                if (slow) {
                    if (PROF) n_slow++;
                    // a symbol outside the tables' fast cases -- a code longer than the index bits, mostly: the wave-uniform path, one symbol
                    win_ensure(w, S.in_ring, lane);
                    ub_fill(u, w, S.in_ring);
                    uint32_t k2, le, base;                               // as the fat entries say it, or worked out from the symbol
                    {
                        const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.lit.pt[(uint32_t)u.buf & ((1u << LIT_P) - 1u)]);
                        if (e & 15u) { k2 = (e >> 8) & 7u; le = (e >> 4) & 15u; base = e >> 16; u.buf >>= (e & 15u); u.cnt -= (int)(e & 15u); w.P += (e & 15u); }
                        else {
                            int sym = -1;
                            const uint32_t rev15 = __brev((uint32_t)u.buf & 0x7FFFu) >> 17;
                            for (int L = LIT_P + 1; L <= 15; ++L) {
                                const uint32_t c = rev15 >> (15 - L);
                                const uint32_t dd = c - S.lit.first[L];
                                if (dd < S.lit.count[L]) { sym = S.lit.sorted[S.lit.offs[L] + dd]; u.buf >>= L; u.cnt -= L; w.P += (uint32_t)L; break; }
                            }
                            sym = __builtin_amdgcn_readfirstlane(sym);
                            if (sym < 0) { err = INF_BAD_CODE; break; }
                            const uint32_t f = huff_entry<1>((uint32_t)sym, 1u, false);
                            k2 = (f >> 8) & 7u; le = (f >> 4) & 15u; base = f >> 16;
                        }
                    }
                    if (k2 == (uint32_t)K_BAD) { err = INF_BAD_CODE; break; }
                    if (k2 == (uint32_t)K_LIT) {
                        if (wpos + 1 > B.out_len) { err = INF_OVERRUN; break; }
                        if (lane == 0) S.ring[wpos & RMASK] = (unsigned char)base;
                        wpos += 1;
                    } else if (k2 == (uint32_t)K_EOB) eob = true;
                    else {
                        const uint32_t len = base + ub_take(u, w, S.in_ring, (int)le, lane);
                        if (u.cnt < 32) { win_ensure(w, S.in_ring, lane); ub_fill(u, w, S.in_ring); }
                        uint32_t de, dbase;
                        {
                            const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.dist.pt[(uint32_t)u.buf & ((1u << DIST_P) - 1u)]);
                            if (e & 15u) {
                                if (e & 0x100u) { err = INF_BAD_CODE; break; }
                                de = (e >> 4) & 15u; dbase = e >> 16; u.buf >>= (e & 15u); u.cnt -= (int)(e & 15u); w.P += (e & 15u);
                            } else {
                                int ds = -1;
                                const uint32_t rev15 = __brev((uint32_t)u.buf & 0x7FFFu) >> 17;
                                for (int L = DIST_P + 1; L <= 15; ++L) {
                                    const uint32_t c = rev15 >> (15 - L);
                                    const uint32_t dd = c - S.dist.first[L];
                                    if (dd < S.dist.count[L]) { ds = S.dist.sorted[S.dist.offs[L] + dd]; u.buf >>= L; u.cnt -= L; w.P += (uint32_t)L; break; }
                                }
                                ds = __builtin_amdgcn_readfirstlane(ds);
                                if (ds < 0 || ds >= 30) { err = INF_BAD_CODE; break; }
                                const uint32_t f = huff_entry<2>((uint32_t)ds, 1u, false);
                                de = (f >> 4) & 15u; dbase = f >> 16;
                            }
                        }
                        const uint32_t dist = dbase + ub_take(u, w, S.in_ring, (int)de, lane);
                        if (dist > wpos) { err = INF_BAD_DIST; break; }
                        if (wpos + len > B.out_len) { err = INF_OVERRUN; break; }
                        copy_match(wpos, len, dist);
                        wpos += len;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                lap(PH_SLOW);
                while (wpos - fpos >= (uint32_t)FLUSH) flush_full();
                if (PROF) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                lap(PH_FLUSH);
            }
            if (!err && w.P > end_bits) err = INF_INPUT_END;
        }
        __builtin_amdgcn_wave_barrier();
        if (!err) {
            if (wpos != B.out_len) err = INF_BAD_SIZE;
            else {
                while (wpos - fpos >= (uint32_t)FLUSH) flush_full();
                if (wpos > fpos) flush_rest(wpos - fpos);
                if (crc != B.crc) err = INF_BAD_CRC;
            }
        }
    } else if (B.crc != 0) err = INF_BAD_CRC;
    if (lane == 0) verdict[bi] = (uint32_t)err;
    if (PROF && lane == 0) {
        for (int k = 0; k < 8; ++k) prof[(size_t)bi * 12 + k] = ph[k];
        prof[(size_t)bi * 12 + 8] = n_win; prof[(size_t)bi * 12 + 9] = n_near; prof[(size_t)bi * 12 + 10] = n_far; prof[(size_t)bi * 12 + 11] = B.out_len | (n_slow << 32);
    }
}


int inflate_w_setup() {                  // once per device: the CRC-32 slice tables in device memory, the kernel's LDS size
    uint32_t tab[4][256];
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
        tab[0][i] = c;
    }
    for (int t = 1; t < 4; ++t)
        for (uint32_t i = 0; i < 256; ++i) tab[t][i] = (tab[t - 1][i] >> 8) ^ tab[0][tab[t - 1][i] & 255u];
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_crc_tab), tab, sizeof(tab)));
    {
        // columns of the zero operator for (63 - lane) pieces: lane 63's is the identity, lane l's = one more piece than lane l + 1's
        std::vector<uint32_t> nib((size_t)8 * 16 * 64);
        uint32_t col[32];
        for (int i = 0; i < 32; ++i) col[i] = 1u << i;
        for (int lane = 63; lane >= 0; --lane) {
            for (int k = 0; k < 8; ++k)
                for (uint32_t v = 0; v < 16; ++v) {
                    uint32_t x = 0;
                    for (int b = 0; b < 4; ++b) if (v >> b & 1u) x ^= col[4 * k + b];
                    nib[((size_t)k * 16 + v) * 64 + lane] = x;
                }
            for (int i = 0; i < 32; ++i) {
                uint32_t c = col[i];
                for (int k = 0; k < 8 * PIECE; ++k) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
                col[i] = c;
            }
        }
        HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_crc_lane_nib), nib.data(), nib.size() * 4));
    }
    HIPCHK(hipFuncSetAttribute((const void *)k_bgzf_inflate_w<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(InfLdsW)));
    HIPCHK(hipFuncSetAttribute((const void *)k_bgzf_inflate_w<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(InfLdsW)));
    HIPCHK(hipFuncSetAttribute((const void *)k_bgzf_inflate_w<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(InfLdsW)));
    return HGX_OK;
}

CrcOp make_crc_op() {
    CrcOp op;
    for (int lv = 0; lv < 7; ++lv)
        for (int i = 0; i < 32; ++i) {
            uint32_t c = 1u << i;
            for (int k = 0; k < 8 * (PIECE << lv); ++k) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
            op.col[lv][i] = c;
        }
    return op;
}

}   // namespace

// the BGZF blocks of `data` (host bytes; n bytes) as descriptors: HGX_OK, or the host reader's error for a malformed container
int hgx_bgzf_scan(const unsigned char *data, size_t n, std::vector<hgx_bgzf_block> &blocks, size_t *total_out) {
    blocks.clear();
    size_t off = 0, total = 0;
    auto rd16 = [&](size_t p) { return (unsigned)data[p] | ((unsigned)data[p + 1] << 8); };
    auto rd32 = [&](size_t p) { return (uint32_t)data[p] | ((uint32_t)data[p + 1] << 8) | ((uint32_t)data[p + 2] << 16) | ((uint32_t)data[p + 3] << 24); };
    while (off < n) {
        if (off + 18 > n || data[off] != 0x1f || data[off + 1] != 0x8b || data[off + 2] != 8 || !(data[off + 3] & 4)) {
            hgx_set_error("not a BGZF block at offset %zu", off);
            return HGX_EPARSE;
        }
        const unsigned xlen = rd16(off + 10);
        if (off + 12 + xlen > n) { hgx_set_error("truncated BGZF header at offset %zu", off); return HGX_EPARSE; }
        long bsize = -1;
        for (size_t p = off + 12; p + 4 <= off + 12 + xlen;) {
            const unsigned slen = rd16(p + 2);
            if (data[p] == 66 && data[p + 1] == 67 && slen == 2 && p + 6 <= off + 12 + xlen) bsize = (long)rd16(p + 4);
            p += 4 + slen;
        }
        if (bsize < 0) { hgx_set_error("BGZF block without BC subfield at offset %zu", off); return HGX_EPARSE; }
        const size_t blen = (size_t)bsize + 1;
        if (off + blen > n || blen < 12 + xlen + 8) { hgx_set_error("truncated BGZF block at offset %zu", off); return HGX_EPARSE; }
        hgx_bgzf_block b;
        b.in_off = off + 12 + xlen;
        b.in_len = blen - 12 - xlen - 8;
        b.crc = rd32(off + blen - 8);
        b.out_len = rd32(off + blen - 4);
        b.out_off = total;
        total += b.out_len;
        blocks.push_back(b);
        off += blen;
    }
    if (total_out) *total_out = total;
    return HGX_OK;
}

// The same on several host threads.  The blocks of a BGZF file form a chain (each BSIZE leads to the next header), and every hop of
// the walk is a cache miss in bytes another core has just written: 5 007 blocks of a 20 MB BAM took 0.55 ms on one thread -- in front
// of the device inflate's launch (tools/bam_gap_timeline.py).  Here the file is cut into ranges; every range but the first FINDS a
// block start (the gzip magic with a BC subfield whose BSIZE leads to three more such headers, or to the file's end) and walks from
// there to the first block boundary at or behind its end; the ranges must link up -- a walk ends exactly where the next one began --
// or the chain is walked by one thread after all (which also words the errors).  Same descriptors, same order.
int hgx_bgzf_scan_par(const unsigned char *data, size_t n, std::vector<hgx_bgzf_block> &blocks, size_t *total_out, int n_threads) {
    const int T = (int)std::min<size_t>({(size_t)std::max(1, n_threads), (size_t)16, n / (1u << 20)});
    if (T < 2) return hgx_bgzf_scan(data, n, blocks, total_out);
    auto rd16 = [&](size_t p) { return (unsigned)data[p] | ((unsigned)data[p + 1] << 8); };
    auto rd32 = [&](size_t p) { return (uint32_t)data[p] | ((uint32_t)data[p + 1] << 8) | ((uint32_t)data[p + 2] << 16) | ((uint32_t)data[p + 3] << 24); };
    // length of the block whose header starts at `off` (0 = not a block the serial walk would take), its payload's place
    auto block_at = [&](size_t off, hgx_bgzf_block *b) -> size_t {
        if (off + 18 > n || data[off] != 0x1f || data[off + 1] != 0x8b || data[off + 2] != 8 || !(data[off + 3] & 4)) return 0;
        const unsigned xlen = rd16(off + 10);
        if (off + 12 + xlen > n) return 0;
        long bsize = -1;
        for (size_t p = off + 12; p + 4 <= off + 12 + xlen;) {
            const unsigned slen = rd16(p + 2);
            if (data[p] == 66 && data[p + 1] == 67 && slen == 2 && p + 6 <= off + 12 + xlen) bsize = (long)rd16(p + 4);
            p += 4 + slen;
        }
        if (bsize < 0) return 0;
        const size_t blen = (size_t)bsize + 1;
        if (off + blen > n || blen < 12 + xlen + 8) return 0;
        if (b) { b->in_off = off + 12 + xlen; b->in_len = blen - 12 - xlen - 8; b->crc = rd32(off + blen - 8); b->out_len = rd32(off + blen - 4); b->out_off = 0; }
        return blen;
    };
    std::vector<std::vector<hgx_bgzf_block>> part((size_t)T);
    std::vector<size_t> first((size_t)T, n), last((size_t)T, n), out_bytes((size_t)T, 0);
    std::vector<int> bad((size_t)T, 0);
    hgx_run_workers(T, [&](int r) {
        const size_t lo = n * (size_t)r / (size_t)T, hi = n * (size_t)(r + 1) / (size_t)T;
        size_t off = lo;
        if (r > 0) {
            off = n;
            for (size_t p = lo; p + 18 <= n && p < hi;) {
                const unsigned char *q = (const unsigned char *)memchr(data + p, 0x1f, hi - p);
                if (!q) break;
                p = (size_t)(q - data);
                size_t at = p;
                int hops = 0;
                for (; hops < 4 && at < n; ++hops) {
                    const size_t bl = block_at(at, nullptr);
                    if (!bl) break;
                    at += bl;
                }
                if (hops == 4 || (hops > 0 && at == n)) { off = p; break; }
                ++p;
            }
        }
        first[(size_t)r] = off;
        std::vector<hgx_bgzf_block> &mine = part[(size_t)r];
        mine.reserve((hi - lo) / 2048 + 16);
        size_t tot = 0;
        while (off < hi) {
            hgx_bgzf_block b;
            const size_t bl = block_at(off, &b);
            if (!bl) { bad[(size_t)r] = 1; break; }
            b.out_off = tot;
            tot += b.out_len;
            mine.push_back(b);
            off += bl;
        }
        last[(size_t)r] = off;
        out_bytes[(size_t)r] = tot;
    });
    bool linked = first[0] == 0;
    for (int r = 0; r < T && linked; ++r) linked = !bad[(size_t)r] && last[(size_t)r] == (r + 1 < T ? first[(size_t)r + 1] : n);
    if (!linked) return hgx_bgzf_scan(data, n, blocks, total_out);
    size_t n_blocks = 0, total = 0;
    for (auto &v : part) n_blocks += v.size();
    blocks.clear();
    blocks.reserve(n_blocks);
    for (int r = 0; r < T; ++r) {
        for (hgx_bgzf_block b : part[(size_t)r]) { b.out_off += total; blocks.push_back(b); }
        total += out_bytes[(size_t)r];
    }
    if (total_out) *total_out = total;
    return HGX_OK;
}

// test entry (hgx.h; host only, no GPU): the container walked by one thread and by `n_threads` -- *same = both gave the same verdict
// and, where they took the file, the same descriptors
extern "C" int hgx_bgzf_scan_compare(const void *bgzf, size_t n_bytes, int32_t n_threads, int64_t *n_blocks, int32_t *same) {
    ARGCHK(bgzf && n_blocks && same);
    std::vector<hgx_bgzf_block> a, b;
    size_t ta = 0, tb = 0;
    const int ra = hgx_bgzf_scan((const unsigned char *)bgzf, n_bytes, a, &ta);
    const int rb = hgx_bgzf_scan_par((const unsigned char *)bgzf, n_bytes, b, &tb, n_threads);
    *n_blocks = ra == HGX_OK ? (int64_t)a.size() : -1;
    bool eq = ra == rb;
    if (eq && ra == HGX_OK) {
        eq = ta == tb && a.size() == b.size();
        for (size_t k = 0; k < a.size() && eq; ++k)
            eq = a[k].in_off == b[k].in_off && a[k].in_len == b[k].in_len && a[k].out_off == b[k].out_off && a[k].out_len == b[k].out_len && a[k].crc == b[k].crc;
    }
    *same = eq ? 1 : 0;
    return HGX_OK;
}

// inflate `blocks` of the BGZF bytes at d_in (device; padded by >= 1 KB) into d_out (device); *bad = blocks whose verdict is not 0
// (after a stream synchronisation).  Blocks of more than 64 KB of payload, or a stream beyond 32-bit offsets: HGX_EINVAL.
// `staging` (optional): registered host memory of at least hgx_bgzf_inflate_staging_bytes(n_blocks) bytes for the block table on its
// way up and the verdicts on their way down.  A copy out of pageable memory is not queued: the runtime makes the host wait until the
// stream has drained -- the deflated file was still landing -- and only then stages it: the kernel started 0.4 ms after the file's
// last byte (tools/bam_gap_timeline.py).
size_t hgx_bgzf_inflate_staging_bytes(size_t n_blocks) { return n_blocks * (sizeof(BlockDesc) + 4) + 64; }
int hgx_bgzf_inflate_dev(const unsigned char *d_in, const hgx_bgzf_block *blocks, size_t n_blocks, unsigned char *d_out, hipStream_t st, int *bad,
                         void *staging) {
    ARGCHK(bad && (n_blocks == 0 || (d_in && blocks && d_out)));
    *bad = 0;
    if (n_blocks == 0) return HGX_OK;
    if (n_blocks >= (1u << 30)) { hgx_set_error("too many BGZF blocks for one launch"); return HGX_EINVAL; }
    std::vector<BlockDesc> h_own(staging ? 0 : n_blocks);
    BlockDesc *h = staging ? (BlockDesc *)staging : h_own.data();
    for (size_t i = 0; i < n_blocks; ++i) {
        const hgx_bgzf_block &b = blocks[i];
        if (b.out_len > 65536 || b.in_off + b.in_len >= (1ull << 32) || b.out_off + b.out_len >= (1ull << 32)) {
            hgx_set_error("BGZF block %zu does not fit the device inflate (payload %zu bytes)", i, (size_t)b.out_len);
            return HGX_EINVAL;
        }
        h[i] = BlockDesc{(uint32_t)b.in_off, (uint32_t)b.in_len, (uint32_t)b.out_off, (uint32_t)b.out_len, b.crc};
    }
    DevBuf b_desc, b_verdict;
    ALLOC(b_desc, n_blocks * sizeof(BlockDesc));
    ALLOC(b_verdict, n_blocks * 4);
    struct Drain { hipStream_t s; ~Drain() { (void)hipStreamSynchronize(s); } } drain{st};
    HIPCHK(hipMemcpyAsync(b_desc.p, h, n_blocks * sizeof(BlockDesc), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(b_verdict.p, 0xFF, n_blocks * 4, st));
    static const CrcOp op = make_crc_op();
    HGX_ONCE_PER_DEVICE({ const int rc_ = inflate_w_setup(); if (rc_) return rc_; });
    if (hgx_switch_has("front", "inflate_v1")) {                    // round 4's form: one symbol per trip (kept as the comparison form)
        HGX_ONCE_PER_DEVICE(HIPCHK(hipFuncSetAttribute((const void *)k_bgzf_inflate, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(InfLds))));
        k_bgzf_inflate<<<(unsigned)n_blocks, 64, sizeof(InfLds), st>>>(d_in, b_desc.as<BlockDesc>(), (int)n_blocks, d_out, op, b_verdict.as<uint32_t>());
    } else if (hgx_switch_has("front", "inflate_prof")) {           // the lane-parallel form with clock64 laps per phase (a measuring aid)
        DevBuf b_prof;
        ALLOC(b_prof, n_blocks * 12 * 8);
        HIPCHK(hipMemsetAsync(b_prof.p, 0, n_blocks * 12 * 8, st));
        k_bgzf_inflate_w<true><<<(unsigned)n_blocks, 64, sizeof(InfLdsW), st>>>(d_in, b_desc.as<BlockDesc>(), (int)n_blocks, d_out, op, b_verdict.as<uint32_t>(),
                                                                                b_prof.as<unsigned long long>());
        std::vector<unsigned long long> hp(n_blocks * 12);
        HIPCHK(hipMemcpyAsync(hp.data(), b_prof.p, hp.size() * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        double tot[12] = {0};
        double n_slow = 0;
        for (size_t i = 0; i < n_blocks; ++i) {
            n_slow += (double)(hp[i * 12 + 11] >> 32);
            hp[i * 12 + 11] &= 0xFFFFFFFFull;
            for (int k = 0; k < 12; ++k) tot[k] += (double)hp[i * 12 + k];
        }
        static const char *const nm[8] = {"header", "decode", "walk", "literals", "near matches", "far matches", "flush", "slow"};
        fprintf(stderr, "[k_bgzf_inflate_w] %zu blocks, per block: %.0f bytes, %.0f windows, %.0f near + %.0f far matches, %.1f slow symbols; clock64 ticks per block:", n_blocks,
                tot[11] / n_blocks, tot[8] / n_blocks, tot[9] / n_blocks, tot[10] / n_blocks, n_slow / n_blocks);
        for (int k = 0; k < 8; ++k) fprintf(stderr, " %s %.0f", nm[k], tot[k] / n_blocks);
        fprintf(stderr, "\n");
    } else if (hgx_switch_has("front", "inflate_small_pool")) {
        // (test switch: 32 entries of the second-level pool instead of all -- most long codes then overflow into the wave-uniform path,
        // which the tests want to see taken)
        k_bgzf_inflate_w<false, true><<<(unsigned)n_blocks, 64, sizeof(InfLdsW), st>>>(d_in, b_desc.as<BlockDesc>(), (int)n_blocks, d_out, op, b_verdict.as<uint32_t>(), nullptr);
    } else
        k_bgzf_inflate_w<false><<<(unsigned)n_blocks, 64, sizeof(InfLdsW), st>>>(d_in, b_desc.as<BlockDesc>(), (int)n_blocks, d_out, op, b_verdict.as<uint32_t>(), nullptr);
    HIPCHK(hipGetLastError());
    std::vector<uint32_t> v_own(staging ? 0 : n_blocks);
    uint32_t *v = staging ? (uint32_t *)((char *)staging + ((n_blocks * sizeof(BlockDesc) + 63) & ~(size_t)63)) : v_own.data();
    HIPCHK(hipMemcpyAsync(v, b_verdict.p, n_blocks * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    int n_bad = 0;
    for (size_t i = 0; i < n_blocks; ++i) n_bad += v[i] != 0 ? 1 : 0;
    *bad = n_bad;
    return HGX_OK;
}

// test / tool entry (hgx.h): a whole BGZF file in host memory -> its payload in host memory, inflated on the device
extern "C" int hgx_bgzf_inflate(const void *bgzf, size_t n_bytes, void *out, size_t out_cap, size_t *n_out, int32_t *bad_blocks, void *stream) {
    ARGCHK(bgzf && n_out && bad_blocks && (out || out_cap == 0));
    hipStream_t st = (hipStream_t)stream;
    std::vector<hgx_bgzf_block> blocks;
    size_t total = 0;
    int rc = hgx_bgzf_scan((const unsigned char *)bgzf, n_bytes, blocks, &total);
    if (rc) return rc;
    *n_out = total;
    *bad_blocks = 0;
    if (total > out_cap) { hgx_set_error("output buffer too small (%zu < %zu)", out_cap, total); return HGX_EINVAL; }
    DevBuf b_in, b_out;
    ALLOC(b_in, n_bytes + 2048);
    ALLOC(b_out, std::max<size_t>(total, 16));
    struct Drain { hipStream_t s; ~Drain() { (void)hipStreamSynchronize(s); } } drain{st};
    HIPCHK(hipMemcpyAsync(b_in.p, bgzf, n_bytes, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync((char *)b_in.p + n_bytes, 0, 2048, st));
    int bad = 0;
    rc = hgx_bgzf_inflate_dev(b_in.as<unsigned char>(), blocks.data(), blocks.size(), b_out.as<unsigned char>(), st, &bad, nullptr);
    if (rc) return rc;
    *bad_blocks = bad;
    if (total) HIPCHK(hipMemcpyAsync(out, b_out.p, total, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return HGX_OK;
}
