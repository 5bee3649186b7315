// hgx_inflate.hip -- BGZF inflate on the device (row 8f-3, round 4): ONE WAVEFRONT PER BGZF BLOCK.
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

#include <cstdint>
#include <cstring>
#include <vector>

#include "hgx_common.hpp"
#include "hgx_internal.hpp"

namespace {

#ifndef HGX_INF_RING
#define HGX_INF_RING 8192
#define HGX_INF_FLUSH 2048
#endif
constexpr int RING = HGX_INF_RING, RMASK = RING - 1, FLUSH = HGX_INF_FLUSH, PIECE = FLUSH / 64;      // (see the header: the ring holds the near window only)
constexpr int LIT_P = 10, DIST_P = 8;

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
template <class T, int P>
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
                const uint16_t e = (uint16_t)((s << 4) | L | ((mark_literals && s < 256) ? 0x8000 : 0));      // (literal/length table: bit 15 = a literal)
                for (uint32_t k = rev; k < (1u << P); k += 1u << L) H.pt[k] = e;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    return true;
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
                    // (an incomplete literal/length code is legal only when it has a single code: zlib accepts that; so do we)
                    if (!huff_build<HuffLds, LIT_P>(S.lit, S.lens + 32, n_lit, lane, true, true)) { err = INF_BAD_LENGTHS; break; }
                }
                if (!huff_build<DistLds, DIST_P>(S.dist, S.lens + 32 + n_lit, n_dist, lane, true)) { err = INF_BAD_LENGTHS; break; }
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
            if (data[p] == 66 && data[p + 1] == 67 && slen == 2) bsize = (long)rd16(p + 4);
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

// inflate `blocks` of the BGZF bytes at d_in (device; padded by >= 1 KB) into d_out (device); *bad = blocks whose verdict is not 0
// (after a stream synchronisation).  Blocks of more than 64 KB of payload, or a stream beyond 32-bit offsets: HGX_EINVAL.
int hgx_bgzf_inflate_dev(const unsigned char *d_in, const hgx_bgzf_block *blocks, size_t n_blocks, unsigned char *d_out, hipStream_t st, int *bad) {
    ARGCHK(bad && (n_blocks == 0 || (d_in && blocks && d_out)));
    *bad = 0;
    if (n_blocks == 0) return HGX_OK;
    if (n_blocks >= (1u << 30)) { hgx_set_error("too many BGZF blocks for one launch"); return HGX_EINVAL; }
    std::vector<BlockDesc> h(n_blocks);
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
    HIPCHK(hipMemcpyAsync(b_desc.p, h.data(), n_blocks * sizeof(BlockDesc), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(b_verdict.p, 0xFF, n_blocks * 4, st));
    static const CrcOp op = make_crc_op();
    HGX_ONCE_PER_DEVICE(HIPCHK(hipFuncSetAttribute((const void *)k_bgzf_inflate, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(InfLds))));
    k_bgzf_inflate<<<(unsigned)n_blocks, 64, sizeof(InfLds), st>>>(d_in, b_desc.as<BlockDesc>(), (int)n_blocks, d_out, op, b_verdict.as<uint32_t>());
    HIPCHK(hipGetLastError());
    std::vector<uint32_t> v(n_blocks);
    HIPCHK(hipMemcpyAsync(v.data(), b_verdict.p, n_blocks * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    int n_bad = 0;
    for (uint32_t x : v) n_bad += x != 0 ? 1 : 0;
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
    rc = hgx_bgzf_inflate_dev(b_in.as<unsigned char>(), blocks.data(), blocks.size(), b_out.as<unsigned char>(), st, &bad);
    if (rc) return rc;
    *bad_blocks = bad;
    if (total) HIPCHK(hipMemcpyAsync(out, b_out.p, total, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return HGX_OK;
}
