// hgx_front_core.hpp -- the per-key and per-pair logic of the DEVICE front end (rows 8a-2 .. 8a-5 on the GPU), written once as
// plain functions over POD tables so that the same source is compiled
//   * by hipcc into the kernels of hgx_front.hip (one lane per distinct decode key / per read pair), and
//   * by g++ into the lab library's emulation (csrc/lab: the kernels run as loops on the CPU), which the CPU test-suite
//     compares bit for bit with the pinned host front end (hgx_sam.cpp) -- the GPU tests then compare device vs host.
// It restates this repository's own host front end (hgx_sam.cpp Parser::decode / error_correct / ambiguous / exon_pieces,
// hgx_host.cpp hgx_intern_piece), which in turn follows the reference:
//   CIGAR x MD x Zs walk        hisatgenotype_typing_core.py:899-1124
//   error_correct               hisatgenotype_typing_core.py:119-243
//   novel variants, cmp_list2   hisatgenotype_typing_core.py:404-431, 1126-1164, 1351-1368
//   identify_ambigious_diffs    hisatgenotype_typing_common.py:1663-1955
//   haplotype assembly          hisatgenotype_typing_core.py:1386-1406
//   get_exon_haplotypes         hisatgenotype_typing_core.py:718-792
//   add_count's span scan       hisatgenotype_typing_core.py:641-670   (as piece masks)
//   pair protocol               hisatgenotype_typing_core.py:1238-1347, 1545-1587
// No allocation, no exceptions, no recursion, fixed-size scratch: whatever does not fit, and every input on which the
// reference would raise, makes the function return a negative FE_E* code -- the caller then DECLINES the whole call and the
// host front end (which reproduces the reference's failure, message and all) takes it.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define FE_HD __host__ __device__
#else
#define FE_HD
#endif

// ---- limits of the fixed-size scratch ------------------------------------------------------------------------------------
#define FE_MAX_CMP 64        // cmp_list entries of one record
#define FE_MAX_ZS 48         // Zs items
#define FE_MAX_OPS 24        // CIGAR ops
#define FE_MAX_SIDE 32       // alternatives of one side (left_alt_set / right_alt_set)
#define FE_MAX_SIDE_IDS 256  // ... and their ids, pooled
#define FE_MAX_MID 48        // ids between cmp_left and cmp_right
#define FE_MAX_IDS 64       // ids of one haplotype
#define FE_MAX_NW 16         // 32-variant words one piece may span on the device
#define FE_MAX_EXTRA 64      // as hgx_intern_piece
#define FE_MAX_EXON_PIECES 12
#define FE_MAX_JOIN 192      // characters of a joined id list

// decline codes (diagnostics only: any of them sends the call to the host front end)
#define FE_E_ZS -101         // malformed / unresolvable Zs
#define FE_E_MD -102         // MD missing / malformed / exhausted
#define FE_E_CIGAR -103      // malformed CIGAR, unsupported op, misplaced soft clip
#define FE_E_SHORT -104      // read shorter than its CIGAR
#define FE_E_ASSERT -105     // an assertion of the reference would fire
#define FE_E_CAP -106        // a scratch limit of this file
#define FE_E_NOVEL -107      // a novel indel outside the id encoding's range
#define FE_E_AMB -108        // KeyError / IndexError / assert / check_amb_uniqueness inside identify_ambigious_diffs
#define FE_E_PIECE -109      // piece with left > right, too many spanning variants, too wide for the device path
#define FE_E_PAIR -110       // more than 65535 refs for one pair and level
#define FE_E_POOL -111       // an output pool of the launch is full

#if defined(FE_DEBUG) && !defined(__HIP_DEVICE_COMPILE__)
#include <stdio.h>
#define FE_FAIL(code) (fprintf(stderr, "[hgx_front_core] decline %d at line %d\n", (code), __LINE__), (code))
#else
#define FE_FAIL(code) (code)
#endif

#define FE_T_MATCH 0
#define FE_T_MISMATCH 1
#define FE_T_INSERTION 2
#define FE_T_DELETION 3
#define FE_VAR_INSERTION 0
#define FE_VAR_SINGLE 1
#define FE_VAR_DELETION 2
#define FE_BASE_HLA 0

// Variant ids: [0, V) known, -1 "unknown", -2 n/a (match).  Novel variants get no table on the device: an id only has to be
// equal for equal variants, larger than V, and -- for the indels, which stay in haplotypes -- give back type, position and
// length (get_exon_haplotypes reads them): bit 30 | deletion << 29 | position << 12 | length.  Novel singles never leave
// cmp_list (cmp_list2 folds them into matches, typing_core.py:1351-1368): one marker value.
#define FE_NOVEL_BIT 0x40000000
#define FE_NOVEL_SINGLE 0x7fffffff
#define FE_NO_SLOT 0xFFFFFFFFu

struct FeLocus {
    int32_t V, n_ref, base_kind, n_exons, n_hv;
    const int32_t *pos, *right, *len, *maxright;
    const uint8_t *type, *linked;
    const char *base;
    const uint32_t *linked_bits;     // [n_words]
    const char *backbone;
    const int32_t *exons;            // [n_exons][2]
    const int32_t *hv_index;         // "hv<n>" -> variant or -1
    const int32_t *name_off;         // [V + 1] into name_pool
    const char *name_pool;
    // alternatives (typing_common.py:1424-1657), dir 0 = Alts_left_list (anchor = right end), 1 = Alts_right_list (left end)
    int32_t n_alt[2];
    const int32_t *alt_anchor[2];    // [n] sorted
    const int32_t *alt_key_off[2];   // [n + 1] into alt_ints: the key as left, vars..., right
    const int32_t *alt_str_off[2];   // [n + 1] into alt_chars: the key as the reference spells it (host only: the lab emulation's cross-check)
    const int32_t *alt_list_off[2];  // [n + 1] into alt_ht_off: the record's alternatives
    const int32_t *alt_ht_off;       // [n_hts + 1] into alt_ints
    const int32_t *alt_ints;
    const char *alt_chars;
};

struct FeParse {
    int32_t num_editdist, error_correction;
};

struct FePile {
    const uint8_t *nt_set;           // [n_ref] 4-bit masks A=1 C=2 G=4 T=8
    const uint32_t *counts;          // [n_ref][6] A C G T N D
};

// one distinct decode key: where its fields lie in `text` (the gathered key text of the key route, the SAM text or the inflated
// BAM stream of the record route)
#define FE_K_HAS_ZS 1
#define FE_K_HAS_MD 2
#define FE_K_BIN_CIGAR 4             // cigar_off -> n ops of uint32 little endian (len << 4 | op), cigar_len = n
#define FE_K_PACKED_SEQ 8            // seq_off -> 4-bit bases, two per byte ("=ACMGRSVTWYHKDBN")
struct FeKey {
    int32_t pos;                     // POS - (base_locus + 1)
    uint32_t n_pile;                 // records of the key that count into the pileup
    uint32_t slot;                   // decode slot or FE_NO_SLOT (pileup only)
    uint32_t cigar_off, seq_off, zs_off, md_off;
    uint32_t seq_len;
    uint16_t cigar_len, zs_len, md_len;
    uint16_t flags;
    uint32_t task;                   // many-task batches (the samples of one locus in one pass): whose pileup corrects this key
};

// read bases of a record, text or BAM-packed
struct FeSeq {
    const unsigned char *p;
    int len;
    bool packed;
    FE_HD char at(int i) const {
        if (!packed) return (char)p[i];
        const int nib = (i & 1) ? (p[i >> 1] & 15) : (p[i >> 1] >> 4);
        // "=ACMGRSVTWYHKDBN"[nib] out of two registers (the string literal was a memory load per base)
        const uint64_t tab = nib < 8 ? 0x565352474D43413DULL : 0x4E42444B48595754ULL;
        return (char)((tab >> (8 * (nib & 7))) & 0xff);
    }
};
FE_HD inline FeSeq fe_seq_of(const FeKey &K, const char *text) {
    FeSeq q;
    q.p = (const unsigned char *)text + K.seq_off;
    q.len = (int)K.seq_len;
    q.packed = (K.flags & FE_K_PACKED_SEQ) != 0;
    return q;
}
// CIGAR ops one by one, from text ("76M2D74M") or BAM words.  next(): 1 = an op, 0 = end, -1 = what strtol would not read as
// plain digits followed by an op character
struct FeCigar {
    const unsigned char *p;
    int n, at;
    bool bin;
    FE_HD int next(char &op, int &len) {
        if (bin) {
            if (at >= n) return 0;
            const unsigned char *w = p + 4 * at;
            const uint32_t v = (uint32_t)w[0] | ((uint32_t)w[1] << 8) | ((uint32_t)w[2] << 16) | ((uint32_t)w[3] << 24);
            ++at;
            op = (v & 15) < 9 ? "MIDNSHP=X"[v & 15] : '?';
            len = (int)(v >> 4);
            return 1;
        }
        if (at >= n) return 0;
        long v = 0;
        int nd = 0;
        while (at < n && p[at] >= '0' && p[at] <= '9') { if (v > 100000000) return -1; v = v * 10 + (p[at++] - '0'); nd++; }
        if (nd == 0 || at >= n) return -1;
        op = (char)p[at++];
        len = (int)v;
        return 1;
    }
};
FE_HD inline FeCigar fe_cigar_of(const FeKey &K, const char *text) {
    FeCigar c;
    c.p = (const unsigned char *)text + K.cigar_off;
    c.n = K.cigar_len;
    c.at = 0;
    c.bin = (K.flags & FE_K_BIN_CIGAR) != 0;
    return c;
}

struct FeCmp {
    int32_t type;                    // FE_T_* | read base << 8 (mismatch entries: the base the read shows AFTER error correction)
    int32_t pos, len, id;
};
FE_HD inline int fe_type(const FeCmp &c) { return c.type & 3; }
FE_HD inline char fe_base(const FeCmp &c) { return (char)((c.type >> 8) & 0xff); }
FE_HD inline int32_t fe_mk(int type, char base) { return type | ((int32_t)(unsigned char)base << 8); }

FE_HD inline int fe_lower_bound(const int32_t *a, int n, int key) {     // typing_common.py:406-422
    int lo = 0, hi = n;
    while (lo < hi) {
        const int m = (lo + hi) / 2;
        if (a[m] < key) lo = m + 1;
        else hi = m;
    }
    return lo;
}
FE_HD inline int fe_nt_bit(char c) { return c == 'A' ? 1 : c == 'C' ? 2 : c == 'G' ? 4 : c == 'T' ? 8 : 0; }
FE_HD inline char fe_single_nt(int m) { return m == 1 ? 'A' : m == 2 ? 'C' : m == 4 ? 'G' : 'T'; }
FE_HD inline bool fe_is_acgt(char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }

// accessors over known + novel ids (novel: indels only)
FE_HD inline bool fe_is_hv(const FeLocus &L, int id) { return id >= 0 && id < L.V; }
FE_HD inline int fe_vtype(const FeLocus &L, int id) { return id < L.V ? (int)L.type[id] : ((id >> 29) & 1 ? FE_VAR_DELETION : FE_VAR_INSERTION); }
FE_HD inline int fe_vpos(const FeLocus &L, int id) { return id < L.V ? L.pos[id] : (id >> 12) & 0x1ffff; }
FE_HD inline int fe_vlen(const FeLocus &L, int id) { return id < L.V ? L.len[id] : id & 0xfff; }

// first KNOWN variant at `pos` of the wanted type and size / base (typing_core.py:949-961, 1005-1017, 1045-1057), or -1
FE_HD inline int fe_lookup(const FeLocus &L, int pos, int type, int key) {
    for (int j = fe_lower_bound(L.pos, L.V, pos); j < L.V && L.pos[j] == pos; ++j) {
        if (L.type[j] != type) continue;
        if (type == FE_VAR_SINGLE ? L.base[j] == (char)key : L.len[j] == key) return j;
    }
    return -1;
}

// ---- error_correct (typing_core.py:119-243) over cl[start, n): returns the number of corrections, < 0 = decline ------------
FE_HD inline int fe_error_correct(const FeLocus &L, const FePile &P, const FeSeq &seq, int read_pos, FeCmp *cl, int start,
                                  int &n_cl, FeCmp *out) {
    const int n_ref = L.n_ref;
    const int seq_len = seq.len;
    int ncorr = 0, n_out = 0;
    bool stopped = false;
    for (int i = start; i < n_cl; ++i) {
        FeCmp c = cl[i];
        const int c_len = c.len;
        if (stopped || c.pos >= n_ref) {
            stopped = true;
            if (n_out >= FE_MAX_CMP) return FE_FAIL(FE_E_CAP);
            out[n_out++] = c;
            continue;
        }
        if (fe_type(c) == FE_T_MATCH) {
            int last = 0;
            for (int j = 0; j < c.len; ++j) {
                // eight bases at a time while none of them is a candidate (a pileup set that is not empty and does not hold the read's
                // base): one 8-byte look at the sets, eight independent looks at the read -- the common case of a read without errors
                // costs a round trip per eight bases instead of two per base
                while (j + 8 <= c.len && read_pos + j + 8 <= seq_len && c.pos + j + 8 <= n_ref) {
                    uint64_t s8, b8 = 0;
                    __builtin_memcpy(&s8, P.nt_set + c.pos + j, 8);
                    for (int k = 0; k < 8; ++k) b8 |= (uint64_t)fe_nt_bit(seq.at(read_pos + j + k)) << (8 * k);
                    const uint64_t t = s8 & b8;
                    const uint64_t t_zero = ~(((t & 0x7F7F7F7F7F7F7F7FULL) + 0x7F7F7F7F7F7F7F7FULL) | t) & 0x8080808080808080ULL;
                    const uint64_t s_zero = ~(((s8 & 0x7F7F7F7F7F7F7F7FULL) + 0x7F7F7F7F7F7F7F7FULL) | s8) & 0x8080808080808080ULL;
                    if (t_zero & ~s_zero) break;                   // a candidate among the eight: one by one from here
                    j += 8;
                }
                if (j >= c.len) break;
                if (read_pos + j >= seq_len || c.pos + j >= n_ref) continue;
                char b = seq.at(read_pos + j);
                const int s = P.nt_set[c.pos + j];
                if (s != 0 && !(s & fe_nt_bit(b))) {
                    b = (s & (s - 1)) ? 'N' : fe_single_nt(s);
                    if (b == L.backbone[c.pos + j]) return FE_FAIL(FE_E_ASSERT);
                    FeCmp m;
                    m.type = fe_mk(FE_T_MISMATCH, b);
                    m.pos = c.pos + j;
                    m.len = 1;
                    m.id = -1;
                    ncorr++;
                    if (b != 'N') m.id = fe_lookup(L, c.pos + j, FE_VAR_SINGLE, b);
                    if (n_out + 2 > FE_MAX_CMP) return FE_FAIL(FE_E_CAP);
                    if (j > last) { FeCmp t; t.type = FE_T_MATCH; t.pos = c.pos + last; t.len = j - last; t.id = -2; out[n_out++] = t; }
                    out[n_out++] = m;
                    last = j + 1;
                }
            }
            if (last < c.len) {
                if (n_out >= FE_MAX_CMP) return FE_FAIL(FE_E_CAP);
                FeCmp t; t.type = FE_T_MATCH; t.pos = c.pos + last; t.len = c.len - last; t.id = -2;
                out[n_out++] = t;
            }
        } else {
            if (read_pos >= seq_len) return FE_FAIL(FE_E_SHORT);
            char b = seq.at(read_pos);
            const int s = P.nt_set[c.pos];
            if (s != 0 && !(s & fe_nt_bit(b))) {
                b = (s & (s - 1)) ? 'N' : fe_single_nt(s);
                if (b == 'N') { c.id = -1; c.type = fe_mk(fe_type(c), b); }
                else if (b == L.backbone[c.pos]) { c.type = FE_T_MATCH; c.len = 1; c.id = -2; ncorr++; }
                else { c.id = fe_lookup(L, c.pos, FE_VAR_SINGLE, b); c.type = fe_mk(fe_type(c), b); }
            }
            if (n_out >= FE_MAX_CMP) return FE_FAIL(FE_E_CAP);
            out[n_out++] = c;
        }
        read_pos += c_len;
    }
    n_cl = start;
    for (int k = 0; k < n_out; ++k) {                                  // merge adjacent matches (typing_core.py:225-235)
        if (fe_type(out[k]) == FE_T_MATCH && n_cl > start && fe_type(cl[n_cl - 1]) == FE_T_MATCH) cl[n_cl - 1].len += out[k].len;
        else {
            if (n_cl >= FE_MAX_CMP) return FE_FAIL(FE_E_CAP);
            cl[n_cl++] = out[k];
        }
    }
    return ncorr;
}

// ---- one record -> cmp_list with novel ids (typing_core.py:876-1164).  1 = kept, 0 = dropped, < 0 = decline --------------
FE_HD inline int fe_decode(const FeLocus &L, const FeParse &o, const FePile &P, int pos, FeCigar cigar, const FeSeq &seq,
                           const char *zs, int zs_len, const char *md, int md_n, FeCmp *cl, int &n_cl, FeCmp *tmp) {
    const int seq_len = seq.len;
    int zs_gap[FE_MAX_ZS], zs_id[FE_MAX_ZS];
    char zs_type[FE_MAX_ZS];
    int n_zs = 0;
    {
        int p = 0;
        while (p < zs_len) {
            if (n_zs >= FE_MAX_ZS) return FE_FAIL(FE_E_CAP);
            int gap = 0, nd = 0;
            while (p < zs_len && zs[p] >= '0' && zs[p] <= '9') { if (gap > 100000000) return FE_FAIL(FE_E_ZS); gap = gap * 10 + (zs[p++] - '0'); nd++; }
            if (nd == 0) return FE_FAIL(FE_E_ZS);                 // (sign / blank forms of strtol: left to the host)
            if (p + 2 >= zs_len || zs[p] != '|' || zs[p + 2] != '|') return FE_FAIL(FE_E_ZS);
            const char t = zs[p + 1];
            p += 3;
            int q = p;
            while (q < zs_len && zs[q] != ',') ++q;
            int id = -1;
            if (q - p > 2 && zs[p] == 'h' && zs[p + 1] == 'v' && !(q - p > 3 && zs[p + 2] == '0')) {
                long num = 0;
                bool digits = true;
                for (int c = p + 2; c < q; ++c) {
                    if (zs[c] < '0' || zs[c] > '9') { digits = false; break; }
                    num = num * 10 + (zs[c] - '0');
                    if (num > 100000000) { digits = false; break; }
                }
                if (digits && num < (long)L.n_hv) id = L.hv_index[num];
            }
            if (id < 0) return FE_FAIL(FE_E_ZS);                  // another spelling of an id, or an id the locus does not have
            zs_gap[n_zs] = gap; zs_type[n_zs] = t; zs_id[n_zs] = id;
            n_zs++;
            p = q < zs_len ? q + 1 : q;
        }
    }
    if (md_n <= 0) return FE_FAIL(FE_E_MD);
    char op_c[FE_MAX_OPS];
    int op_n[FE_MAX_OPS];
    int n_ops = 0;
    for (;;) {
        char op;
        int n;
        const int r = cigar.next(op, n);
        if (r == 0) break;
        if (r < 0) return FE_FAIL(FE_E_CIGAR);
        if (n_ops >= FE_MAX_OPS) return FE_FAIL(FE_E_CAP);
        op_c[n_ops] = op;
        op_n[n_ops++] = n;
    }
    int md_i = 0, md_len = 0;
    int zs_i = 0;
    int zs_pos = n_zs ? zs_gap[0] : 0;
    int rp = 0, gp = pos;
    int n_ec = 0;
    bool bad = false;
    n_cl = 0;
#define FE_PUSH(T, P_, L_, I_)                                   \
    do {                                                         \
        if (n_cl >= FE_MAX_CMP) return FE_FAIL(FE_E_CAP);                 \
        cl[n_cl].type = (T); cl[n_cl].pos = (P_); cl[n_cl].len = (L_); cl[n_cl].id = (I_); \
        n_cl++;                                                  \
    } while (0)
#define FE_ZS_ADVANCE(consume)                                   \
    do {                                                         \
        zs_i++;                                                  \
        if (consume) zs_pos += 1;                                \
        if (zs_i < n_zs) zs_pos += zs_gap[zs_i];                 \
    } while (0)
    for (int ci = 0; ci < n_ops; ++ci) {
        const char op = op_c[ci];
        const int n = op_n[ci];
        if (op == 'M') {
            bool first = true;
            int used = 0;
            const int start = n_cl;
            for (;;) {
                if (!first || md_len == 0) {
                    if (md_i >= md_n) return FE_FAIL(FE_E_MD);
                    if (md[md_i] >= '0' && md[md_i] <= '9') {
                        long num = 0;
                        while (md_i < md_n && md[md_i] >= '0' && md[md_i] <= '9') { num = num * 10 + (md[md_i++] - '0'); if (num > 100000000) return FE_FAIL(FE_E_MD); }
                        md_len += (int)num;
                    }
                }
                if (md_len >= n) {
                    md_len -= n;
                    if (n > used) FE_PUSH(FE_T_MATCH, gp + used, n - used, -2);
                    break;
                }
                first = false;
                if (rp + md_len >= seq_len) return FE_FAIL(FE_E_SHORT);
                const char base = seq.at(rp + md_len);
                if (md_i >= md_n || !fe_is_acgt(md[md_i])) return FE_FAIL(FE_E_MD);
                md_i++;
                if (md_len > used) FE_PUSH(FE_T_MATCH, gp + used, md_len - used, -2);
                int id;
                if (rp + md_len == zs_pos && zs_i < n_zs) {
                    if (zs_type[zs_i] != 'S') return FE_FAIL(FE_E_ASSERT);
                    id = zs_id[zs_i];
                    FE_ZS_ADVANCE(true);
                } else id = fe_lookup(L, gp + md_len, FE_VAR_SINGLE, base);
                FE_PUSH(fe_mk(FE_T_MISMATCH, base), gp + md_len, 1, id);
                used = md_len + 1;
                md_len += 1;
                if (md_len == n) { md_len = 0; break; }
            }
            if (o.error_correction) {
                const int r = fe_error_correct(L, P, seq, rp, cl, start, n_cl, tmp);
                if (r < 0) return r;
                n_ec += r;
            }
        } else if (op == 'I') {
            int id;
            if (rp == zs_pos && zs_i < n_zs) {
                if (zs_type[zs_i] != 'I') return FE_FAIL(FE_E_ASSERT);
                id = zs_id[zs_i];
                FE_ZS_ADVANCE(false);
            } else id = fe_lookup(L, gp, FE_VAR_INSERTION, n);
            FE_PUSH(FE_T_INSERTION, gp, n, id);
            for (int k = rp; k < rp + n && k < seq_len; ++k)
                if (seq.at(k) == 'N') bad = true;
        } else if (op == 'D') {
            if (md_i < md_n && md[md_i] == '0') md_i++;
            if (md_i >= md_n || md[md_i] != '^') return FE_FAIL(FE_E_MD);
            md_i++;
            while (md_i < md_n && fe_is_acgt(md[md_i])) md_i++;
            int id;
            if (rp == zs_pos && zs_i < n_zs && zs_type[zs_i] == 'D') {
                id = zs_id[zs_i];
                FE_ZS_ADVANCE(false);
            } else id = fe_lookup(L, gp, FE_VAR_DELETION, n);
            FE_PUSH(FE_T_DELETION, gp, n, id);
            if (gp < L.n_ref) {                                    // artificial-deletion check (typing_core.py:1064-1077)
                const uint32_t *c = P.counts + (size_t)gp * 6;
                const uint64_t dc = c[5], nc = (uint64_t)c[0] + c[1] + c[2] + c[3] + c[4];
                if (L.base_kind == FE_BASE_HLA && dc * 6 < nc) bad = true;
            }
        } else if (op == 'S') {
            if (ci == 0) zs_pos += n;
            else if (ci + 1 != n_ops) return FE_FAIL(FE_E_CIGAR);
        } else return FE_FAIL(FE_E_CIGAR);
        if (op == 'M' || op == 'N' || op == 'D') gp += n;
        if (op == 'M' || op == 'I' || op == 'S') rp += n;
    }
#undef FE_PUSH
#undef FE_ZS_ADVANCE
    if (gp > L.n_ref) return 0;
    if (n_ec > (o.num_editdist > 1 ? o.num_editdist : 1)) return 0;
    if (bad) return 0;
    for (int k = 0; k < n_cl; ++k) {                                 // novel variants (typing_core.py:1126-1164)
        FeCmp &c = cl[k];
        const int t = fe_type(c);
        if (t == FE_T_MATCH || c.id != -1) continue;
        if (t == FE_T_MISMATCH) {
            if (fe_base(c) != 'N') c.id = FE_NOVEL_SINGLE;
        } else {
            if (c.pos < 0 || c.pos > 0x1ffff || c.len < 0 || c.len > 0xfff) return FE_FAIL(FE_E_NOVEL);
            c.id = FE_NOVEL_BIT | (t == FE_T_DELETION ? 1 << 29 : 0) | (c.pos << 12) | c.len;
        }
    }
    return 1;
}

// cmp_list2 (typing_core.py:1351-1368), in place
FE_HD inline void fe_cmp_list2(const FeLocus &L, FeCmp *cl, int &n) {
    int m = 0;
    for (int k = 0; k < n; ++k) {
        const FeCmp c = cl[k];
        const int t = fe_type(c);
        if (t == FE_T_MATCH) {
            if (m > 0 && fe_type(cl[m - 1]) == FE_T_MATCH) cl[m - 1].len += c.len;
            else cl[m++] = c;
        } else if (t == FE_T_MISMATCH && (c.id == -1 || c.id >= L.V)) {
            if (m > 0 && fe_type(cl[m - 1]) == FE_T_MATCH) cl[m - 1].len += 1;
            else { cl[m].type = FE_T_MATCH; cl[m].pos = c.pos; cl[m].len = 1; cl[m].id = -2; m++; }
        } else cl[m++] = c;
    }
    n = m;
}

// ---- identify_ambigious_diffs (typing_common.py:1663-1955) ------------------------------------------------------------------
struct FeSide {                      // one side's alternatives: coordinate + ids, ids pooled
    int n;
    int coord[FE_MAX_SIDE];
    short off[FE_MAX_SIDE], cnt[FE_MAX_SIDE];
    int n_ids;
    int ids[FE_MAX_SIDE_IDS];
};

// adds (coord, ids[0..n)) unless present; < 0 = decline
FE_HD inline int fe_side_add(FeSide &s, int coord, const int *ids, int n) {
    for (int k = 0; k < s.n; ++k) {
        if (s.coord[k] != coord || s.cnt[k] != n) continue;
        bool eq = true;
        for (int i = 0; i < n && eq; ++i) eq = s.ids[s.off[k] + i] == ids[i];
        if (eq) return 0;
    }
    if (s.n >= FE_MAX_SIDE || s.n_ids + n > FE_MAX_SIDE_IDS) return FE_FAIL(FE_E_CAP);
    s.coord[s.n] = coord;
    s.off[s.n] = (short)s.n_ids;
    s.cnt[s.n] = (short)n;
    for (int i = 0; i < n; ++i) s.ids[s.n_ids + i] = ids[i];
    s.n_ids += n;
    s.n++;
    return 0;
}

// key.find(cur_join) != -1 (typing_common.py:1744, 1868) on variant ids instead of text.  The reference joins the ids of the entries up
// to the current one with '-' ("hv12-hv40") and searches the alternative's key ("529-hv8-hv12-hv40-606") for that text.  Every id
// spells "hv<digits>" (a locus with other names is not taken: hgx_front_tables_build), the key's first and last fields are
// numbers, so a match can only START at the first letter of one of the key's ids and every id of the needle but the last is followed
// by '-': those must equal the key's ids; the needle's LAST id only has to be a decimal prefix of the key's ("hv4" is found in
// "...-hv40-..."): the reference's quirk, kept.  A novel id spells "nv<k>": no key contains an 'n'.
// kv[0..nk) = the key's variant ids, cur[0..n) = the needle's, n > 0.
FE_HD inline bool fe_name_is_prefix(const FeLocus &L, int a, int b) {              // name(a) is a prefix of name(b)
    const int a0 = L.name_off[a], al = L.name_off[a + 1] - a0, b0 = L.name_off[b], bl = L.name_off[b + 1] - b0;
    if (al > bl) return false;
    for (int k = 0; k < al; ++k) if (L.name_pool[a0 + k] != L.name_pool[b0 + k]) return false;
    return true;
}
FE_HD inline bool fe_key_contains(const FeLocus &L, const int *kv, int nk, const int *cur, int n) {
    for (int t = 0; t < n; ++t) if (cur[t] < 0 || cur[t] >= L.V) return false;
    const int last = cur[n - 1];
    for (int p = 0; p + n <= nk; ++p) {
        int t = 0;
        while (t < n - 1 && kv[p + t] == cur[t]) ++t;
        if (t == n - 1 && (kv[p + t] == last || fe_name_is_prefix(L, last, kv[p + t]))) return true;
    }
    return false;
}
#if defined(HGX_LAB) && !defined(__HIP_DEVICE_COMPILE__)
// The lab library's CPU emulation checks the predicate above against the reference's own form -- the text search -- on every
// candidate it meets (a difference declines the input with FE_E_ASSERT: tests/lab_front_cases.py expects none).
#define FE_CHECK_KEY_TEXT 1
FE_HD inline int fe_join(const FeLocus &L, const int *ids, int n, char *buf) {
    int w = 0;
    for (int i = 0; i < n; ++i) {
        if (i) { if (w >= FE_MAX_JOIN) return -1; buf[w++] = '-'; }
        const int id = ids[i];
        if (id >= 0 && id < L.V) {
            const int a = L.name_off[id], b = L.name_off[id + 1];
            if (w + (b - a) > FE_MAX_JOIN) return -1;
            for (int k = a; k < b; ++k) buf[w++] = L.name_pool[k];
        } else {
            if (w + 2 > FE_MAX_JOIN) return -1;
            buf[w++] = 'n'; buf[w++] = 'v';
        }
    }
    return w;
}
FE_HD inline bool fe_contains(const char *hay, int n_hay, const char *needle, int n_needle) {     // str.find(...) != -1
    if (n_needle == 0) return true;
    for (int s = 0; s + n_needle <= n_hay; ++s) {
        int k = 0;
        while (k < n_needle && hay[s + k] == needle[k]) ++k;
        if (k == n_needle) return true;
    }
    return false;
}
FE_HD inline bool fe_key_text_agrees(const FeLocus &L, int dir, int j, const int *cur, int n, bool got) {
    char join[FE_MAX_JOIN];
    const int nj = fe_join(L, cur, n, join);
    if (nj < 0) return true;                                        // (longer than the scratch: nothing to compare with)
    const int s0 = L.alt_str_off[dir][j], s1 = L.alt_str_off[dir][j + 1];
    return fe_contains(L.alt_chars + s0, s1 - s0, join, nj) == got;
}
#endif

FE_HD inline int fe_ambiguous(const FeLocus &L, const FeCmp *c2, int n, int &cmp_left, int &cmp_right, FeSide &lset, FeSide &rset) {
    const int n_ref = L.n_ref;
    cmp_left = 0;
    cmp_right = n - 1;
    const int left = c2[0].pos, right = c2[n - 1].pos + c2[n - 1].len - 1;
    lset.n = lset.n_ids = 0;
    rset.n = rset.n_ids = 0;
    int cur[FE_MAX_IDS], part[FE_MAX_IDS], sids[FE_MAX_IDS];
    int rc;
#define FE_SKIP(c) (fe_type(c) == FE_T_MATCH ? false : fe_type(c) == FE_T_INSERTION ? true : !fe_is_hv(L, (c).id))
    // ---- left direction ----
    bool found = false;
    if (L.n_alt[0] > 0)
    for (int i = n - 1; i >= 0; --i) {
        const FeCmp ci = c2[i];
        if (FE_SKIP(ci)) continue;
        const int ti = fe_type(ci);
        const int cur_left = ci.pos;
        const int cur_right = (ti == FE_T_MATCH || ti == FE_T_DELETION) ? ci.pos + ci.len - 1 : ci.pos;
        const int na = L.n_alt[0];
        const int *anchor = L.alt_anchor[0];
        const int hi = fe_lower_bound(anchor, na, cur_right + 1);
        int j = (hi + 1 < na ? hi + 1 : na) - 1;
        if (j < 0 || anchor[j] < cur_left) continue;                 // no table entry anchored inside this entry
        int n_cur = 0, seqlen = 0;
        for (int k = 0; k <= i; ++k) {
            const int tk = fe_type(c2[k]);
            if (tk != FE_T_MATCH && c2[k].id != -1) { if (n_cur >= FE_MAX_IDS) return FE_FAIL(FE_E_CAP); cur[n_cur++] = c2[k].id; }
            if (tk == FE_T_MATCH) { const int e = c2[k].pos + c2[k].len < n_ref ? c2[k].pos + c2[k].len : n_ref; seqlen += e - c2[k].pos > 0 ? e - c2[k].pos : 0; }
            else if (tk == FE_T_MISMATCH) seqlen += 1;
        }
        bool i_found = false;
        for (; j >= 0; --j) {
            const int r_anchor = anchor[j];
            if (r_anchor < cur_left) break;
            if (r_anchor > cur_right) continue;
            const int *key_ht = L.alt_ints + L.alt_key_off[0][j];
            if (n_cur > 0) {
                const bool has = fe_key_contains(L, key_ht + 1, L.alt_key_off[0][j + 1] - L.alt_key_off[0][j] - 2, cur, n_cur);
#ifdef FE_CHECK_KEY_TEXT
                if (!fe_key_text_agrees(L, 0, j, cur, n_cur, has)) return FE_FAIL(FE_E_ASSERT);
#endif
                if (!has) continue;
            }
            const int flen = (L.alt_key_off[0][j + 1] - L.alt_key_off[0][j]) - 1;          // fields of key.split('-')[:-1]
            if (n_cur + 1 == flen) {
                if (left < key_ht[0]) continue;
            } else {
                int k = flen - n_cur - 1;
                if (k < 0) k += flen;                                   // Python negative index
                if (k <= 0 || k >= flen) return FE_FAIL(FE_E_AMB);
                if (left <= L.right[key_ht[k]]) continue;
            }
            i_found = true;
            for (int a = L.alt_list_off[0][j]; a < L.alt_list_off[0][j + 1]; ++a) {
                const int *alt = L.alt_ints + L.alt_ht_off[a];
                const int alt_n = L.alt_ht_off[a + 1] - L.alt_ht_off[a];
                const int a_right = alt[alt_n - 1];
                if (a_right > cur_right) return FE_FAIL(FE_E_AMB);
                int seq_pos = cur_right - a_right, cur_pos = a_right;
                int n_part = 0;                                         // built back to front: part[FE_MAX_IDS - n_part ..]
                for (int k = alt_n - 2; k >= 1; --k) {
                    const int v = alt[k];
                    int vp = L.pos[v];
                    if (L.type[v] == FE_VAR_DELETION) vp = vp + L.len[v] - 1;
                    if (vp > cur_pos) return FE_FAIL(FE_E_AMB);
                    int nsp = seq_pos + (cur_pos - vp);
                    if (nsp >= seqlen) break;
                    int ncp;
                    if (L.type[v] == FE_VAR_SINGLE) { nsp += 1; ncp = vp - 1; }
                    else if (L.type[v] == FE_VAR_DELETION) ncp = vp - L.len[v];
                    else return FE_FAIL(FE_E_AMB);
                    if (n_part >= FE_MAX_IDS) return FE_FAIL(FE_E_CAP);
                    part[FE_MAX_IDS - 1 - n_part] = v;
                    n_part++;
                    if (nsp >= seqlen) break;
                    seq_pos = nsp;
                    cur_pos = ncp;
                }
                if (n_part > 0) {
                    const int seq_left = seqlen - seq_pos - 1;
                    int ns = 0;
                    for (int k = 0; k < n_part; ++k) sids[ns++] = part[FE_MAX_IDS - n_part + k];
                    if (found)
                        for (int jj = i + 1; jj < cmp_left; ++jj)
                            if (fe_type(c2[jj]) != FE_T_MATCH && fe_is_hv(L, c2[jj].id)) { if (ns >= FE_MAX_IDS) return FE_FAIL(FE_E_CAP); sids[ns++] = c2[jj].id; }
                    rc = fe_side_add(lset, cur_pos - seq_left, sids, ns);
                    if (rc) return rc;
                }
            }
        }
        if (i_found) {
            if (!found) {
                cmp_left = i + 1;
                rc = fe_side_add(lset, left, cur, n_cur);
                if (rc) return rc;
            }
            found = true;
        }
    }
    if (!found) { rc = fe_side_add(lset, left, cur, 0); if (rc) return rc; }
    // ---- right direction ----
    found = false;
    if (L.n_alt[1] > 0)
    for (int i = 0; i < n; ++i) {
        const FeCmp ci = c2[i];
        if (FE_SKIP(ci)) continue;
        const int ti = fe_type(ci);
        const int cur_left = ci.pos;
        const int cur_right = (ti == FE_T_MATCH || ti == FE_T_DELETION) ? ci.pos + ci.len - 1 : ci.pos;
        const int na = L.n_alt[1];
        const int *anchor = L.alt_anchor[1];
        int j = fe_lower_bound(anchor, na, cur_left);
        if (j >= na || anchor[j] > cur_right) continue;
        int n_cur = 0, seqlen = 0;
        for (int k = i; k < n; ++k) {
            const int tk = fe_type(c2[k]);
            if (tk != FE_T_MATCH && c2[k].id != -1) { if (n_cur >= FE_MAX_IDS) return FE_FAIL(FE_E_CAP); cur[n_cur++] = c2[k].id; }
            if (tk == FE_T_MATCH) { const int e = c2[k].pos + c2[k].len < n_ref ? c2[k].pos + c2[k].len : n_ref; seqlen += e - c2[k].pos > 0 ? e - c2[k].pos : 0; }
            else if (tk == FE_T_MISMATCH) seqlen += 1;
        }
        bool i_found = false;
        for (; j < na; ++j) {
            const int r_anchor = anchor[j];
            if (r_anchor > cur_right) break;
            if (r_anchor < cur_left) continue;
            const int *key_ht = L.alt_ints + L.alt_key_off[1][j];
            if (n_cur > 0) {
                const bool has = fe_key_contains(L, key_ht + 1, L.alt_key_off[1][j + 1] - L.alt_key_off[1][j] - 2, cur, n_cur);
#ifdef FE_CHECK_KEY_TEXT
                if (!fe_key_text_agrees(L, 1, j, cur, n_cur, has)) return FE_FAIL(FE_E_ASSERT);
#endif
                if (!has) continue;
            }
            const int flen = (L.alt_key_off[1][j + 1] - L.alt_key_off[1][j]) - 1;          // fields of key.split('-')[1:]
            const int *f = key_ht + 1;
            if (n_cur + 1 == flen) {
                if (right > f[flen - 1]) continue;
            } else {
                const int k = n_cur;
                if (k >= flen) return FE_FAIL(FE_E_AMB);
                if (k == flen - 1) return FE_FAIL(FE_E_AMB);
                if (right >= L.pos[f[k]]) continue;
            }
            i_found = true;
            for (int a = L.alt_list_off[1][j]; a < L.alt_list_off[1][j + 1]; ++a) {
                const int *alt = L.alt_ints + L.alt_ht_off[a];
                const int alt_n = L.alt_ht_off[a + 1] - L.alt_ht_off[a];
                const int a_left = alt[0];
                if (cur_left > a_left) return FE_FAIL(FE_E_AMB);
                int seq_pos = a_left - cur_left, cur_pos = a_left;
                int n_part = 0;
                for (int k = 1; k + 1 < alt_n; ++k) {
                    const int v = alt[k];
                    const int vp = L.pos[v];
                    if (vp < cur_pos) return FE_FAIL(FE_E_AMB);
                    int nsp = seq_pos + (vp - cur_pos);
                    if (nsp >= seqlen) break;
                    int ncp;
                    if (L.type[v] == FE_VAR_SINGLE) { nsp += 1; ncp = vp + 1; }
                    else if (L.type[v] == FE_VAR_DELETION) ncp = vp + L.len[v];
                    else return FE_FAIL(FE_E_AMB);
                    if (n_part >= FE_MAX_IDS) return FE_FAIL(FE_E_CAP);
                    part[n_part++] = v;
                    if (nsp >= seqlen) break;
                    seq_pos = nsp;
                    cur_pos = ncp;
                }
                if (n_part > 0) {
                    const int seq_left = seqlen - seq_pos - 1;
                    if (seq_left < 0) return FE_FAIL(FE_E_AMB);
                    int ns = 0;
                    if (found)
                        for (int jj = cmp_right + 1; jj < i; ++jj)
                            if (fe_type(c2[jj]) != FE_T_MATCH && fe_is_hv(L, c2[jj].id)) { if (ns >= FE_MAX_IDS) return FE_FAIL(FE_E_CAP); sids[ns++] = c2[jj].id; }
                    for (int k = 0; k < n_part; ++k) { if (ns >= FE_MAX_IDS) return FE_FAIL(FE_E_CAP); sids[ns++] = part[k]; }
                    rc = fe_side_add(rset, cur_pos + seq_left, sids, ns);
                    if (rc) return rc;
                }
            }
        }
        if (i_found) {
            if (!found) {
                cmp_right = i - 1;
                rc = fe_side_add(rset, right, cur, n_cur);
                if (rc) return rc;
            }
            found = true;
        }
    }
#undef FE_SKIP
    if (!found) { rc = fe_side_add(rset, right, cur, 0); if (rc) return rc; }
    if (cmp_right < cmp_left) {
        cmp_left = 0;
        lset.n = lset.n_ids = 0;
        rc = fe_side_add(lset, left, cur, 0);
        if (rc) return rc;
    }
    // check_amb_uniqueness (validation_check.py:313-341, always on: quirk Q1): non-empty id lists are unique over both sides
    for (int x = 0; x < lset.n + rset.n; ++x) {
        const FeSide &sx = x < lset.n ? lset : rset;
        const int kx = x < lset.n ? x : x - lset.n;
        if (sx.cnt[kx] == 0) continue;
        for (int y = 0; y < x; ++y) {
            const FeSide &sy = y < lset.n ? lset : rset;
            const int ky = y < lset.n ? y : y - lset.n;
            if (sy.cnt[ky] != sx.cnt[kx]) continue;
            bool eq = true;
            for (int t = 0; t < sx.cnt[kx] && eq; ++t) eq = sx.ids[sx.off[kx] + t] == sy.ids[sy.off[ky] + t];
            if (eq) return FE_FAIL(FE_E_AMB);
        }
    }
    return 0;
}

// ---- haplotypes of a key: left alternative x mid x right alternative (typing_core.py:1386-1406) ----------------------------
struct FeHtView {                    // haplotype (l, r) of a key without materialising it
    const FeSide *ls, *rs;
    const int *mid;
    int n_mid, l, r;
    FE_HD int left() const { return ls->coord[l]; }
    FE_HD int right() const { return rs->coord[r]; }
    FE_HD int n_ids() const { return ls->cnt[l] + n_mid + rs->cnt[r]; }
    FE_HD int id(int k) const {
        if (k < ls->cnt[l]) return ls->ids[ls->off[l] + k];
        k -= ls->cnt[l];
        if (k < n_mid) return mid[k];
        return rs->ids[rs->off[r] + (k - n_mid)];
    }
};
FE_HD inline bool fe_ht_equal(const FeHtView &a, const FeHtView &b) {
    if (a.left() != b.left() || a.right() != b.right() || a.n_ids() != b.n_ids()) return false;
    const int n = a.n_ids();
    for (int k = 0; k < n; ++k) if (a.id(k) != b.id(k)) return false;
    return true;
}

// ---- get_exon_haplotypes (typing_core.py:718-792): pieces as (left, right, [i0, i1) of the haplotype's ids) -----------------
struct FeExonPiece { int left, right, i0, i1; };
FE_HD inline int fe_exon_pieces(const FeLocus &L, int ht_left, int ht_right, const int *ids, int n_ids, FeExonPiece *out, int &n_out) {
    n_out = 0;
    for (int e = 0; e < L.n_exons; ++e) {
        const int el = L.exons[2 * e], er = L.exons[2 * e + 1];
        int hl = ht_left, hr = ht_right;
        if (el > hr || er < hl) continue;
        int i0 = 0, i1 = n_ids;
        if (hl < el) {
            bool done = false;
            for (int i = 0; i < n_ids; ++i) {
                const int t = fe_vtype(L, ids[i]), p = fe_vpos(L, ids[i]);
                if ((t != FE_VAR_DELETION && p >= el) || (t == FE_VAR_DELETION && p - 1 >= el)) { hl = el; i0 = i; done = true; break; }
                if (t == FE_VAR_DELETION) {
                    const int r = p + fe_vlen(L, ids[i]);
                    if (r >= el) { hl = r; i0 = i + 1; done = true; break; }
                }
            }
            if (!done) { hl = el; i0 = i1 = 0; }
        }
        if (hl < el) return FE_FAIL(FE_E_ASSERT);
        if (hr > er) {
            bool done = false;
            for (int i = i1 - 1; i >= i0; --i) {
                const int t = fe_vtype(L, ids[i]);
                int r = fe_vpos(L, ids[i]);
                if (t == FE_VAR_DELETION) r = r + fe_vlen(L, ids[i]) - 1;
                if ((t != FE_VAR_DELETION && r <= er) || (t == FE_VAR_DELETION && r + 1 <= er)) { hr = er; i1 = i + 1; done = true; break; }
                if (t == FE_VAR_DELETION) {
                    const int l = r - fe_vlen(L, ids[i]);
                    if (l <= er) { hr = l; i1 = i; done = true; break; }
                }
            }
            if (!done) { hr = er; i0 = i1 = 0; }
        }
        if (hl > hr) return FE_FAIL(FE_E_ASSERT);
        if (n_out >= FE_MAX_EXON_PIECES) return FE_FAIL(FE_E_CAP);
        out[n_out].left = hl; out[n_out].right = hr; out[n_out].i0 = i0; out[n_out].i1 = i1;
        n_out++;
    }
    return 0;
}

// ---- piece "left-ids-right" -> (lo_word, n_words, masks[2 * n_words]) as hgx_intern_piece (hgx_host.cpp; typing_core.py:641-670) --
FE_HD inline uint64_t fe_piece_hash(uint16_t lo, uint16_t nw, const uint32_t *m) {       // PieceTable::hash
    uint64_t h = 1469598103934665603ull ^ lo ^ ((uint64_t)nw << 16);
    for (int i = 0; i < 2 * (int)nw; ++i) { h ^= m[i]; h *= 1099511628211ull; h ^= h >> 29; }
    return h;
}
// PieceTable::hash orders the piece table but is weak as an identity (lo_word is XOR-ed into the seed and the first mask word into
// that: (lo 8, mask 8) and (lo 0, mask 0) collide).  The device de-duplicates candidates on this one (splitmix64 steps over
// (lo, nw) and the (MP, P) word pairs); equal keys are still compared word for word.
FE_HD inline uint64_t fe_mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}
FE_HD inline uint64_t fe_piece_key(uint16_t lo, uint16_t nw, const uint32_t *m) {
    uint64_t h = fe_mix64(0x243F6A8885A308D3ull ^ lo ^ ((uint64_t)nw << 16));
    for (int i = 0; i < (int)nw; ++i) h = fe_mix64(h + 0x9e3779b97f4a7c15ull + (((uint64_t)m[2 * i + 1] << 32) | m[2 * i]));
    return h;
}
FE_HD inline int fe_piece_masks(const FeLocus &L, int left, int right, const int *ids, int n_ids, int &lo_w, int &nw, uint32_t *buf) {
    if (left > right) return FE_FAIL(FE_E_PIECE);
    const int V = L.V;
    const int i0 = fe_lower_bound(L.pos, V, left), i1 = fe_lower_bound(L.pos, V, right + 1);
    int lo_v = 0x7fffffff, hi_v = -1;
    if (i1 > i0) { lo_v = i0; hi_v = i1 - 1; }
    int extra[FE_MAX_EXTRA], n_extra = 0;
    for (int j = i0 - 1; j >= 0 && L.maxright[j] >= left; --j) {
        if (L.linked[j] && L.right[j] >= left && L.right[j] <= right) {
            if (n_extra >= FE_MAX_EXTRA) return FE_FAIL(FE_E_PIECE);
            extra[n_extra++] = j;
            if (j < lo_v) lo_v = j;
            if (j > hi_v) hi_v = j;
        }
    }
    for (int i = 0; i < n_ids; ++i) {
        const int v = ids[i];
        if (v >= 0 && v < V && L.linked[v]) { if (v < lo_v) lo_v = v; if (v > hi_v) hi_v = v; }
    }
    lo_w = 0;
    int hi_w = 0;
    if (hi_v >= 0) { lo_w = lo_v >> 5; hi_w = hi_v >> 5; }
    nw = hi_w - lo_w + 1;
    if (nw > FE_MAX_NW) return FE_FAIL(FE_E_PIECE);
    for (int k = 0; k < 2 * nw; ++k) buf[k] = 0;
    if (i1 > i0) {
        for (int w = i0 >> 5; w <= (i1 - 1) >> 5; ++w) {
            uint32_t m = 0xffffffffu;
            if (w == (i0 >> 5)) m &= 0xffffffffu << (i0 & 31);
            if (w == ((i1 - 1) >> 5)) m &= 0xffffffffu >> (31 - ((i1 - 1) & 31));
            buf[2 * (w - lo_w)] |= m & L.linked_bits[w];
        }
    }
    for (int k = 0; k < n_extra; ++k) buf[2 * ((extra[k] >> 5) - lo_w)] |= 1u << (extra[k] & 31);
    for (int i = 0; i < n_ids; ++i) {
        const int v = ids[i];
        if (v >= 0 && v < V && L.linked[v]) {
            buf[2 * ((v >> 5) - lo_w)] |= 1u << (v & 31);
            buf[2 * ((v >> 5) - lo_w) + 1] |= 1u << (v & 31);
        }
    }
    return 0;
}

// ---- the whole of one decode key ------------------------------------------------------------------------------------------------
// Output pools of a launch (device memory; cursors bumped with atomics -- the order of allocation is arbitrary, nothing
// downstream depends on it).  A haplotype record in ht_pool: left, right, n_ids, n_exon, first candidate, ids...; its
// candidates are consecutive: the exon pieces in exon order, then the haplotype itself.
struct FePools {
    int32_t *ht_pool;      uint32_t ht_cap;      uint32_t *ht_cursor;
    uint16_t *cand_lo, *cand_nw; uint64_t *cand_key; uint32_t *cand_mask_off; uint32_t cand_cap; uint32_t *cand_cursor;
    uint32_t *mask_pool;   uint32_t mask_cap;    uint32_t *mask_cursor;
    // keep_trace (hgx_parse_opts): per decoded key the intermediates the reference's loop holds after identify_ambigious_diffs
    // (typing_core.py:1351-1384) -- what tests compare with the values recorded from the reference.  NULL = off.  A record:
    //   n_novel, n_cmp, cmp_left, cmp_right, n_left, n_right,
    //   n_novel x (variant type, pos, base | length)   novel variants in the order the walk created them (typing_core.py:1126-1164)
    //   n_cmp x (type | base << 8, pos, len, id)       cmp_list2
    //   per left alternative, then per right one: coord, n_ids, ids...
    int32_t *trace_pool;   uint32_t trace_cap;   uint32_t *trace_cursor;  uint32_t *key_trace_off;
};
#define FE_HT_HDR 5
#define FE_TRACE_HDR 6
#define FE_NO_TRACE 0xFFFFFFFFu

#if defined(__HIP_DEVICE_COMPILE__)
#define FE_ATOMIC_ADD(ptr, v) atomicAdd((ptr), (v))
#else
#define FE_ATOMIC_ADD(ptr, v) fe_host_add((ptr), (v))
static inline uint32_t fe_host_add(uint32_t *p, uint32_t v) { const uint32_t o = *p; *p = o + v; return o; }
#endif

// state: 1 = haplotypes follow (ht_off, n_ht), 2 = the record is dropped; return < 0 = decline
// (seq_copy: the key's SEQ bytes where the caller has put them -- the decode kernel stages them in LDS; nullptr = read them from text)
FE_HD inline int fe_key(const FeLocus &L, const FeParse &o, const FePile &P, const FeKey &K, const char *text, const FePools &pools,
                        uint8_t &state, uint32_t &ht_off, uint32_t &n_ht_out, const unsigned char *seq_copy = nullptr) {
    FeCmp cl[FE_MAX_CMP], tmp[FE_MAX_CMP];
    int n_cl = 0;
    state = 2;
    ht_off = 0;
    n_ht_out = 0;
    FeSeq seq_k = fe_seq_of(K, text);
    if (seq_copy) seq_k.p = seq_copy;
    int rc = fe_decode(L, o, P, K.pos, fe_cigar_of(K, text), seq_k, text + K.zs_off, (K.flags & FE_K_HAS_ZS) ? K.zs_len : 0,
                       text + K.md_off, (K.flags & FE_K_HAS_MD) ? K.md_len : 0, cl, n_cl, tmp);
    if (rc <= 0) return rc;
    int n_nov = 0;
    if (pools.trace_pool) {                                          // (tmp is free again: the novel variants, in creation order)
        for (int k = 0; k < n_cl; ++k) {
            const FeCmp c = cl[k];
            const int t = fe_type(c);
            if (t == FE_T_MATCH || c.id < L.V) continue;
            FeCmp &d = tmp[n_nov++];
            d.pos = c.pos;
            if (t == FE_T_MISMATCH) { d.type = FE_VAR_SINGLE; d.len = (int32_t)(unsigned char)fe_base(c); }
            else { d.type = t == FE_T_DELETION ? FE_VAR_DELETION : FE_VAR_INSERTION; d.len = c.len; }
        }
    }
    fe_cmp_list2(L, cl, n_cl);
    if (n_cl <= 0) return FE_FAIL(FE_E_ASSERT);
    FeSide lset, rset;
    int cleft, cright;
    rc = fe_ambiguous(L, cl, n_cl, cleft, cright, lset, rset);
    if (rc) return rc;
    if (pools.trace_pool) {
        uint32_t words = FE_TRACE_HDR + 3u * (uint32_t)n_nov + 4u * (uint32_t)n_cl + 2u * (uint32_t)(lset.n + rset.n) + (uint32_t)(lset.n_ids + rset.n_ids);
        const uint32_t t0 = FE_ATOMIC_ADD(pools.trace_cursor, words);
        if (t0 + words > pools.trace_cap) return FE_FAIL(FE_E_POOL);
        int32_t *w = pools.trace_pool + t0;
        w[0] = n_nov; w[1] = n_cl; w[2] = cleft; w[3] = cright; w[4] = lset.n; w[5] = rset.n;
        w += FE_TRACE_HDR;
        for (int k = 0; k < n_nov; ++k) { w[0] = tmp[k].type; w[1] = tmp[k].pos; w[2] = tmp[k].len; w += 3; }
        for (int k = 0; k < n_cl; ++k) { w[0] = cl[k].type; w[1] = cl[k].pos; w[2] = cl[k].len; w[3] = cl[k].id; w += 4; }
        for (int side = 0; side < 2; ++side) {
            const FeSide &sd = side ? rset : lset;
            for (int a = 0; a < sd.n; ++a) {
                w[0] = sd.coord[a]; w[1] = sd.cnt[a];
                for (int i = 0; i < sd.cnt[a]; ++i) w[2 + i] = sd.ids[sd.off[a] + i];
                w += 2 + sd.cnt[a];
            }
        }
        pools.key_trace_off[K.slot] = t0;
    }
    int mid[FE_MAX_MID], n_mid = 0;
    for (int k = cleft; k <= cright; ++k)
        if (fe_type(cl[k]) != FE_T_MATCH) { if (n_mid >= FE_MAX_MID) return FE_FAIL(FE_E_CAP); mid[n_mid++] = cl[k].id; }
    // pass A: the distinct haplotypes and the size of their records
    uint32_t n_ht = 0, total = 0;
    for (int l = 0; l < lset.n; ++l)
        for (int r = 0; r < rset.n; ++r) {
            const FeHtView h{&lset, &rset, mid, n_mid, l, r};
            bool dup = false;
            for (int l2 = 0; l2 <= l && !dup; ++l2)
                for (int r2 = 0; r2 < (l2 == l ? r : rset.n) && !dup; ++r2) dup = fe_ht_equal(FeHtView{&lset, &rset, mid, n_mid, l2, r2}, h);
            if (dup) continue;
            if (h.n_ids() > FE_MAX_IDS) return FE_FAIL(FE_E_CAP);
            if (h.left() > h.right()) return FE_FAIL(FE_E_PIECE);
            n_ht++;
            total += FE_HT_HDR + (uint32_t)h.n_ids();
        }
    const uint32_t base = FE_ATOMIC_ADD(pools.ht_cursor, total);
    if (base + total > pools.ht_cap) return FE_FAIL(FE_E_POOL);
    // pass B: records, exon pieces, masks
    uint32_t at = base;
    int ids[FE_MAX_IDS];
    for (int l = 0; l < lset.n; ++l)
        for (int r = 0; r < rset.n; ++r) {
            const FeHtView h{&lset, &rset, mid, n_mid, l, r};
            bool dup = false;
            for (int l2 = 0; l2 <= l && !dup; ++l2)
                for (int r2 = 0; r2 < (l2 == l ? r : rset.n) && !dup; ++r2) dup = fe_ht_equal(FeHtView{&lset, &rset, mid, n_mid, l2, r2}, h);
            if (dup) continue;
            const int n_ids = h.n_ids();
            for (int k = 0; k < n_ids; ++k) ids[k] = h.id(k);
            FeExonPiece ex[FE_MAX_EXON_PIECES];
            int n_ex = 0;
            if (L.base_kind == FE_BASE_HLA) {
                rc = fe_exon_pieces(L, h.left(), h.right(), ids, n_ids, ex, n_ex);
                if (rc) return rc;
            }
            const uint32_t c0 = FE_ATOMIC_ADD(pools.cand_cursor, (uint32_t)n_ex + 1);
            if (c0 + (uint32_t)n_ex + 1 > pools.cand_cap) return FE_FAIL(FE_E_POOL);
            int32_t *rec = pools.ht_pool + at;
            rec[0] = h.left(); rec[1] = h.right(); rec[2] = n_ids; rec[3] = n_ex; rec[4] = (int32_t)c0;
            for (int k = 0; k < n_ids; ++k) rec[FE_HT_HDR + k] = ids[k];
            at += FE_HT_HDR + (uint32_t)n_ids;
            for (int p = 0; p <= n_ex; ++p) {
                uint32_t buf[2 * FE_MAX_NW];
                int lo_w, nw;
                if (p < n_ex) rc = fe_piece_masks(L, ex[p].left, ex[p].right, ids + ex[p].i0, ex[p].i1 - ex[p].i0, lo_w, nw, buf);
                else rc = fe_piece_masks(L, h.left(), h.right(), ids, n_ids, lo_w, nw, buf);
                if (rc) return rc;
                const uint32_t m0 = FE_ATOMIC_ADD(pools.mask_cursor, 2u * (uint32_t)nw);
                if (m0 + 2u * (uint32_t)nw > pools.mask_cap) return FE_FAIL(FE_E_POOL);
                for (int k = 0; k < 2 * nw; ++k) pools.mask_pool[m0 + k] = buf[k];
                const uint32_t c = c0 + (uint32_t)p;
                pools.cand_lo[c] = (uint16_t)lo_w;
                pools.cand_nw[c] = (uint16_t)nw;
                pools.cand_key[c] = fe_piece_key((uint16_t)lo_w, (uint16_t)nw, buf);
                pools.cand_mask_off[c] = m0;
            }
        }
    state = 1;
    ht_off = base;
    n_ht_out = n_ht;
    return 0;
}

// ---- the pair protocol (typing_core.py:1238-1347, 1545-1587) over one run of records with equal read ids ---------------------
// rec_info[i] = decode slot | left mate << 30 | first record of its read id << 31, for the records that passed the filters, in
// stream order.  A run has at most three records (one left, one right, one unpaired: typing_core.py:857-872).
#define FE_REC_SLOT(x) ((x) & 0x3fffffffu)
#define FE_REC_LEFT(x) (((x) >> 30) & 1u)
#define FE_REC_HEAD(x) ((x) >> 31)
#define FE_MAX_PAIR_HT 64

FE_HD inline bool fe_rec_equal(const int32_t *a, const int32_t *b) {
    if (a[0] != b[0] || a[1] != b[1] || a[2] != b[2]) return false;
    for (int k = 0; k < a[2]; ++k) if (a[FE_HT_HDR + k] != b[FE_HT_HDR + k]) return false;
    return true;
}

// choose_pairs (typing_core.py:680-716; CODIS D18S51, the stream's LAST pair only: 1547-1552): of the left x right haplotype pairs keep
// those whose inner distance is closest to the sample's median one.  lh / rh = offsets into ht_pool (a record starts with left,
// right); the survivors stay in the order of their first appearance in the l-outer, r-inner walk (choose_pairs_rec of hgx_sam.cpp).
FE_HD inline void fe_choose_pairs(const int32_t *ht_pool, uint32_t *lh, int &n_l, uint32_t *rh, int &n_r, long long expected) {
    if (n_l == 0 || n_r == 0 || (n_l < 2 && n_r < 2)) return;
    uint32_t nl[FE_MAX_PAIR_HT], nr[FE_MAX_PAIR_HT];
    int c_l = 0, c_r = 0;
    long long best = -1;
    for (int a = 0; a < n_l; ++a)
        for (int b = 0; b < n_r; ++b) {
            const long long l_left = ht_pool[lh[a]], l_right = ht_pool[lh[a] + 1], r_left = ht_pool[rh[b]], r_right = ht_pool[rh[b] + 1];
            const long long inter = l_right < r_right ? r_left - l_right - 1 : l_left - r_right - 1;
            const long long cur = expected - inter < 0 ? inter - expected : expected - inter;
            if (best < 0 || cur < best) { best = cur; c_l = c_r = 0; }
            if (cur == best) {
                bool dup = false;
                for (int x = 0; x < c_l && !dup; ++x) dup = nl[x] == lh[a];
                if (!dup) nl[c_l++] = lh[a];
                dup = false;
                for (int x = 0; x < c_r && !dup; ++x) dup = nr[x] == rh[b];
                if (!dup) nr[c_r++] = rh[b];
            }
        }
    for (int x = 0; x < c_l; ++x) lh[x] = nl[x];
    for (int x = 0; x < c_r; ++x) rh[x] = nr[x];
    n_l = c_l;
    n_r = c_r;
}

// The union of a run's haplotypes, left mate's first (uni[] = offsets into ht_pool).  Returns the number of surviving records,
// < 0 = decline.  `choose`: this run is the stream's last pair of a CODIS D18S51 sample -- choose_pairs on the two mates' sets first.
FE_HD inline int fe_pair_union(const uint32_t *rec_info, uint32_t i, uint32_t n_rec, const uint8_t *state, const uint32_t *key_ht_off,
                               const uint32_t *key_n_ht, const int32_t *ht_pool, uint32_t *uni, int &n_uni, bool choose = false,
                               long long expected = -1) {
    uint32_t rh[FE_MAX_PAIR_HT];
    int n_l = 0, n_r = 0, n_surv = 0;
    for (uint32_t k = i; k < n_rec && (k == i || !FE_REC_HEAD(rec_info[k])); ++k) {
        const uint32_t slot = FE_REC_SLOT(rec_info[k]);
        if (state[slot] != 1) continue;
        n_surv++;
        const bool left = FE_REC_LEFT(rec_info[k]) != 0;
        uint32_t off = key_ht_off[slot];
        for (uint32_t t = 0; t < key_n_ht[slot]; ++t) {
            const int32_t *rec = ht_pool + off;
            uint32_t *dst = left ? uni : rh;
            int &nd = left ? n_l : n_r;
            bool dup = false;
            for (int x = 0; x < nd && !dup; ++x) dup = fe_rec_equal(ht_pool + dst[x], rec);
            if (!dup) {
                if (nd >= FE_MAX_PAIR_HT) return FE_FAIL(FE_E_CAP);
                dst[nd++] = off;
            }
            off += FE_HT_HDR + (uint32_t)rec[2];
        }
    }
    if (choose) fe_choose_pairs(ht_pool, uni, n_l, rh, n_r, expected);
    n_uni = n_l;
    for (int x = 0; x < n_r; ++x) {
        bool dup = false;
        for (int y = 0; y < n_uni && !dup; ++y) dup = fe_rec_equal(ht_pool + uni[y], ht_pool + rh[x]);
        if (!dup) {
            if (n_uni >= FE_MAX_PAIR_HT) return FE_FAIL(FE_E_CAP);
            uni[n_uni++] = rh[x];
        }
    }
    return n_surv;
}

// ---- get_mpileup (typing_common.py:1059-1134) over one distinct key, weighted by its group size -----------------------------
// add(cell, weight) with cell = position * 6 + slot (A C G T N D).  `lane` / `n_lanes`: the bases of an M op are dealt out to
// the lanes of a wavefront (0 / 1 = one walker does them all).  Returns 0, or < 0 = decline.
template <class Add>
FE_HD inline int fe_pileup_key(const FeKey &K, const char *text, int n_ref, int lane, int n_lanes, Add add) {
    const FeSeq seq = fe_seq_of(K, text);
    FeCigar cg = fe_cigar_of(K, text);
    const uint32_t w = K.n_pile;
    int rp = 0, gp = K.pos;
    if (!cg.bin && cg.n > 0) {
        const char c0 = (char)cg.p[0];
        if (c0 == '+' || c0 == '-' || c0 == ' ' || (c0 >= 9 && c0 <= 13)) return FE_FAIL(FE_E_CIGAR);   // forms strtol reads: the host's business
    }
    for (;;) {
        char op;
        int ilen;
        const int r = cg.next(op, ilen);
        if (r <= 0) {
            if (r < 0 && cg.at < cg.n) {
                const char c0 = (char)cg.p[cg.at];
                if (c0 == '+' || c0 == '-' || c0 == ' ' || (c0 >= 9 && c0 <= 13)) return FE_FAIL(FE_E_CIGAR);
            }
            break;                                                   // (the host stops at what it cannot read, silently)
        }
        const long len = ilen;
        if (op == 'M') {
            long lim = len < (long)n_ref - gp ? len : (long)n_ref - gp;
            if (lim > 0 && (uint64_t)(rp + lim) > (uint64_t)K.seq_len) return FE_FAIL(FE_E_SHORT);
            for (long j = lane; j < lim; j += n_lanes) {
                const char b = seq.at((int)(rp + j));
                const int s = b == 'A' ? 0 : b == 'C' ? 1 : b == 'G' ? 2 : b == 'T' ? 3 : 4;
                add((uint32_t)((gp + j) * 6 + s), w);
            }
        } else if (op == 'D') {
            const long lim = len < (long)n_ref - gp ? len : (long)n_ref - gp;
            for (long j = lane; j < lim; j += n_lanes) add((uint32_t)((gp + j) * 6 + 5), w);
        }
        if (op == 'M' || op == 'N' || op == 'D') gp += (int)len;
        if (op == 'M' || op == 'I' || op == 'S') rp += (int)len;
    }
    return 0;
}
// nt_set of one position from its six counts (typing_common.py:1124-1134)
FE_HD inline uint8_t fe_nt_set(const uint32_t *c) {
    const uint64_t tot = (uint64_t)c[0] + c[1] + c[2] + c[3] + c[4] + c[5];
    int m = 0;
    if (tot >= 20)
        for (int k = 0; k < 4; ++k)
            if ((double)c[k] >= (double)tot * 0.2 || c[k] >= 7) m |= 1 << k;
    return (uint8_t)m;
}

// =====================================================================================================================================
// The RECORD stage on the device (row 8a-1): one lane per record of the name-ordered stream -- fields (typing_core.py:800-841),
// record filters (typing_core.py:815-872) and the grouping of the records by decode key -- so that the host only reads (and, for
// BAM, inflates and name-sorts) the file.  Restates split_line / note_tag / split_bam / filter_records / group_records of
// hgx_sam.cpp; every input they treat specially (blanks or CRs inside a line, tags whose value int() rejects, fewer than eleven
// fields, float / text typed NM tags, a CG tag, ...) declines.
// =====================================================================================================================================
#define FE_R_HAS_NM 1
#define FE_R_HAS_NH 2
#define FE_R_HAS_ZS 4
#define FE_R_HAS_MD 8
#define FE_R_BIN 16                  // BAM record: binary CIGAR, packed SEQ
#define FE_R_FAILED 32               // the record could not be taken apart (the call declines): every offset and length is zero
#define FE_R_YT_CP 64                // the LAST YT tag of the record spells "CP" (get_pair_interdist, typing_common.py:1224-1231)
struct FeRec {
    uint32_t qname_off;
    uint16_t id_len;                 // read id = QNAME, or QNAME up to the first '|' in simulation mode (typing_core.py:808-809)
    uint16_t bits;
    int32_t flag, pos;               // FLAG, POS (1-based, as in the file)
    int32_t nm, nh;                  // saturated to int32
    uint32_t cigar_off, seq_off, zs_off, md_off, seq_len;
    uint16_t cigar_len, zs_len, md_len;
    uint16_t task;                   // sample of a many-task batch (0 otherwise): keys, read ids and pairs never cross tasks
    uint64_t key;                    // hash of the decode key (task, pos, cigar, seq, Zs, MD)
};

FE_HD inline bool fe_py_int_ok(const unsigned char *t, int n) {      // would Python's int(text) take it?  (hgx_sam.cpp py_int_ok)
    int i = 0;
    if (i < n && (t[i] == '+' || t[i] == '-')) ++i;
    if (i >= n || t[i] < '0' || t[i] > '9') return false;
    for (; i < n; ++i) {
        if (t[i] >= '0' && t[i] <= '9') continue;
        if (t[i] == '_' && i + 1 < n && t[i + 1] >= '0' && t[i + 1] <= '9' && t[i - 1] != '_') continue;
        return false;
    }
    return true;
}
// strtol(text, 0, 10) of a token that fe_py_int_ok accepted (sign, digits; stops at an underscore), saturated to int32
FE_HD inline int32_t fe_strtol32(const unsigned char *t, int n) {
    int i = 0;
    bool neg = false;
    if (i < n && (t[i] == '+' || t[i] == '-')) { neg = t[i] == '-'; ++i; }
    long long v = 0;
    for (; i < n && t[i] >= '0' && t[i] <= '9'; ++i) { v = v * 10 + (t[i] - '0'); if (v > 0x7fffffffll) v = 0x7fffffffll; }
    return (int32_t)(neg ? -v : v);
}
// (8 bytes per load: the compiler emits one unaligned dwordx2 load on gfx950, one mov on x86; both little endian)
FE_HD inline uint64_t fe_load8(const unsigned char *p) { uint64_t w; __builtin_memcpy(&w, p, 8); return w; }
// (The streaming loops below take 32 bytes per round, the four loads issued together: a lane that comes back for the next 8 bytes of
// a line finds it evicted -- 2 048 lanes per CU stream through lines of their own, the L1 holds 128 -- so 8 bytes per round were
// one L2 request per load: 50 requests per record in k_fe_group_flags.)
FE_HD inline uint64_t fe_hash_bytes(const unsigned char *p, int n, uint64_t h) {
    int i = 0;
    for (; i + 32 <= n; i += 32) {
        const uint64_t w0 = fe_load8(p + i), w1 = fe_load8(p + i + 8), w2 = fe_load8(p + i + 16), w3 = fe_load8(p + i + 24);
        h = (h ^ w0) * 0x9E3779B97F4A7C15ull; h ^= h >> 32;
        h = (h ^ w1) * 0x9E3779B97F4A7C15ull; h ^= h >> 32;
        h = (h ^ w2) * 0x9E3779B97F4A7C15ull; h ^= h >> 32;
        h = (h ^ w3) * 0x9E3779B97F4A7C15ull; h ^= h >> 32;
    }
    for (; i + 8 <= n; i += 8) {
        h = (h ^ fe_load8(p + i)) * 0x9E3779B97F4A7C15ull;
        h ^= h >> 32;
    }
    uint64_t w = 0;
    for (int k = 0; i + k < n; ++k) w |= (uint64_t)p[i + k] << (8 * k);
    h = (h ^ w ^ ((uint64_t)n << 56)) * 0xD6E8FEB86659FD93ull;
    return h ^ (h >> 29);
}
// hash of a record's decode key; for a packed SEQ the unused low nibble of an odd length's last byte is left out
FE_HD inline uint64_t fe_rec_key(const FeRec &r, const char *text) {
    const unsigned char *t = (const unsigned char *)text;
    uint64_t h = fe_mix64(0x243F6A8885A308D3ull ^ (uint32_t)r.pos ^ ((uint64_t)r.seq_len << 32));
    h = fe_mix64(h + 0x9e3779b97f4a7c15ull * (uint64_t)(r.task + 1));
    if (r.bits & FE_R_BIN) {
        h = fe_hash_bytes(t + r.cigar_off, 4 * (int)r.cigar_len, h);
        const int full = (int)(r.seq_len / 2);
        h = fe_hash_bytes(t + r.seq_off, full, h);
        if (r.seq_len & 1) h = fe_mix64(h ^ (uint64_t)(t[r.seq_off + full] >> 4));
    } else {
        h = fe_hash_bytes(t + r.cigar_off, r.cigar_len, h);
        h = fe_hash_bytes(t + r.seq_off, (int)r.seq_len, h);
    }
    h = fe_hash_bytes(t + r.zs_off, r.zs_len, h ^ ((r.bits & FE_R_HAS_ZS) ? 1 : 0));
    h = fe_hash_bytes(t + r.md_off, r.md_len, h ^ ((r.bits & FE_R_HAS_MD) ? 2 : 0));
    h = fe_mix64(h);
    return h == ~0ull ? 0 : h;                                      // (all ones = the hash table's empty slot)
}
FE_HD inline bool fe_bytes_equal(const unsigned char *a, const unsigned char *b, int n) {
    int i = 0;
    for (; i + 32 <= n; i += 32) {
        const uint64_t d = (fe_load8(a + i) ^ fe_load8(b + i)) | (fe_load8(a + i + 8) ^ fe_load8(b + i + 8)) |
                           (fe_load8(a + i + 16) ^ fe_load8(b + i + 16)) | (fe_load8(a + i + 24) ^ fe_load8(b + i + 24));
        if (d) return false;
    }
    for (; i + 8 <= n; i += 8) if (fe_load8(a + i) != fe_load8(b + i)) return false;
    for (; i < n; ++i) if (a[i] != b[i]) return false;
    return true;
}
FE_HD inline bool fe_rec_same_key(const FeRec &a, const FeRec &b, const char *text) {      // same_decode_key of hgx_sam.cpp
    const unsigned char *t = (const unsigned char *)text;
    if ((a.bits | b.bits) & FE_R_FAILED) return true;               // (nothing of a failed record may be read; the call has declined)
    if (a.task != b.task || a.pos != b.pos || a.seq_len != b.seq_len || a.cigar_len != b.cigar_len || a.zs_len != b.zs_len || a.md_len != b.md_len) return false;
    if (((a.bits ^ b.bits) & (FE_R_HAS_ZS | FE_R_HAS_MD | FE_R_BIN)) != 0) return false;
    if (a.bits & FE_R_BIN) {
        if (!fe_bytes_equal(t + a.cigar_off, t + b.cigar_off, 4 * (int)a.cigar_len)) return false;
        const int full = (int)(a.seq_len / 2);
        if (!fe_bytes_equal(t + a.seq_off, t + b.seq_off, full)) return false;
        if ((a.seq_len & 1) && (t[a.seq_off + full] >> 4) != (t[b.seq_off + full] >> 4)) return false;
    } else {
        if (!fe_bytes_equal(t + a.cigar_off, t + b.cigar_off, a.cigar_len)) return false;
        if (!fe_bytes_equal(t + a.seq_off, t + b.seq_off, (int)a.seq_len)) return false;
    }
    return fe_bytes_equal(t + a.zs_off, t + b.zs_off, a.zs_len) && fe_bytes_equal(t + a.md_off, t + b.md_off, a.md_len);
}

// a record of the name-ordered stream handed to the record stage: first byte (after block_size for BAM), length, sample
struct FeLine { uint32_t off, len, task; };

// one line of SAM text (without its line end) -> FeRec.  The tab-only split of split_line; anything else declines.  The line is
// scanned eight bytes at a time (a word without tab, blank or CR -- most of SEQ and QUAL -- is skipped whole); `text_bytes` = size
// of the buffer, so that no load reaches past it.
FE_HD inline int fe_parse_text_record(const char *text, size_t text_bytes, uint32_t off, uint32_t len, bool simulation, uint32_t task, FeRec &r) {
    const unsigned char *line = (const unsigned char *)text + off;
    r.bits = 0;
    r.task = (uint16_t)task;
    r.nm = r.nh = 0;
    r.zs_off = r.md_off = off;
    r.zs_len = r.md_len = 0;
    uint32_t col_at[11], col_len[11];
    int nc = 0;
    uint32_t p = 0, tok = 0;
    const uint32_t safe = (size_t)off + len + 8 <= text_bytes ? len : (len >= 8 ? len - 8 : 0);      // words may start below `safe`
    const uint64_t ones = 0x0101010101010101ull, highs = 0x8080808080808080ull;
    for (;;) {
        bool at_end = p >= len;
        // 32 bytes at a time while none of them is a tab, blank or CR (SEQ and QUAL are three quarters of a line): the four loads go
        // out together, one round trip per 32 bytes instead of one per 8
        while (p + 32 <= safe && p + 32 <= len) {
            const uint64_t w0 = fe_load8(line + p), w1 = fe_load8(line + p + 8), w2 = fe_load8(line + p + 16), w3 = fe_load8(line + p + 24);
            uint64_t any = 0;
            const uint64_t ws[4] = {w0, w1, w2, w3};
            for (int k = 0; k < 4; ++k) {
                const uint64_t xt = ws[k] ^ (ones * 0x09), xs = ws[k] ^ (ones * 0x20), xr = ws[k] ^ (ones * 0x0d);
                any |= (((xt - ones) & ~xt) | ((xs - ones) & ~xs) | ((xr - ones) & ~xr)) & highs;
            }
            if (any) break;
            p += 32;
        }
        at_end = p >= len;
        if (!at_end && p < safe) {
            const uint64_t w = fe_load8(line + p);
            const uint64_t xt = w ^ (ones * 0x09), xs = w ^ (ones * 0x20), xr = w ^ (ones * 0x0d);
            const uint64_t hit = (((xt - ones) & ~xt) | ((xs - ones) & ~xs) | ((xr - ones) & ~xr)) & highs;
            uint32_t clean = 8;                                     // bytes of the word before the first tab / blank / CR
            if (hit) {
                uint32_t k = 0;
                uint64_t m = hit;
                while (!(m & 0xff)) { m >>= 8; ++k; }
                clean = k;
            }
            if (p + clean > len) clean = len - p;                   // (bytes past the line's end are somebody else's)
            p += clean;
            if (clean == 8 || p >= len) { if (p < len) continue; at_end = true; }
        }
        unsigned char c = 0;
        if (!at_end) {
            c = line[p];
            if (c == ' ' || c == '\r') return FE_FAIL(FE_E_ASSERT);       // the reference's split() cuts there too: the host's business
            if (c != '\t') { ++p; continue; }
        }
        // a token ends at p (a tab, or the line's end)
        if (p > tok) {
            const uint32_t tl = p - tok;
            if (nc < 11) { col_at[nc] = tok; col_len[nc] = tl; nc++; }
            else {                                                                  // note_tag
                const unsigned char *t = line + tok;
                if (tl < 5) {
                    if (tl >= 2 && t[0] == 'Z' && t[1] == 's') { r.bits |= FE_R_HAS_ZS; r.zs_off = off + p; r.zs_len = 0; }
                    else if (tl >= 2 && t[0] == 'M' && t[1] == 'D') { r.bits |= FE_R_HAS_MD; r.md_off = off + p; r.md_len = 0; }
                    else if (tl >= 2 && t[0] == 'N' && (t[1] == 'M' || t[1] == 'H')) return FE_FAIL(FE_E_ASSERT);   // int('') raises
                    else if (tl >= 2 && t[0] == 'Y' && t[1] == 'T') r.bits &= (uint16_t)~FE_R_YT_CP;                 // YT = col[5:] = ""
                } else if (t[0] == 'Y' && t[1] == 'T') {
                    if (tl == 7 && t[5] == 'C' && t[6] == 'P') r.bits |= FE_R_YT_CP; else r.bits &= (uint16_t)~FE_R_YT_CP;
                } else if (t[0] == 'Z' && t[1] == 's') {
                    if (tl - 5 > 65535) return FE_FAIL(FE_E_CAP);
                    r.bits |= FE_R_HAS_ZS; r.zs_off = off + tok + 5; r.zs_len = (uint16_t)(tl - 5);
                } else if (t[0] == 'M' && t[1] == 'D') {
                    if (tl - 5 > 65535) return FE_FAIL(FE_E_CAP);
                    r.bits |= FE_R_HAS_MD; r.md_off = off + tok + 5; r.md_len = (uint16_t)(tl - 5);
                } else if (t[0] == 'N' && (t[1] == 'M' || t[1] == 'H')) {
                    if (!fe_py_int_ok(t + 5, (int)tl - 5)) return FE_FAIL(FE_E_ASSERT);
                    const int32_t v = fe_strtol32(t + 5, (int)tl - 5);
                    if (t[1] == 'M') { r.bits |= FE_R_HAS_NM; r.nm = v; }
                    else { r.bits |= FE_R_HAS_NH; r.nh = v; }
                }
            }
        }
        if (at_end) break;
        ++p;
        tok = p;
    }
    if (nc < 11) return FE_FAIL(FE_E_ASSERT);
    if (!fe_py_int_ok(line + col_at[1], (int)col_len[1]) || !fe_py_int_ok(line + col_at[3], (int)col_len[3])) return FE_FAIL(FE_E_ASSERT);
    if (col_len[1] > 9 || col_len[3] > 10 || col_len[5] > 65535 || col_len[0] > 65535 || col_len[9] > (1u << 24)) return FE_FAIL(FE_E_CAP);
    r.qname_off = off + col_at[0];
    uint32_t idl = col_len[0];
    if (simulation)
        for (uint32_t k = 0; k < col_len[0]; ++k) if (line[col_at[0] + k] == '|') { idl = k; break; }
    r.id_len = (uint16_t)idl;
    r.flag = fe_strtol32(line + col_at[1], (int)col_len[1]);
    r.pos = fe_strtol32(line + col_at[3], (int)col_len[3]);
    r.cigar_off = off + col_at[5];
    r.cigar_len = (uint16_t)col_len[5];
    r.seq_off = off + col_at[9];
    r.seq_len = col_len[9];
    r.key = fe_rec_key(r, text);
    return 0;
}

// one BAM record (rec_off = the record's first byte after block_size, len = block_size) -> FeRec: split_bam of hgx_sam.cpp
FE_HD inline uint32_t fe_ld32(const unsigned char *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
FE_HD inline int fe_parse_bam_record(const char *text, uint32_t rec_off, uint32_t len, bool simulation, uint32_t task, FeRec &rec) {
    const unsigned char *r = (const unsigned char *)text + rec_off;
    if (len < 32) return FE_FAIL(FE_E_ASSERT);
    const int32_t pos0 = (int32_t)fe_ld32(r + 4);
    const uint32_t l_rn = r[8], n_cig = (uint32_t)r[12] | ((uint32_t)r[13] << 8), flag = (uint32_t)r[14] | ((uint32_t)r[15] << 8);
    const int32_t l_seq = (int32_t)fe_ld32(r + 16);
    if (l_rn == 0) return FE_FAIL(FE_E_ASSERT);
    if (flag & 0x4) {
        // An unmapped record (a mate placed at its partner's position; routine in a BAM over a region): the filters drop it on
        // its flag before anything else of it is looked at (typing_core.py:815-820, typing_common.py:1076), so only its name and
        // flag are taken -- CIGAR "*" / SEQ "*" need not decline the call.
        if (32 + (size_t)l_rn > len || r[32 + l_rn - 1] != 0) return FE_FAIL(FE_E_ASSERT);
        rec.bits = FE_R_BIN;
        rec.task = (uint16_t)task;
        rec.nm = rec.nh = 0;
        rec.cigar_off = rec.seq_off = rec.zs_off = rec.md_off = rec_off;
        rec.seq_len = 0;
        rec.cigar_len = rec.zs_len = rec.md_len = 0;
        rec.qname_off = rec_off + 32;
        uint32_t idl = l_rn - 1;
        if (simulation)
            for (uint32_t k = 0; k + 1 < l_rn; ++k) if (r[32 + k] == '|') { idl = k; break; }
        rec.id_len = (uint16_t)idl;
        rec.flag = (int32_t)flag;
        rec.pos = pos0 + 1;
        rec.key = fe_rec_key(rec, text);
        return 0;
    }
    if (l_seq <= 0) return FE_FAIL(FE_E_ASSERT);                          // (a mapped record without SEQ spells "*": the host's business)
    size_t q = 32 + (size_t)l_rn;
    const size_t cig_at = q;
    q += 4ull * n_cig;
    const size_t seq_at = q;
    q += (size_t)(l_seq + 1) / 2 + (size_t)l_seq;
    if (q > len || r[32 + l_rn - 1] != 0) return FE_FAIL(FE_E_ASSERT);
    if (n_cig == 0) return FE_FAIL(FE_E_ASSERT);                          // ("*")
    rec.bits = FE_R_BIN;
    rec.task = (uint16_t)task;
    rec.nm = rec.nh = 0;
    rec.zs_off = rec.md_off = rec_off;
    rec.zs_len = rec.md_len = 0;
    while (q + 3 <= len) {
        const char t0 = (char)r[q], t1 = (char)r[q + 1], t = (char)r[q + 2];
        q += 3;
        size_t sz = 0;
        if (t == 'A' || t == 'c' || t == 'C') sz = 1;
        else if (t == 's' || t == 'S') sz = 2;
        else if (t == 'i' || t == 'I' || t == 'f') sz = 4;
        else if (t == 'Z' || t == 'H') {
            // the NUL that ends the text, eight bytes per look (a byte per look was a dependent load per character of every MD / Zs)
            size_t e = q;
            bool hit = false;
            while (!hit && e + 8 <= len) {
                const uint64_t w = fe_load8(r + e);
                const uint64_t z = ~(((w & 0x7F7F7F7F7F7F7F7FULL) + 0x7F7F7F7F7F7F7F7FULL) | w) & 0x8080808080808080ULL;     // 0x80 in the zero bytes
                if (z) { e += (size_t)(__builtin_ctzll(z) >> 3); hit = true; }
                else e += 8;
            }
            while (!hit && e < len && r[e] != 0) ++e;
            if (e >= len) return FE_FAIL(FE_E_ASSERT);
            sz = e - q + 1;
        } else if (t == 'B') {
            if (q + 5 > len) return FE_FAIL(FE_E_ASSERT);
            const char st = (char)r[q];
            const uint32_t cnt = fe_ld32(r + q + 1);
            const size_t w = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : (st == 'i' || st == 'I' || st == 'f') ? 4 : 0;
            if (!w) return FE_FAIL(FE_E_ASSERT);
            sz = 5 + w * (size_t)cnt;
            if (t0 == 'C' && t1 == 'G') return FE_FAIL(FE_E_ASSERT);        // the real-CIGAR rule: host
        } else return FE_FAIL(FE_E_ASSERT);
        if (q + sz > len) return FE_FAIL(FE_E_ASSERT);
        const unsigned char *v = r + q;
        const bool is_text = t == 'Z' || t == 'H';
        if (t0 == 'Z' && t1 == 's') {
            rec.bits |= FE_R_HAS_ZS;
            if (is_text) { if (sz - 1 > 65535) return FE_FAIL(FE_E_CAP); rec.zs_off = rec_off + (uint32_t)q; rec.zs_len = (uint16_t)(sz - 1); }
            else { rec.zs_off = rec_off; rec.zs_len = 0; }
        } else if (t0 == 'M' && t1 == 'D') {
            rec.bits |= FE_R_HAS_MD;
            if (is_text) { if (sz - 1 > 65535) return FE_FAIL(FE_E_CAP); rec.md_off = rec_off + (uint32_t)q; rec.md_len = (uint16_t)(sz - 1); }
            else { rec.md_off = rec_off; rec.md_len = 0; }
        } else if (t0 == 'N' && (t1 == 'M' || t1 == 'H')) {
            long long x;
            if (t == 'c') x = (signed char)v[0];
            else if (t == 'C') x = v[0];
            else if (t == 's') x = (short)((uint32_t)v[0] | ((uint32_t)v[1] << 8));
            else if (t == 'S') x = (long long)((uint32_t)v[0] | ((uint32_t)v[1] << 8));
            else if (t == 'i') x = (int32_t)fe_ld32(v);
            else if (t == 'I') x = (long long)fe_ld32(v);
            else return FE_FAIL(FE_E_ASSERT);                              // float / text typed: read through their spelling on the host
            if (x > 0x7fffffffll) x = 0x7fffffffll;
            if (t1 == 'M') { rec.bits |= FE_R_HAS_NM; rec.nm = (int32_t)x; }
            else { rec.bits |= FE_R_HAS_NH; rec.nh = (int32_t)x; }
        } else if (t0 == 'Y' && t1 == 'T') {
            if (is_text && sz == 3 && v[0] == 'C' && v[1] == 'P') rec.bits |= FE_R_YT_CP; else rec.bits &= (uint16_t)~FE_R_YT_CP;
        }
        q += sz;
    }
    for (uint32_t k = 0; k < n_cig; ++k) if ((r[cig_at + 4 * k] & 15) >= 9) return FE_FAIL(FE_E_CIGAR);
    rec.qname_off = rec_off + 32;
    uint32_t idl = l_rn - 1;
    if (simulation)
        for (uint32_t k = 0; k + 1 < l_rn; ++k) if (r[32 + k] == '|') { idl = k; break; }
    rec.id_len = (uint16_t)idl;
    rec.flag = (int32_t)flag;
    rec.pos = pos0 + 1;
    rec.cigar_off = rec_off + (uint32_t)cig_at;
    rec.cigar_len = (uint16_t)n_cig;
    rec.seq_off = rec_off + (uint32_t)seq_at;
    rec.seq_len = (uint32_t)l_seq;
    rec.key = fe_rec_key(rec, text);
    return 0;
}

FE_HD inline bool fe_same_read_id(const FeRec &a, const FeRec &b, const char *text) {
    return a.task == b.task && a.id_len == b.id_len &&
           fe_bytes_equal((const unsigned char *)text + a.qname_off, (const unsigned char *)text + b.qname_off, a.id_len);
}

// ---- get_pair_interdist (typing_common.py:1187-1265; CODIS D18S51 only) over the records of the stream ---------------------------
// A record counts iff it is aligned, has NH <= 1 and its YT tag says "CP"; runs of such records with one read id that hold exactly
// two of them and are FOLLOWED by another such record (the reference appends a run's distance when the id changes: the stream's last
// run never gets there) give one inner distance each.  The sample's expected distance = element len / 2 of the sorted list, taken
// from a histogram here: bin 0 = below -FE_INTERDIST_HALF, bin 1 + d + FE_INTERDIST_HALF = distance d, the last bin = above
// (HGX_INTERDIST_* of include/hgx.h: the form the shards of a locus exchange).
#define FE_INTERDIST_HALF 65536
#define FE_INTERDIST_BINS (2 * FE_INTERDIST_HALF + 2)
FE_HD inline bool fe_rec_in_interdist(const FeRec &f) {
    return !(f.bits & FE_R_FAILED) && !(f.flag & 0x4) && (f.bits & FE_R_HAS_NH) && f.nh <= 1 && (f.bits & FE_R_YT_CP);
}
// [left, right] of the record on the backbone: POS and POS + the M / N / D lengths - 1.  < 0: a CIGAR text the host reads its own way.
FE_HD inline int fe_rec_span(const FeRec &f, const char *text, long long &left, long long &right) {
    FeCigar cg;
    cg.p = (const unsigned char *)text + f.cigar_off;
    cg.n = f.cigar_len;
    cg.at = 0;
    cg.bin = (f.bits & FE_R_BIN) != 0;
    long long r = f.pos;
    for (;;) {
        char op;
        int len;
        const int k = cg.next(op, len);
        if (k < 0) return FE_FAIL(FE_E_CIGAR);
        if (k == 0) break;
        if (op == 'M' || op == 'N' || op == 'D') r += len;
    }
    left = f.pos;
    right = r - 1;
    return 0;
}
FE_HD inline uint32_t fe_interdist_bin(long long d) {
    return d < -(long long)FE_INTERDIST_HALF ? 0u : d > (long long)FE_INTERDIST_HALF - 1 ? (uint32_t)FE_INTERDIST_BINS - 1u : (uint32_t)(1 + d + FE_INTERDIST_HALF);
}
// a (the run's first counted record), b (its second): the distance typing_common.py:1243-1251 appends
FE_HD inline int fe_interdist_of(const FeRec &a, const FeRec &b, const char *text, long long &dist) {
    long long l1, r1, l2, r2;
    int rc = fe_rec_span(a, text, l1, r1);
    if (rc) return rc;
    rc = fe_rec_span(b, text, l2, r2);
    if (rc) return rc;
    dist = l1 <= l2 ? l2 - r1 - 1 : l1 - r2 - 1;
    return 0;
}

// record filters (typing_core.py:815-872; filter_records of hgx_sam.cpp).  `head[i]` = record i opens a group of equal read ids.
// A record passes up to the mate rule iff it is aligned, inside the locus, has its tags, few enough edits, one hit and is
// concordant (or discordant pairs count); of the passing records of a group the FIRST left mate, the first right mate and the
// first unpaired one are kept.  Returns 1 kept, 0 dropped, < 0 where the reference raises (the host says how).
struct FeFilter { int32_t num_editdist, allow_discordant, base_locus; };
FE_HD inline int fe_rec_passes(const FeRec &f, const FeFilter &o) {
    if (f.pos - (o.base_locus + 1) < 0) return 0;
    if (f.flag & 0x4) return 0;
    if (!(f.bits & FE_R_HAS_NM) || !(f.bits & FE_R_HAS_NH)) return FE_FAIL(FE_E_ASSERT);     // quirk Q8
    if (f.nm > o.num_editdist) return 0;
    if (f.nh > 1) return 0;
    if (!o.allow_discordant && !(f.flag & 0x2)) return 0;
    if (!(f.flag & 0x40) && !(f.flag & 0x80) && !o.allow_discordant) return FE_FAIL(FE_E_ASSERT);    // assert allow_discordant
    return 1;
}
FE_HD inline int fe_rec_side(const FeRec &f) { return (f.flag & 0x40) ? 0 : (f.flag & 0x80) ? 1 : 2; }
FE_HD inline int fe_rec_kept(const FeRec *recs, const uint8_t *head, uint32_t i, const FeFilter &o) {
    const int p = fe_rec_passes(recs[i], o);
    if (p <= 0) return p;
    const int side = fe_rec_side(recs[i]);
    for (uint32_t j = i; j > 0 && !head[j]; ) {
        --j;
        const int pj = fe_rec_passes(recs[j], o);
        if (pj < 0) return pj;
        if (pj == 1 && fe_rec_side(recs[j]) == side) return 0;
        if (head[j]) break;
    }
    return 1;
}
FE_HD inline bool fe_rec_in_pileup(const FeRec &f, const FeFilter &o) {      // typing_common.py:1076-1090
    return !(f.flag & 0x4) && f.pos - (o.base_locus + 1) >= 0 && (o.allow_discordant || (f.flag & 0x2));
}
