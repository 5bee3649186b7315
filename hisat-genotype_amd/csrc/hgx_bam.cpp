// hgx_bam.cpp -- alignment ingestion without samtools, host side of libhgx.
//
// The reference pipes `samtools view <file> [chr:left-right]` through `sort -k1,1 -s` (typing_core.py:436-468) and feeds the
// text to its loop.  hgx_read_alignments produces exactly that record stream from a SAM text file or a BAM file:
//   * BGZF (SAM/BAM specification v1, section 4.1): blocks are independent deflate streams -> inflated in parallel (zlib),
//     CRC32 + ISIZE checked;
//   * BAM records (section 4.2) -> SAM text lines, in parallel over record ranges: the eleven mandatory fields and the tags of
//     types A c C s S i I f Z H B (floats as %g, like samtools);
//   * header lines dropped; optional region list in samtools syntax ("name" or "name:left-right", 1-based inclusive): the records
//     that OVERLAP a region (reference span from the CIGAR, as `samtools view file r1 r2` selects them), region after region;
//   * name grouping: STABLE sort of the records by QNAME, bytewise (LC_ALL=C `sort -k1,1 -s`): parallel chunk sorts + merges.
// hisat-genotype_amd/bamio.py is the pure-Python statement of the same formats; tests compare the two byte for byte.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "hgx_internal.hpp"

namespace {

template <class F>
void par_for(int n_threads, size_t n, F fn) {      // fn(thread, begin, end) over [0, n) in contiguous ranges, on the worker pool
    hgx_par_ranges(n_threads, n, fn);
}

inline uint16_t rd16(const unsigned char *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t rd32(const unsigned char *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline int32_t rdi32(const unsigned char *p) { return (int32_t)rd32(p); }

// byte buffer without value initialisation (hundreds of MB are overwritten right after allocation)
struct Bytes {
    unsigned char *p = nullptr;
    size_t n = 0;
    Bytes() = default;
    Bytes(const Bytes &) = delete;
    Bytes &operator=(const Bytes &) = delete;
    ~Bytes() { hgx_host_free(p); }
    void alloc(size_t k) { hgx_host_free(p); p = (unsigned char *)hgx_host_alloc(k ? k : 1); n = k; }
    void release() { hgx_host_free(p); p = nullptr; n = 0; }
    void swap(Bytes &o) { std::swap(p, o.p); std::swap(n, o.n); }
    unsigned char *data() { return p; }
    const unsigned char *data() const { return p; }
    size_t size() const { return n; }
    unsigned char &operator[](size_t i) { return p[i]; }
    const unsigned char &operator[](size_t i) const { return p[i]; }
};

struct Block { size_t in_off, in_len, out_off, out_len; uint32_t crc; };

// libdeflate, if the system has it (no headers needed: four entry points of its stable C API, resolved with dlopen): its
// DEFLATE decoder and CRC-32 are 2-3x faster than zlib's, and inflating the BGZF blocks is the largest BAM-only share of the
// CPU seconds of a file -> result call.  zlib remains the fallback (and HGX_NO_LIBDEFLATE=1 forces it; the tests run both).
struct FastInflate {
    void *lib = nullptr;
    void *(*alloc)() = nullptr;
    int (*decompress)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;      // 0 = LIBDEFLATE_SUCCESS
    void (*release)(void *) = nullptr;
    uint32_t (*crc32)(uint32_t, const void *, size_t) = nullptr;
};
const FastInflate *fast_inflate() {
    static const FastInflate f = [] {
        FastInflate x;
        for (const char *name : {"libdeflate.so.0", "libdeflate.so"}) {
            x.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (x.lib) break;
        }
        if (!x.lib) return x;
        x.alloc = (void *(*)())dlsym(x.lib, "libdeflate_alloc_decompressor");
        x.decompress = (int (*)(void *, const void *, size_t, void *, size_t, size_t *))dlsym(x.lib, "libdeflate_deflate_decompress");
        x.release = (void (*)(void *))dlsym(x.lib, "libdeflate_free_decompressor");
        x.crc32 = (uint32_t (*)(uint32_t, const void *, size_t))dlsym(x.lib, "libdeflate_crc32");
        if (!x.alloc || !x.decompress || !x.release || !x.crc32) x.lib = nullptr;
        return x;
    }();
    if (!f.lib || getenv("HGX_NO_LIBDEFLATE")) return nullptr;
    return &f;
}

// inflate every BGZF block of `data` into one buffer
// `on_part` (may be empty): called with [begin, end) of the output as soon as those bytes are complete -- the blocks are inflated
// in a few groups, and a group's bytes travel (the device front end's upload) while the next group is inflated
int bgzf_inflate(const Bytes &data, int n_threads, Bytes &out, const std::function<void(size_t, size_t)> &on_part = nullptr) {
    std::vector<Block> blocks;
    size_t off = 0, total = 0;
    const size_t n = data.size();
    while (off < n) {
        if (off + 18 > n || data[off] != 0x1f || data[off + 1] != 0x8b || data[off + 2] != 8 || !(data[off + 3] & 4)) {
            hgx_set_error("not a BGZF block at offset %zu", off);
            return HGX_EPARSE;
        }
        const unsigned xlen = rd16(&data[off + 10]);
        if (off + 12 + xlen > n) { hgx_set_error("truncated BGZF header at offset %zu", off); return HGX_EPARSE; }
        long bsize = -1;
        for (size_t p = off + 12; p + 4 <= off + 12 + xlen;) {
            const unsigned slen = rd16(&data[p + 2]);
            if (data[p] == 66 && data[p + 1] == 67 && slen == 2) bsize = rd16(&data[p + 4]);
            p += 4 + slen;
        }
        if (bsize < 0) { hgx_set_error("BGZF block without BC subfield at offset %zu", off); return HGX_EPARSE; }
        const size_t blen = (size_t)bsize + 1;
        if (off + blen > n || blen < 12 + xlen + 8) { hgx_set_error("truncated BGZF block at offset %zu", off); return HGX_EPARSE; }
        Block b;
        b.in_off = off + 12 + xlen;
        b.in_len = blen - 12 - xlen - 8;
        b.crc = rd32(&data[off + blen - 8]);
        b.out_len = rd32(&data[off + blen - 4]);
        b.out_off = total;
        total += b.out_len;
        blocks.push_back(b);
        off += blen;
    }
    out.alloc(total + 1);                    // (+1: room for the terminator of a last text line without '\n')
    out.n = total;
    std::vector<int> bad(std::max(1, n_threads), 0);
    const FastInflate *fi = fast_inflate();
    const size_t n_groups = (on_part && total > (64u << 20)) ? 4 : 1;
    for (size_t g = 0; g < n_groups; ++g) {
    const size_t g0 = blocks.size() * g / n_groups, g1 = blocks.size() * (g + 1) / n_groups;
    par_for(n_threads, g1 - g0, [&](int t, size_t b0, size_t b1) {
        void *dec = fi ? fi->alloc() : nullptr;
        for (size_t i = g0 + b0; i < g0 + b1; ++i) {
            const Block &b = blocks[i];
            if (b.out_len == 0) continue;
            if (dec) {
                size_t got = 0;
                if (fi->decompress(dec, &data[b.in_off], b.in_len, &out[b.out_off], b.out_len, &got) != 0 || got != b.out_len ||
                    fi->crc32(0, &out[b.out_off], b.out_len) != b.crc) { bad[t] = 1; break; }
                continue;
            }
            z_stream zs;
            memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, -15) != Z_OK) { bad[t] = 1; return; }
            zs.next_in = const_cast<unsigned char *>(&data[b.in_off]);
            zs.avail_in = (uInt)b.in_len;
            zs.next_out = &out[b.out_off];
            zs.avail_out = (uInt)b.out_len;
            const int rc = inflate(&zs, Z_FINISH);
            inflateEnd(&zs);
            if (rc != Z_STREAM_END || zs.total_out != b.out_len ||
                (uint32_t)crc32(crc32(0L, Z_NULL, 0), &out[b.out_off], (uInt)b.out_len) != b.crc) { bad[t] = 1; break; }
        }
        if (dec) fi->release(dec);
    });
    for (int v : bad) if (v) { hgx_set_error("corrupt BGZF block (inflate / CRC32 / ISIZE mismatch)"); return HGX_EPARSE; }
    if (on_part && g1 > g0) on_part(blocks[g0].out_off, g1 < blocks.size() ? blocks[g1].out_off : total);
    }
    return HGX_OK;
}

void put_int(PString &s, long long v) {
    char buf[24];
    int n = snprintf(buf, sizeof buf, "%lld", v);
    s.append(buf, (size_t)n);
}

// The real CIGAR of a record whose operation count does not fit 16 bits (SAM/BAM spec 4.2.2): the CIGAR field holds the
// placeholder <l_seq>S<ref span>N and the operations live in a CG:B:I tag.  As htslib does on reading (bam_tag2cigar: mapped
// record, first op = S over the whole read, CG of type B with 32-bit items, at least as many items as placeholder ops), the tag
// is taken for the CIGAR and dropped from the tag list.  Returns the tag's [begin, end) within the record and its items, or false.
bool find_real_cigar(const unsigned char *r, size_t len, size_t tags_at, int32_t ref_id, int32_t pos, uint32_t n_cig,
                     const unsigned char *cig, int32_t l_seq, size_t &tag_b, size_t &tag_e, const unsigned char *&items, uint32_t &n_items) {
    if (n_cig == 0 || ref_id < 0 || pos < 0) return false;
    const uint32_t c0 = rd32(cig);
    if ((c0 & 15) != 4 || (int64_t)(c0 >> 4) != (int64_t)l_seq) return false;
    size_t q = tags_at;
    while (q + 3 <= len) {
        const size_t b = q;
        const char t0 = (char)r[q], t1 = (char)r[q + 1], t = (char)r[q + 2];
        q += 3;
        size_t sz = 0;
        switch (t) {
            case 'A': case 'c': case 'C': sz = 1; break;
            case 's': case 'S': sz = 2; break;
            case 'i': case 'I': case 'f': sz = 4; break;
            case 'Z': case 'H': {
                const void *e = memchr(r + q, 0, len - q);
                if (!e) return false;
                sz = (size_t)((const unsigned char *)e - (r + q)) + 1;
            } break;
            case 'B': {
                if (q + 5 > len) return false;
                const char st = (char)r[q];
                const uint32_t cnt = rd32(r + q + 1);
                const size_t w = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : (st == 'i' || st == 'I' || st == 'f') ? 4 : 0;
                if (!w) return false;
                sz = 5 + w * (size_t)cnt;
                if (t0 == 'C' && t1 == 'G') {
                    if ((st != 'I' && st != 'i') || cnt < n_cig || cnt >= (1u << 29) || q + sz > len) return false;
                    tag_b = b; tag_e = q + sz; items = r + q + 5; n_items = cnt;
                    return true;
                }
            } break;
            default: return false;
        }
        if (t0 == 'C' && t1 == 'G') return false;              // a CG tag of another type is an ordinary tag
        if (q + sz > len) return false;
        q += sz;
    }
    return false;
}

// one BAM record (after its block_size word) -> SAM text line (no newline)
bool bam_record_text(const unsigned char *r, size_t len, const std::vector<std::string> &refs, PString &s) {
    static const char CIG[] = "MIDNSHP=X", SEQ[] = "=ACMGRSVTWYHKDBN";
    if (len < 32) return false;
    const int32_t ref_id = rdi32(r), pos = rdi32(r + 4);
    const unsigned l_rn = r[8], mapq = r[9];
    const unsigned n_cig = rd16(r + 12), flag = rd16(r + 14);
    const int32_t l_seq = rdi32(r + 16), nref = rdi32(r + 20), npos = rdi32(r + 24), tlen = rdi32(r + 28);
    size_t q = 32;
    if (l_rn == 0 || l_seq < 0 || q + l_rn + 4ull * n_cig + (size_t)(l_seq + 1) / 2 + (size_t)l_seq > len) return false;
    s.append((const char *)r + q, l_rn - 1);
    q += l_rn;
    s.push_back('\t'); put_int(s, flag);
    s.push_back('\t');
    if (ref_id >= 0 && (size_t)ref_id < refs.size()) s.append(refs[ref_id].data(), refs[ref_id].size()); else s.push_back('*');
    s.push_back('\t'); put_int(s, (long long)pos + 1);
    s.push_back('\t'); put_int(s, mapq);
    s.push_back('\t');
    size_t cg_b = 0, cg_e = 0;
    const unsigned char *cg_items = nullptr;
    uint32_t cg_n = 0;
    const size_t tags_at = q + 4ull * n_cig + (size_t)(l_seq + 1) / 2 + (size_t)l_seq;
    const bool long_cigar = find_real_cigar(r, len, tags_at, ref_id, pos, n_cig, r + q, l_seq, cg_b, cg_e, cg_items, cg_n);
    if (long_cigar) {
        for (uint32_t k = 0; k < cg_n; ++k) {
            const uint32_t v = rd32(cg_items + 4 * k);
            put_int(s, v >> 4);
            s.push_back((v & 15) < 9 ? CIG[v & 15] : '?');
        }
    } else {
        if (n_cig == 0) s.push_back('*');
        for (unsigned k = 0; k < n_cig; ++k) {
            const uint32_t v = rd32(r + q + 4 * k);
            put_int(s, v >> 4);
            s.push_back((v & 15) < 9 ? CIG[v & 15] : '?');
        }
    }
    q += 4ull * n_cig;
    s.push_back('\t');
    if (nref < 0) s.push_back('*');
    else if (nref == ref_id) s.push_back('=');
    else if ((size_t)nref < refs.size()) s.append(refs[nref].data(), refs[nref].size());
    else s.push_back('*');
    s.push_back('\t'); put_int(s, (long long)npos + 1);
    s.push_back('\t'); put_int(s, tlen);
    s.push_back('\t');
    if (l_seq == 0) s.push_back('*');
    else {
        const size_t at = s.size();
        s.resize(at + (size_t)l_seq);
        char *d = &s[at];
        for (int32_t i = 0; i + 1 < l_seq; i += 2) {
            const unsigned b = r[q + i / 2];
            d[i] = SEQ[b >> 4];
            d[i + 1] = SEQ[b & 15];
        }
        if (l_seq & 1) d[l_seq - 1] = SEQ[r[q + l_seq / 2] >> 4];
    }
    q += (size_t)(l_seq + 1) / 2;
    s.push_back('\t');
    if (l_seq == 0 || r[q] == 0xff) s.push_back('*');
    else {
        const size_t at = s.size();
        s.resize(at + (size_t)l_seq);
        char *d = &s[at];
        for (int32_t i = 0; i < l_seq; ++i) d[i] = (char)(r[q + i] + 33);
    }
    q += (size_t)l_seq;
    while (q + 3 <= len) {          // tags
        if (long_cigar && q == cg_b) { q = cg_e; continue; }      // the CG tag became the CIGAR
        s.push_back('\t');
        s.append((const char *)r + q, 2);
        const char t = (char)r[q + 2];
        q += 3;
        auto need = [&](size_t k) { return q + k <= len; };
        char buf[40];
        switch (t) {
            case 'A': if (!need(1)) return false; s += ":A:"; s.push_back((char)r[q]); q += 1; break;
            case 'c': if (!need(1)) return false; s += ":i:"; put_int(s, (int8_t)r[q]); q += 1; break;
            case 'C': if (!need(1)) return false; s += ":i:"; put_int(s, r[q]); q += 1; break;
            case 's': if (!need(2)) return false; s += ":i:"; put_int(s, (int16_t)rd16(r + q)); q += 2; break;
            case 'S': if (!need(2)) return false; s += ":i:"; put_int(s, rd16(r + q)); q += 2; break;
            case 'i': if (!need(4)) return false; s += ":i:"; put_int(s, rdi32(r + q)); q += 4; break;
            case 'I': if (!need(4)) return false; s += ":i:"; put_int(s, rd32(r + q)); q += 4; break;
            case 'f': {
                if (!need(4)) return false;
                float f; const uint32_t u = rd32(r + q); memcpy(&f, &u, 4);
                s += ":f:"; s.append(buf, (size_t)snprintf(buf, sizeof buf, "%g", f)); q += 4;
            } break;
            case 'Z': case 'H': {
                const void *e = memchr(r + q, 0, len - q);
                if (!e) return false;
                s.push_back(':'); s.push_back(t); s.push_back(':');
                s.append((const char *)r + q, (const unsigned char *)e - (r + q));
                q = (size_t)((const unsigned char *)e - r) + 1;
            } break;
            case 'B': {
                if (!need(5)) return false;
                const char st = (char)r[q];
                const uint32_t cnt = rd32(r + q + 1);
                q += 5;
                const size_t w = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : (st == 'i' || st == 'I' || st == 'f') ? 4 : 0;
                if (!w || !need(w * (size_t)cnt)) return false;
                s += ":B:"; s.push_back(st);
                for (uint32_t k = 0; k < cnt; ++k, q += w) {
                    s.push_back(',');
                    switch (st) {
                        case 'c': put_int(s, (int8_t)r[q]); break;
                        case 'C': put_int(s, r[q]); break;
                        case 's': put_int(s, (int16_t)rd16(r + q)); break;
                        case 'S': put_int(s, rd16(r + q)); break;
                        case 'i': put_int(s, rdi32(r + q)); break;
                        case 'I': put_int(s, rd32(r + q)); break;
                        default: {
                            float f; const uint32_t u = rd32(r + q); memcpy(&f, &u, 4);
                            s.append(buf, (size_t)snprintf(buf, sizeof buf, "%g", f));
                        }
                    }
                }
            } break;
            default: return false;
        }
    }
    return q == len;
}

typedef hgx_line Line;           // {p, len, klen = QNAME length, key = its first 8 bytes, big endian}
struct BRec { size_t first; uint32_t second; };      // a BAM record in the inflated stream: (offset after block_size, length)
typedef RawVec<BRec> BRecVec;                        // (no zero fill on resize: every element is written right after)

inline bool line_less(const Line &a, const Line &b) {
    if (a.key != b.key) return a.key < b.key;
    const uint32_t m = std::min(a.klen, b.klen);
    if (m > 8) { const int c = memcmp(a.p + 8, b.p + 8, m - 8); if (c) return c < 0; }
    return a.klen < b.klen;
}

void make_line(const char *p, size_t len, Line &l) {
    const char *tab = (const char *)memchr(p, '\t', len);
    const size_t k = tab ? (size_t)(tab - p) : len;
    l.p = const_cast<char *>(p); l.len = (uint32_t)len; l.klen = (uint32_t)k;
    uint64_t key = 0;
    for (size_t i = 0; i < 8; ++i) key = (key << 8) | (i < k ? (unsigned char)p[i] : 0);
    l.key = key;
}

// stable sort by QNAME: a sample sort.  Splitters from an evenly spaced sample, every element's bucket found by binary search
// (ranges of the input counted side by side, then scattered to their places in range order: the scatter is stable), buckets sorted
// side by side.  (Sorted chunks + rounds of pairwise merges ended in ONE thread merging the whole table: 9.7 ms for 1 M records.)
typedef RawVec<Line> LineVec;
void sort_lines(LineVec &v, int n_threads) {
    const size_t n = v.size();
    const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(n_threads, 1), n / 20000));
    if (T == 1) { std::stable_sort(v.begin(), v.end(), line_less); return; }
    const int B = T;                                                   // buckets
    std::vector<Line> sample;
    const size_t S = (size_t)B * 64;
    for (size_t k = 0; k < S; ++k) sample.push_back(v[n * k / S]);
    std::stable_sort(sample.begin(), sample.end(), line_less);
    std::vector<Line> split;                                           // B - 1 splitters; bucket b = elements in (split[b-1], split[b]]
    for (int b = 1; b < B; ++b) split.push_back(sample[S * b / B]);
    std::vector<uint16_t> bucket_of(n);
    std::vector<std::vector<size_t>> cnt(T, std::vector<size_t>(B, 0));
    par_for(T, n, [&](int t, size_t lo, size_t hi) {
        std::vector<size_t> &c = cnt[t];
        for (size_t i = lo; i < hi; ++i) {
            const int b = (int)(std::lower_bound(split.begin(), split.end(), v[i], line_less) - split.begin());    // equal names: one bucket
            bucket_of[i] = (uint16_t)b;
            c[b]++;
        }
    });
    std::vector<size_t> start(B + 1, 0);
    for (int b = 0; b < B; ++b) {
        size_t tot = 0;
        for (int t = 0; t < T; ++t) { const size_t c = cnt[t][b]; cnt[t][b] = start[b] + tot; tot += c; }
        start[b + 1] = start[b] + tot;
    }
    LineVec tmp(n);
    par_for(T, n, [&](int t, size_t lo, size_t hi) {
        std::vector<size_t> &at = cnt[t];
        for (size_t i = lo; i < hi; ++i) tmp[at[bucket_of[i]]++] = v[i];
    });
    std::atomic<size_t> next{0};
    par_for(T, (size_t)T, [&](int, size_t, size_t) {
        for (size_t b; (b = next.fetch_add(1)) < (size_t)B;)
            std::stable_sort(tmp.begin() + start[b], tmp.begin() + start[b + 1], line_less);
    });
    v.swap(tmp);
}

// ---- regions (samtools view syntax) ---------------------------------------------------------------------------------
// "name" = the whole reference; "name:l-r" / "name:l" / "name:-r" = 1-based inclusive span (commas allowed in numbers).  A string that
// names a reference as a whole wins over its "name:span" reading (HLA contigs contain ':'), as htslib resolves it.
struct Region {
    std::string whole;            // the full string (matches RNAME == whole: entire reference)
    std::string name;             // part before the last ':' when the rest parses as a span ("" = no such reading)
    int64_t left0 = 0, right0 = INT64_MAX;
};

bool parse_span(const char *p, const char *e, int64_t &l0, int64_t &r0) {
    auto num = [&](const char *&q, int64_t &v) {
        bool any = false;
        v = 0;
        while (q < e && ((*q >= '0' && *q <= '9') || *q == ',')) {
            if (*q != ',') { v = v * 10 + (*q - '0'); any = true; }
            ++q;
        }
        return any;
    };
    int64_t a = 0, b = 0;
    const bool ha = num(p, a);
    l0 = ha ? a - 1 : 0;
    r0 = INT64_MAX;
    if (p == e) return ha;
    if (*p != '-') return false;
    ++p;
    if (p == e) return ha;                      // "name:l-"
    if (!num(p, b) || p != e) return false;
    r0 = b - 1;
    return true;
}

std::vector<Region> parse_regions(const char *regions) {
    std::vector<Region> out;
    if (!regions) return out;
    const char *p = regions;
    while (*p) {
        const char *e = p;
        while (*e && *e != '\n') ++e;
        if (e > p) {
            Region r;
            r.whole.assign(p, e);
            const size_t colon = r.whole.rfind(':');
            if (colon != std::string::npos && colon > 0) {
                int64_t l0, r0;
                if (parse_span(r.whole.data() + colon + 1, r.whole.data() + r.whole.size(), l0, r0)) {
                    r.name = r.whole.substr(0, colon);
                    r.left0 = l0 < 0 ? 0 : l0;
                    r.right0 = r0;
                }
            }
            out.push_back(std::move(r));
        }
        p = *e ? e + 1 : e;
    }
    return out;
}

// does a record on reference `rname` spanning [pos0, end0] belong to region r?
inline bool region_hit(const Region &r, const char *rname, size_t rl, int64_t pos0, int64_t end0) {
    if (rl == r.whole.size() && memcmp(rname, r.whole.data(), rl) == 0) return true;
    if (!r.name.empty() && rl == r.name.size() && memcmp(rname, r.name.data(), rl) == 0) return end0 >= r.left0 && pos0 <= r.right0;
    return false;
}

// reference bases consumed by a SAM CIGAR string (M D N = X); 0 for "*"
inline int64_t cigar_text_reflen(const char *p, const char *e) {
    int64_t n = 0, tot = 0;
    for (; p < e; ++p) {
        const char c = *p;
        if (c >= '0' && c <= '9') { n = n * 10 + (c - '0'); continue; }
        if (c == 'M' || c == 'D' || c == 'N' || c == '=' || c == 'X') tot += n;
        n = 0;
    }
    return tot;
}

// BAM header: 1 = complete (refs, *body0 = offset of the first record), 0 = more bytes needed, -1 = not a BAM / malformed
int parse_bam_header(const unsigned char *raw, size_t n, std::vector<std::string> &refs, size_t *body0) {
    refs.clear();
    if (n < 4) return 0;
    if (memcmp(raw, "BAM\1", 4) != 0) return -1;
    if (n < 12) return 0;
    size_t p = 8 + (size_t)rd32(&raw[4]);
    if (p + 4 > n) return 0;
    const uint32_t n_ref = rd32(&raw[p]);
    p += 4;
    for (uint32_t i = 0; i < n_ref; ++i) {
        if (p + 4 > n) return 0;
        const uint32_t l_name = rd32(&raw[p]);
        if (l_name == 0) return -1;
        if (p + 4 + l_name + 4 > n) return 0;
        refs.emplace_back((const char *)&raw[p + 4], l_name - 1);
        p += 4 + l_name + 4;
    }
    *body0 = p;
    return 1;
}

// one BGZF block on this thread (zlib): false = corrupt
bool inflate_one(const unsigned char *data, const hgx_bgzf_block &b, unsigned char *dst) {
    if (b.out_len == 0) return true;
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = const_cast<unsigned char *>(data + b.in_off);
    zs.avail_in = (uInt)b.in_len;
    zs.next_out = dst;
    zs.avail_out = (uInt)b.out_len;
    const int rc = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    return rc == Z_STREAM_END && zs.total_out == b.out_len && (uint32_t)crc32(crc32(0L, Z_NULL, 0), dst, (uInt)b.out_len) == b.crc;
}

// the first bytes of a BGZF block's payload without inflating the block (a bgzipped SAM text, or anything else that is not a BAM,
// must not cost an upload and a device inflate whose output is thrown away): a few microseconds of zlib
bool peek_is_bam(const unsigned char *data, const std::vector<hgx_bgzf_block> &blocks) {
    for (const hgx_bgzf_block &b : blocks) {
        if (b.out_len == 0) continue;
        if (b.out_len < 4) return false;
        unsigned char head[4] = {0, 0, 0, 0};
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) return false;
        zs.next_in = const_cast<unsigned char *>(data + b.in_off);
        zs.avail_in = (uInt)b.in_len;
        zs.next_out = head;
        zs.avail_out = 4;
        const int rc = inflate(&zs, Z_SYNC_FLUSH);
        const bool ok = (rc == Z_OK || rc == Z_STREAM_END || rc == Z_BUF_ERROR) && zs.avail_out == 0 && memcmp(head, "BAM\1", 4) == 0;
        inflateEnd(&zs);
        return ok;
    }
    return false;
}

}   // namespace

// What a deferred stream's owner needs to pull ONE region list out of it (hgx_alignment_parse_dev: a file opened once, a locus at a
// time): the descriptor hgx_read_alignment_lines would have made had it been given these regions.  1 = not expressible (more than
// one region): the caller reads the file the ordinary way for this locus.
int hgx_deferred_for_regions(const char *regions, bool text, size_t body0, const std::vector<std::string> &refs, hgx_bam_deferred &d) {
    const std::vector<Region> regs = parse_regions(regions);
    const bool filtered = regions != nullptr && regions[0] != 0;
    if (regs.size() > 1 || (filtered && regs.size() != 1)) return 1;
    d = hgx_bam_deferred();
    d.on = true;
    d.text = text;
    d.body0 = body0;
    d.filtered = filtered;
    if (text) {
        if (filtered) { d.region_whole = regs[0].whole; d.region_name = regs[0].name; d.left0 = regs[0].left0; d.right0 = regs[0].right0; }
        return 0;
    }
    d.ref_action.assign(refs.size(), filtered ? 0 : 1);
    if (filtered) {
        const Region &r = regs[0];
        d.left0 = r.left0; d.right0 = r.right0;
        for (size_t i = 0; i < refs.size(); ++i) {
            if (refs[i].size() == r.whole.size() && memcmp(refs[i].data(), r.whole.data(), r.whole.size()) == 0) d.ref_action[i] = 1;
            else if (!r.name.empty() && refs[i].size() == r.name.size() && memcmp(refs[i].data(), r.name.data(), r.name.size()) == 0) d.ref_action[i] = 2;
        }
    }
    return 0;
}

// The reader proper: the records of `path` as a line table, stable-sorted by QNAME, over buffers `out` owns.  Every line is
// followed by one byte the parser may overwrite (its terminator).
int hgx_read_alignment_lines(const char *path, const char *regions, int n_threads, hgx_align_lines &out, bool keep_binary) {
    if (n_threads <= 0) n_threads = hgx_default_threads();
    n_threads = std::max(1, std::min(n_threads, 512));
    const std::vector<Region> regs = parse_regions(regions);
    const bool filtered = regions != nullptr && regions[0] != 0;      // an empty list after parsing keeps nothing, like an unknown name
    const size_t n_reg = std::max<size_t>(1, regs.size());
    const bool prof = getenv("HGX_PARSE_PROFILE") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = now();
    auto lap = [&](const char *what) {
        if (!prof) return;
        const double t = now();
        fprintf(stderr, "[hgx_read_alignments] %-18s %8.1f ms\n", what, (t - t_prev) * 1e3);
        t_prev = t;
    };
    // one line of SAM text -> the worker's lists (per region; list 0 when nothing is filtered).  Records = non-empty lines that do
    // not start with '@'.
    auto take_line = [&](const char *p, size_t len, std::vector<std::vector<Line>> &mine) {
        if (len && p[len - 1] == '\r') --len;
        if (!len || *p == '@') return;
        if (!filtered) { Line l; make_line(p, len, l); mine[0].push_back(l); return; }
        const char *f[6];               // FLAG = 2nd field, RNAME = 3rd, POS = 4th (1-based), CIGAR = 6th
        const char *q = p, *le = p + len;
        int nf = 0;
        while (nf < 6 && (q = (const char *)memchr(q, '\t', (size_t)(le - q))) != nullptr) f[nf++] = ++q;
        if (nf != 6) return;
        const long flag = strtol(f[0], nullptr, 10);
        const int64_t pos0 = strtol(f[2], nullptr, 10) - 1;
        const int64_t reflen = (flag & 4) ? 0 : cigar_text_reflen(f[4], f[5] - 1);
        const int64_t end0 = pos0 + (reflen > 0 ? reflen : 1) - 1;
        const size_t rl = (size_t)(f[2] - 1 - f[1]);
        for (size_t g = 0; g < regs.size(); ++g)
            if (region_hit(regs[g], f[1], rl, pos0, end0)) { Line l; make_line(p, len, l); mine[g].push_back(l); }
    };
    std::vector<std::vector<std::vector<Line>>> text_part;      // [worker][region] line lists of a SAM text
    bool on_raw_done = false;
    int text_nt = 0;
    bool text_scanned = false, text_defer = false;
    Bytes data;
    {
        const int fd = open(path, O_RDONLY);
        if (fd < 0) { hgx_set_error("cannot open %s", path); return HGX_EINVAL; }
        struct stat sb;
        if (fstat(fd, &sb) != 0 || sb.st_size < 0) { close(fd); hgx_set_error("cannot stat %s", path); return HGX_EINVAL; }
        data.alloc((size_t)sb.st_size + 1);                // not zero-filled: every byte is written by the reads below
        data.n = (size_t)sb.st_size;                       // (+1: room for the terminator of a last line without '\n')
        std::vector<int> bad(n_threads, 0);
        unsigned char magic[4] = {0, 0, 0, 0};
        if (sb.st_size >= 4 && pread(fd, magic, 4, 0) != 4) { close(fd); hgx_set_error("short read on %s", path); return HGX_EINVAL; }
        const bool is_text = !(magic[0] == 0x1f && magic[1] == 0x8b) && memcmp(magic, "BAM\1", 4) != 0;
        // SAM text whose line table the caller makes itself (the device front end): any size from its gate on, not only big files
        const bool text_defer_ok = out.defer_text && out.on_raw && keep_binary && regs.size() <= 1 && (!filtered || regs.size() == 1) &&
                                   data.size() >= out.defer_min_bytes && data.size() < (1ull << 32) - 64;
        if (is_text && (data.size() > (8u << 20) || text_defer_ok)) {
            const int n_threads_file = n_threads;
            const int n_threads = data.size() > (8u << 20) ? n_threads_file : 1;       // (a small file: one reader, no thread start-up)
            // Big SAM text: every worker reads its byte range in 1 MB pieces and scans each piece for lines while it is still in
            // its cache (the separate scan of the whole text was a second trip through 400 MB of DRAM).  A worker owns the lines
            // that START in its range; the one that runs over the end of the range is finished after all ranges are in.
            // With an upload waiting (out.on_raw: the device front end) the file is read in a few phases, and each phase's bytes are
            // handed over while the next phase is read: the transfer hides behind the read.
            const char *ph_env = getenv("HGX_READ_PHASES");
            // (eight: the device front end takes the first six apart while the seventh lands, the seventh beside the eighth -- what waits
            // for the file's last byte is an eighth of the text, csrc/hgx_front.hip records_split)
            const int n_phase = (out.on_raw && data.size() > (64u << 20)) ? (ph_env ? std::max(1, atoi(ph_env)) : 8) : 1;
            // the caller makes the line table itself (the device front end: newline scan, region filter, name order as kernels):
            // the workers only read, the bytes go up phase by phase
            text_defer = text_defer_ok;
            text_nt = n_threads * n_phase;
            if (!text_defer) text_part.assign(text_nt, std::vector<std::vector<Line>>(n_reg));
            std::vector<size_t> tail((size_t)text_nt, (size_t)-1);
            char *base = (char *)data.data();
            const size_t total = data.size();
            for (int ph = 0; ph < n_phase; ++ph) {
            const size_t ph_b = total * (size_t)ph / n_phase, ph_e = total * (size_t)(ph + 1) / n_phase;
            par_for(n_threads, ph_e - ph_b, [&](int t, size_t rb0, size_t re0) {
                const size_t b0 = ph_b + rb0, e0 = ph_b + re0;
                if (text_defer) {
                    for (size_t have = b0; have < e0;) {
                        const ssize_t got = pread(fd, base + have, std::min<size_t>(e0 - have, 4u << 20), (off_t)have);
                        if (got <= 0) { bad[t] = 1; return; }
                        have += (size_t)got;
                    }
                    return;
                }
                std::vector<std::vector<Line>> &mine = text_part[ph * n_threads + t];
                if (!filtered) mine[0].reserve((e0 - b0) / 300 + 16);
                unsigned char prev = '\n';
                if (b0 > 0 && pread(fd, &prev, 1, (off_t)(b0 - 1)) != 1) { bad[t] = 1; return; }
                bool skipping = prev != '\n';                // the head of my range is the tail of a line owned by the previous range
                size_t have = b0;
                const char *p = base + b0;
                while (have < e0) {
                    const size_t want = std::min<size_t>(e0 - have, 1u << 20);
                    size_t got_all = 0;
                    while (got_all < want) {
                        const ssize_t got = pread(fd, base + have + got_all, want - got_all, (off_t)(have + got_all));
                        if (got <= 0) { bad[t] = 1; return; }
                        got_all += (size_t)got;
                    }
                    have += want;
                    const char *lim = base + have;
                    if (skipping) {
                        const char *q = (const char *)memchr(p, '\n', (size_t)(lim - p));
                        if (!q) { p = lim; continue; }
                        p = q + 1;
                        skipping = false;
                    }
                    while (p < base + e0) {
                        const char *e = (const char *)memchr(p, '\n', (size_t)(lim - p));
                        if (!e) break;
                        take_line(p, (size_t)(e - p), mine);
                        p = e + 1;
                    }
                }
                if (!skipping && p < base + e0) tail[ph * n_threads + t] = (size_t)(p - base);
            });
            bool any_bad = false;
            for (int v : bad) any_bad = any_bad || v;
            if (any_bad) break;
            if (out.on_raw) { out.on_raw(base, total, ph_b, ph_e); on_raw_done = true; }
            }
            close(fd);
            for (int v : bad) if (v) { hgx_set_error("short read on %s", path); return HGX_EINVAL; }
            if (text_defer) {
                hgx_bam_deferred &d = out.deferred;
                d.on = true;
                d.text = true;
                d.filtered = filtered;
                if (filtered) { d.region_whole = regs[0].whole; d.region_name = regs[0].name; d.left0 = regs[0].left0; d.right0 = regs[0].right0; }
                out.binary = false;
                out.lines.clear();
                hgx_host_free(out.raw);
                out.raw = (char *)data.p;
                out.raw_bytes = data.n;
                data.p = nullptr;
                data.n = 0;
                lap("read file (lines left to the device)");
                return HGX_OK;
            }
            for (int t = 0; t < text_nt; ++t)
                if (tail[t] != (size_t)-1) {
                    const char *p = base + tail[t], *end = base + total;
                    const char *e = (const char *)memchr(p, '\n', (size_t)(end - p));
                    take_line(p, (size_t)((e ? e : end) - p), text_part[t]);
                }
            text_scanned = true;
        } else {
            par_for(data.size() > (8u << 20) ? n_threads : 1, data.size(), [&](int t, size_t b, size_t e) {
                while (b < e) {
                    const ssize_t got = pread(fd, data.data() + b, e - b, (off_t)b);
                    if (got <= 0) { bad[t] = 1; return; }
                    b += (size_t)got;
                }
            });
            close(fd);
            for (int v : bad) if (v) { hgx_set_error("short read on %s", path); return HGX_EINVAL; }
        }
    }
    lap(text_scanned ? "read file + lines" : "read file");
    Bytes raw;
    if (data.size() >= 2 && data[0] == 0x1f && data[1] == 0x8b && keep_binary && out.defer_walk && out.inflate_dev && regs.size() <= 1 &&
        (!filtered || regs.size() == 1)) {
        // The caller inflates, walks, filters and sorts on the device: the host hops through the container, inflates the block(s) that
        // hold the BAM header to learn the references and where the records begin, and hands the deflated bytes over.  Whatever
        // does not fit (not a BAM, a malformed container or header, a small stream, a block the kernel does not take) goes the
        // ordinary way below, which also words the errors.
        std::vector<hgx_bgzf_block> blocks;
        size_t total = 0;
        // (a bgzipped SAM text or any other BGZF file that is not a BAM must not cost an upload and a device inflate whose
        // output is thrown away -- the first block's first four bytes say which it is, before anything is sent)
        bool looks_bam = false;
        {
            std::vector<hgx_bgzf_block> first;
            const size_t n = data.size();
            if (n >= 18 && (data[3] & 4)) {
                const size_t xlen = (size_t)data[10] | ((size_t)data[11] << 8);
                size_t blen = 0;
                for (size_t q = 12; q + 4 <= 12 + xlen && q + 6 <= n;) {
                    const size_t slen = (size_t)data[q + 2] | ((size_t)data[q + 3] << 8);
                    if (data[q] == 66 && data[q + 1] == 67 && slen == 2) blen = ((size_t)data[q + 4] | ((size_t)data[q + 5] << 8)) + 1;
                    q += 4 + slen;
                }
                if (blen && blen <= n && hgx_bgzf_scan(data.data(), blen, first, nullptr) == HGX_OK && first.size() == 1)
                    looks_bam = first[0].out_len == 0 || peek_is_bam(data.data(), first);      // (an empty first block: the full scan decides)
            }
        }
        const bool dbg_t = getenv("HGX_BAM_DEBUG") != nullptr;
        const auto t_dbg0 = std::chrono::steady_clock::now();
        auto dbg_mark = [&](const char *what) {
            if (dbg_t) fprintf(stderr, "[bam deferred] %-28s +%.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_dbg0).count());
        };
        if (looks_bam && out.comp_early && data.size() >= out.defer_min_bytes / 64) out.comp_early(data.data(), data.size());   // (BGZF rarely deflates below 1 : 64)
        dbg_mark("upload queued");
        if (looks_bam && hgx_bgzf_scan_par(data.data(), data.size(), blocks, &total, std::min(n_threads, 8)) == HGX_OK && total < (1ull << 32) - 64 &&
            peek_is_bam(data.data(), blocks)) {
            dbg_mark("container scanned");
            std::vector<unsigned char> head;
            std::vector<std::string> refs;
            size_t body0 = 0;
            int st_h = 0, dev_rc = 1;
            // the caller's inflate (the kernel's launch and its wait) on this thread, the BAM header -- the first block or two through
            // zlib: 0.2-0.3 ms -- on a worker beside it (round 5: it stood in front of the launch)
            hgx_run_workers(2, [&](int w) {
                if (w == 0) {
                    dbg_mark("worker 0 running");
                    if (total >= out.defer_min_bytes) dev_rc = out.inflate_dev(data.data(), data.size(), blocks, total);
                    dbg_mark("inflate_dev back");
                } else {
                    for (size_t k = 0; k < blocks.size() && st_h == 0 && head.size() < (64u << 20); ++k) {
                        const size_t at = head.size();
                        head.resize(at + blocks[k].out_len);
                        if (!inflate_one(data.data(), blocks[k], head.data() + at)) { st_h = -1; break; }
                        st_h = parse_bam_header(head.data(), head.size(), refs, &body0);
                    }
                }
            });
            if (st_h == 1 && total >= body0 && total - body0 >= out.defer_min_bytes && refs.size() < 65536 && dev_rc == 0) {
                hgx_bam_deferred &d = out.deferred;
                d.on = true;
                d.on_device = true;
                d.body0 = body0;
                d.filtered = filtered;
                d.ref_action.assign(refs.size(), filtered ? 0 : 1);
                if (filtered) {
                    const Region &r = regs[0];
                    d.left0 = r.left0; d.right0 = r.right0;
                    for (size_t i = 0; i < refs.size(); ++i) {
                        if (refs[i].size() == r.whole.size() && memcmp(refs[i].data(), r.whole.data(), r.whole.size()) == 0) d.ref_action[i] = 1;
                        else if (!r.name.empty() && refs[i].size() == r.name.size() && memcmp(refs[i].data(), r.name.data(), r.name.size()) == 0) d.ref_action[i] = 2;
                    }
                }
                out.binary = true;
                out.ref_names = refs;
                out.lines.clear();
                hgx_host_free(out.raw);
                out.raw = nullptr;
                out.raw_bytes = total;
                lap("container + header (inflate, records: the device)");
                return HGX_OK;
            }
        }
    }
    if (data.size() >= 2 && data[0] == 0x1f && data[1] == 0x8b) {
        std::function<void(size_t, size_t)> part;
        if (out.on_raw) {
            part = [&](size_t b, size_t e) { out.on_raw((const char *)raw.data(), raw.size(), b, e); };
            on_raw_done = true;
        }
        const int rc = bgzf_inflate(data, n_threads, raw, part);
        if (out.comp_sync) out.comp_sync();                    // (an upload of these bytes begun by comp_early may still be reading them: on
        if (rc) return rc;                                     //  EVERY way out of here that gives `data` back)
        data.release();
    } else raw.swap(data);
    lap("inflate");
    if (out.on_raw && !on_raw_done) out.on_raw((const char *)raw.data(), raw.size(), 0, raw.size());       // (the device front end's upload starts here)
    LineVec &lines = out.lines;
    lines.clear();
    if (raw.size() >= 4 && memcmp(raw.data(), "BAM\1", 4) == 0) {
        const size_t n = raw.size();
        if (n < 12) { hgx_set_error("truncated BAM header"); return HGX_EPARSE; }
        size_t p = 8 + (size_t)rd32(&raw[4]);
        if (p + 4 > n) { hgx_set_error("truncated BAM header"); return HGX_EPARSE; }
        const uint32_t n_ref = rd32(&raw[p]);
        p += 4;
        std::vector<std::string> refs;
        for (uint32_t i = 0; i < n_ref; ++i) {
            if (p + 4 > n) { hgx_set_error("truncated BAM reference list"); return HGX_EPARSE; }
            const uint32_t l_name = rd32(&raw[p]);
            if (l_name == 0 || p + 4 + l_name + 4 > n) { hgx_set_error("truncated BAM reference list"); return HGX_EPARSE; }
            refs.emplace_back((const char *)&raw[p + 4], l_name - 1);
            p += 4 + l_name + 4;
        }
        if (keep_binary && out.defer_walk && regs.size() <= 1 && (!filtered || regs.size() == 1) && n - p >= out.defer_min_bytes &&
            n < (1ull << 32) - 64 && refs.size() < 65536) {
            // the caller walks, filters and sorts the records itself (the device front end): hand the stream over as it is
            hgx_bam_deferred &d = out.deferred;
            d.on = true;
            d.body0 = p;
            d.filtered = filtered;
            d.ref_action.assign(refs.size(), filtered ? 0 : 1);
            if (filtered) {
                const Region &r = regs[0];
                d.left0 = r.left0; d.right0 = r.right0;
                for (size_t i = 0; i < refs.size(); ++i) {
                    if (refs[i].size() == r.whole.size() && memcmp(refs[i].data(), r.whole.data(), r.whole.size()) == 0) d.ref_action[i] = 1;
                    else if (!r.name.empty() && refs[i].size() == r.name.size() && memcmp(refs[i].data(), r.name.data(), r.name.size()) == 0) d.ref_action[i] = 2;
                }
            }
            out.binary = true;
            out.ref_names = refs;
            lines.clear();
            hgx_host_free(out.raw);
            out.raw = (char *)raw.p;
            out.raw_bytes = raw.n;
            raw.p = nullptr;
            raw.n = 0;
            lap("  (records left to the device)");
            return HGX_OK;
        }
        // ---- record chain ------------------------------------------------------------------------------------------
        // The records form a chain (each block_size leads to the next) that one thread walks at ~15 ns per record.  For big
        // files the inflated stream is cut into ranges instead: every worker but the first GUESSES a record start in its
        // range (a header that is plausible and leads to three more plausible headers) and walks from there past the end of
        // its range.  The guesses are then CHECKED: worker t's walk must end exactly where worker t+1's begins; the first
        // mismatch falls back to the plain walk from that point on.  The record list is therefore exact, not heuristic.
        auto plausible = [&](size_t o) -> bool {
            if (o + 36 > n) return false;
            const uint32_t bs = rd32(&raw[o]);
            if (bs < 32 || o + 4 + (size_t)bs > n) return false;
            const unsigned char *r = &raw[o + 4];
            const int32_t rid = rdi32(r), pos = rdi32(r + 4), nrid = rdi32(r + 20), npos = rdi32(r + 24), l_seq = rdi32(r + 16);
            const uint32_t l_rn = r[8], n_cig = rd16(r + 12);
            if (rid < -1 || rid >= (int32_t)refs.size() || nrid < -1 || nrid >= (int32_t)refs.size()) return false;
            if (pos < -1 || npos < -1 || l_seq < 0 || l_rn == 0) return false;
            if (32 + (size_t)l_rn + 4 * (size_t)n_cig + (size_t)(l_seq + 1) / 2 + (size_t)l_seq > bs) return false;
            if (r[32 + l_rn - 1] != 0) return false;
            for (uint32_t k = 0; k + 1 < l_rn; ++k) if (r[32 + k] < 33 || r[32 + k] > 126) return false;
            return true;
        };
        BRecVec chain;          // every record: (offset after block_size, length)
        const size_t body0 = p;
        const char *chain_min = hgx_test_switch("bam_chain_min");              // bytes of records from which the chain is walked in ranges (tests)
        const size_t chain_min_bytes = chain_min ? (size_t)strtoull(chain_min, nullptr, 10) : (32u << 20);
        const int W = (n - body0 > chain_min_bytes) ? std::max(1, std::min(n_threads, 64)) : 1;
        bool chain_error = false;
        size_t err_at = 0;
        auto walk = [&](size_t from, size_t until, BRecVec &dst, size_t &stop) -> bool {
            size_t q = from;
            while (q < until && q < n) {
                if (q + 8192 < n) {       // a pointer chase through memory other cores just wrote: touch the lines ahead
                    __builtin_prefetch(&raw[q + 4096]);
                    __builtin_prefetch(&raw[q + 4096 + 64]);
                    __builtin_prefetch(&raw[q + 4096 + 128]);
                    __builtin_prefetch(&raw[q + 4096 + 192]);
                }
                if (q + 4 > n) { stop = q; return false; }
                const uint32_t bs = rd32(&raw[q]);
                if (bs < 32 || q + 4 + (size_t)bs > n) { stop = q; return false; }
                dst.push_back({q + 4, bs});
                q += 4 + (size_t)bs;
            }
            stop = q;
            return true;
        };
        if (W == 1) {
            size_t stop = 0;
            if (!walk(body0, n, chain, stop)) { chain_error = true; err_at = stop; }
        } else {
            std::vector<BRecVec> part(W);
            std::vector<size_t> first(W, 0), stop(W, 0);
            std::vector<int> okw(W, 0);
            par_for(W, (size_t)W, [&](int, size_t tb, size_t te) {
                for (size_t t = tb; t < te; ++t) {
                    const size_t lo = body0 + (n - body0) * t / W, hi = body0 + (n - body0) * (t + 1) / W;
                    size_t o = lo;
                    if (t > 0) {
                        bool found = false;
                        for (; o < hi; ++o) {
                            if (!plausible(o)) continue;
                            size_t q = o;
                            int good = 0;
                            while (good < 4 && q < n && plausible(q)) { q += 4 + (size_t)rd32(&raw[q]); ++good; }
                            if (good == 4 || q == n) { found = true; break; }
                        }
                        if (!found) { okw[t] = 0; first[t] = hi; stop[t] = hi; continue; }
                    }
                    first[t] = o;
                    part[t].reserve((hi - lo) / 200 + 16);
                    okw[t] = walk(o, hi, part[t], stop[t]) ? 1 : 0;
                }
            });
            // stitch: accept worker t's records iff the chain so far ends exactly at its first record.  The usual case -- every
            // guess was right -- is recognised first and copied side by side (a serial insert of 16 MB was 3-4 ms)
            bool all_good = first[0] == body0 && okw[0];
            for (int t = 1; t < W && all_good; ++t) all_good = okw[t] && first[t] == stop[t - 1];
            all_good = all_good && stop[W - 1] == n;
            if (all_good) {
                std::vector<size_t> off(W + 1, 0);
                for (int t = 0; t < W; ++t) off[t + 1] = off[t] + part[t].size();
                chain.resize(off[W]);
                par_for(W, (size_t)W, [&](int, size_t tb, size_t te) {
                    for (size_t t = tb; t < te; ++t)
                        if (!part[t].empty()) memcpy(&chain[off[t]], part[t].data(), part[t].size() * sizeof(chain[0]));
                });
            }
            size_t at = all_good ? n : body0;
            for (int t = 0; t < W && !chain_error && !all_good; ++t) {
                if (at == first[t] && okw[t]) {
                    chain.insert(chain.end(), part[t].begin(), part[t].end());
                    at = stop[t];
                } else if (at < (t + 1 < W ? first[t + 1] : n) || !okw[t]) {
                    // the guess was wrong (or the range held no record start): walk on plainly up to the next range's first record
                    const size_t until = t + 1 < W ? body0 + (n - body0) * (t + 1) / W : n;
                    size_t st2 = 0;
                    if (!walk(at, until, chain, st2)) { chain_error = true; err_at = st2; }
                    at = st2;
                }
            }
            if (!chain_error && at != n) {
                size_t st2 = 0;
                if (!walk(at, n, chain, st2)) { chain_error = true; err_at = st2; }
            }
        }
        if (chain_error) { hgx_set_error("truncated BAM record at offset %zu", err_at); return HGX_EPARSE; }
        // ---- region filter (parallel over the chain) -----------------------------------------------------------------
        BRecVec recs;
        if (!filtered) recs.swap(chain);
        else {
            const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_threads, chain.size() / 4096 + 1));
            std::vector<std::vector<BRecVec>> hit(T, std::vector<BRecVec>(n_reg));
            par_for(T, chain.size(), [&](int t, size_t b, size_t e) {
                for (size_t i = b; i < e; ++i) {
                    const unsigned char *r = &raw[chain[i].first];
                    const uint32_t bs = chain[i].second;
                    const int32_t rid = rdi32(r), pos = rdi32(r + 4);
                    if (rid < 0 || (size_t)rid >= refs.size()) continue;
                    // reference span from the CIGAR (bam_endpos: an unmapped or zero-length record counts as one base)
                    const uint32_t l_rn = r[8], n_cig = rd16(r + 12), flag = rd16(r + 14);
                    int64_t reflen = 0;
                    if (!(flag & 4) && 32 + (size_t)l_rn + 4 * (size_t)n_cig <= bs) {
                        const unsigned char *c = r + 32 + l_rn;
                        for (uint32_t k = 0; k < n_cig; ++k) {
                            const uint32_t v = rd32(c + 4 * k), op = v & 15;
                            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) reflen += v >> 4;
                        }
                    }
                    const int64_t end0 = (int64_t)pos + (reflen > 0 ? reflen : 1) - 1;
                    const std::string &nm = refs[rid];
                    for (size_t g = 0; g < regs.size(); ++g)
                        if (region_hit(regs[g], nm.data(), nm.size(), pos, end0)) hit[t][g].push_back(chain[i]);
                }
            });
            std::vector<size_t> off((size_t)n_reg * T + 1, 0);   // region after region, file order inside: offsets, then copies side by side
            for (size_t g = 0; g < n_reg; ++g)
                for (int t = 0; t < T; ++t) off[g * T + t + 1] = off[g * T + t] + hit[t][g].size();
            recs.resize(off[(size_t)n_reg * T]);
            par_for(T, (size_t)T, [&](int, size_t tb, size_t te) {
                for (size_t t = tb; t < te; ++t)
                    for (size_t g = 0; g < n_reg; ++g)
                        if (!hit[t][g].empty()) memcpy(&recs[off[g * T + t]], hit[t][g].data(), hit[t][g].size() * sizeof(recs[0]));
            });
        }
        lap("  BAM record walk");
        if (keep_binary) {
            // the parser reads the records as they are (hgx_sam.cpp: split_bam): only the sort keys are made here
            out.binary = true;
            out.ref_names = refs;
            lines.resize(recs.size());
            std::vector<int> badrec(std::max(1, n_threads), 0);
            par_for(n_threads, recs.size(), [&](int t, size_t b, size_t e) {
                for (size_t i = b; i < e; ++i) {
                    unsigned char *r = &raw[recs[i].first];
                    const uint32_t bs = recs[i].second, l_rn = r[8];
                    if (l_rn == 0 || 32 + (size_t)l_rn > bs || r[32 + l_rn - 1] != 0) { badrec[t] = 1; return; }
                    Line &l = lines[i];
                    l.p = (char *)r + 32;
                    l.len = bs;
                    l.klen = l_rn - 1;
                    uint64_t key = 0;
                    for (uint32_t k = 0; k < 8; ++k) key = (key << 8) | (k < l.klen ? r[32 + k] : 0);
                    l.key = key;
                }
            });
            for (int v : badrec) if (v) { hgx_set_error("malformed BAM record"); return HGX_EPARSE; }
            lap("  BAM sort keys");
        } else {
        const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_threads, recs.size() / 2000 + 1));
        out.chunks.clear();
        out.chunks.resize(T);
        std::vector<int> bad(T, 0);
        lines.resize(recs.size());
        par_for(T, recs.size(), [&](int t, size_t b, size_t e) {
            PString &s = out.chunks[t];
            s.reserve((e - b) * 420);
            std::vector<uint32_t> ends;
            ends.reserve(e - b);
            for (size_t i = b; i < e; ++i) {
                if (!bam_record_text(&raw[recs[i].first], recs[i].second, refs, s)) { bad[t] = 1; return; }
                if (s.size() > 0xFFFFFFF0ull) { bad[t] = 2; return; }
                ends.push_back((uint32_t)s.size());
                s.push_back('\n');                              // the byte after every line belongs to the parser
            }
            uint32_t prev = 0;                            // the chunk's text no longer moves: its line table
            for (size_t i = b; i < e; ++i) {
                make_line(s.data() + prev, ends[i - b] - prev, lines[i]);
                prev = ends[i - b] + 1;
            }
        });
        for (int v : bad) if (v) { hgx_set_error(v == 2 ? "BAM chunk too large" : "malformed BAM record"); return HGX_EPARSE; }
        raw.release();                                    // the records now live as text in out.chunks
        lap("  BAM -> text");
        }
    } else {
        // SAM text: byte ranges are split among the workers; a worker owns the lines that START in its range; per worker and
        // region a list, concatenated region-major (file order inside).  (A big file was scanned while it was read, above.)
        const char *base = (const char *)raw.data(), *end = base + raw.size();
        const int nt = text_scanned ? text_nt : (raw.size() > (4u << 20) ? n_threads : 1);
        std::vector<std::vector<std::vector<Line>>> &part = text_part;
        if (!text_scanned) {
            part.assign(nt, std::vector<std::vector<Line>>(n_reg));
            par_for(nt, raw.size(), [&](int t, size_t b0, size_t e0) {
                const char *p = base + b0;
                if (b0 > 0) {                       // skip the tail of a line owned by the previous range
                    const char *q = (const char *)memchr(p - 1, '\n', (size_t)(end - (p - 1)));
                    p = q ? q + 1 : end;
                }
                std::vector<std::vector<Line>> &mine = part[t];
                if (!filtered) mine[0].reserve((e0 - b0) / 300 + 16);
                while (p < base + e0 && p < end) {
                    const char *e = (const char *)memchr(p, '\n', (size_t)(end - p));
                    if (!e) e = end;
                    take_line(p, (size_t)(e - p), mine);
                    p = e + 1;
                }
            });
        }
        // offsets of every (region, worker) list in the final table, then a parallel copy
        std::vector<size_t> off((size_t)nt * n_reg + 1, 0);
        size_t tot = 0;
        for (size_t g = 0; g < n_reg; ++g)
            for (int t = 0; t < nt; ++t) { off[g * nt + t] = tot; tot += part[t][g].size(); }
        lines.resize(tot);
        par_for(std::min(nt, n_threads), (size_t)nt, [&](int, size_t b, size_t e) {
            for (size_t t = b; t < e; ++t)
                for (size_t g = 0; g < n_reg; ++g)
                    if (!part[t][g].empty()) memcpy(&lines[off[g * nt + t]], part[t][g].data(), part[t][g].size() * sizeof(Line));
        });
    }
    lap("decode / split");
    // an aligner writes its records grouped by read already: a stable sort would not move anything
    bool sorted = true;
    {
        std::vector<int> unsorted(n_threads, 0);
        par_for(lines.size() > 100000 ? n_threads : 1, lines.size(), [&](int t, size_t b, size_t e) {
            for (size_t i = std::max<size_t>(b, 1); i < e; ++i)
                if (line_less(lines[i], lines[i - 1])) { unsorted[t] = 1; return; }
        });
        for (int v : unsorted) if (v) sorted = false;
    }
    if (!sorted) sort_lines(lines, n_threads);
    lap(sorted ? "name order check" : "name sort");
    hgx_host_free(out.raw);
    out.raw = (char *)raw.p;                 // the line table points into it (SAM text); BAM text lives in out.chunks
    out.raw_bytes = raw.n;
    raw.p = nullptr;
    raw.n = 0;
    return HGX_OK;
}

// many tasks' BAM files, read side by side and left deflated (hgx_internal.hpp: hgx_bgzf_task)
int hgx_bgzf_tasks_read(std::vector<hgx_bgzf_task> &tasks, const char *const *paths, const char *const *regions, int n_tasks, int n_threads,
                        const hgx_front_alloc *mem, const std::function<void(int)> &on_task) {
    HARGCHK(paths && n_tasks >= 0);
    tasks.clear();
    tasks.resize((size_t)n_tasks);
    if (n_threads <= 0) n_threads = hgx_default_threads();
    n_threads = std::max(1, std::min(n_threads, 512));
    hgx_par_tasks(std::min(n_threads, std::max(n_tasks, 1)), (size_t)n_tasks, [&](int, size_t t) {
        hgx_bgzf_task &T = tasks[t];
        try {
            if (!paths[t]) return;
            const int fd = open(paths[t], O_RDONLY);
            if (fd < 0) return;
            struct stat sb;
            if (fstat(fd, &sb) != 0 || sb.st_size < 28) { close(fd); return; }
            {
                std::unique_ptr<hgx_big_alloc_scope> pinned;
                if (mem && mem->alloc) pinned.reset(new hgx_big_alloc_scope(*mem, 64u << 10));
                T.data = (unsigned char *)hgx_host_alloc((size_t)sb.st_size + 1);
            }
            T.n = (size_t)sb.st_size;
            size_t got_all = 0;
            while (got_all < T.n) {
                const ssize_t got = pread(fd, T.data + got_all, T.n - got_all, (off_t)got_all);
                if (got <= 0) break;
                got_all += (size_t)got;
            }
            close(fd);
            if (got_all != T.n || T.data[0] != 0x1f || T.data[1] != 0x8b) return;
            const std::vector<Region> regs = parse_regions(regions ? regions[t] : nullptr);
            const bool filtered = regions && regions[t] && regions[t][0] != 0;
            if (regs.size() > 1 || (filtered && regs.size() != 1)) return;
            if (hgx_bgzf_scan(T.data, T.n, T.blocks, &T.total) != HGX_OK || T.total >= (1ull << 32) - 64) return;
            std::vector<unsigned char> head;
            std::vector<std::string> refs;
            size_t body0 = 0;
            int st_h = 0;
            for (size_t k = 0; k < T.blocks.size() && st_h == 0 && head.size() < (64u << 20); ++k) {
                const size_t at = head.size();
                head.resize(at + T.blocks[k].out_len);
                if (!inflate_one(T.data, T.blocks[k], head.data() + at)) { st_h = -1; break; }
                st_h = parse_bam_header(head.data(), head.size(), refs, &body0);
            }
            if (st_h != 1 || body0 > T.total || refs.size() >= 65536) return;
            hgx_bam_deferred &d = T.def;
            d.on = true; d.on_device = true; d.body0 = body0; d.filtered = filtered;
            d.ref_action.assign(refs.size(), filtered ? 0 : 1);
            if (filtered) {
                const Region &r = regs[0];
                d.left0 = r.left0; d.right0 = r.right0;
                for (size_t i = 0; i < refs.size(); ++i) {
                    if (refs[i].size() == r.whole.size() && memcmp(refs[i].data(), r.whole.data(), r.whole.size()) == 0) d.ref_action[i] = 1;
                    else if (!r.name.empty() && refs[i].size() == r.name.size() && memcmp(refs[i].data(), r.name.data(), r.name.size()) == 0) d.ref_action[i] = 2;
                }
            }
            T.ok = true;
            if (on_task) on_task((int)t);
        } catch (const std::exception &) {
            T.ok = false;
        }
    });
    return HGX_OK;
}

extern "C" int hgx_read_alignments(const char *path, const char *regions, int32_t n_threads, char **text_out, size_t *n_bytes_out) {
    HARGCHK(path && text_out && n_bytes_out);
    *text_out = nullptr;
    *n_bytes_out = 0;
    try {
        hgx_align_lines al;
        const int rc = hgx_read_alignment_lines(path, regions, n_threads, al);
        if (rc) return rc;
        if (n_threads <= 0) n_threads = hgx_default_threads();
        n_threads = std::max(1, std::min(n_threads, 512));
        const LineVec &lines = al.lines;
        size_t total = 0;
        std::vector<size_t> offs(lines.size() + 1, 0);
        for (size_t i = 0; i < lines.size(); ++i) { offs[i] = total; total += (size_t)lines[i].len + 1; }
        offs[lines.size()] = total;
        char *out = (char *)hgx_host_alloc(total + 1);
        par_for(total > (8u << 20) ? n_threads : 1, lines.size(), [&](int, size_t b, size_t e) {
            for (size_t i = b; i < e; ++i) {
                memcpy(out + offs[i], lines[i].p, lines[i].len);
                out[offs[i] + lines[i].len] = '\n';
            }
        });
        out[total] = 0;
        *text_out = out;
        *n_bytes_out = total;
        return HGX_OK;
    } catch (const std::exception &e) {
        hgx_set_error("hgx_read_alignments: %s", e.what());
        return HGX_ENOMEM;
    }
}

extern "C" int hgx_free_text(char *text) {
    hgx_host_free(text);
    return HGX_OK;
}

// ---- BAM writer -----------------------------------------------------------------------------------------------------
// SAM text (records; header lines are skipped) -> BAM: records encoded in parallel, BGZF blocks of <= 0xff00 payload bytes
// deflated in parallel (zlib), one EOF block.  Optionally the records are ordered by (reference, position) first, as
// `samtools sort` leaves an alignment file -- what the reference's pipeline stores (typing_common.py:1041-1050).  The
// pure-Python statement of the same format is bamio.write_bam.
namespace {

inline void put32(std::vector<unsigned char> &o, uint32_t v) { o.push_back(v & 255); o.push_back((v >> 8) & 255); o.push_back((v >> 16) & 255); o.push_back(v >> 24); }
inline void put16(std::vector<unsigned char> &o, uint32_t v) { o.push_back(v & 255); o.push_back((v >> 8) & 255); }

int reg2bin(int64_t beg, int64_t end) {
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

struct SamRec { const char *p; uint32_t len; int32_t rid; int32_t pos0; };

// one SAM line -> one BAM record (block_size word included) appended to `o`; false if the line is malformed
bool encode_record(const char *line, size_t len, const std::vector<std::string> &refs, std::vector<unsigned char> &o) {
    const char *f[64];
    size_t fl[64];
    int nf = 0;
    const char *p = line, *end = line + len;
    while (p <= end && nf < 64) {
        const char *q = (const char *)memchr(p, '\t', (size_t)(end - p));
        if (!q) q = end;
        f[nf] = p; fl[nf++] = (size_t)(q - p);
        p = q + 1;
    }
    if (nf < 11) return false;
    auto ref_id = [&](const char *s, size_t n) -> int32_t {
        for (size_t i = 0; i < refs.size(); ++i) if (refs[i].size() == n && memcmp(refs[i].data(), s, n) == 0) return (int32_t)i;
        return -1;
    };
    const int32_t rid = ref_id(f[2], fl[2]);
    const int32_t nid = (fl[6] == 1 && f[6][0] == '=') ? rid : ref_id(f[6], fl[6]);
    const long flag = strtol(f[1], nullptr, 10), pos0 = strtol(f[3], nullptr, 10) - 1, mapq = strtol(f[4], nullptr, 10);
    const long pnext0 = strtol(f[7], nullptr, 10) - 1, tlen = strtol(f[8], nullptr, 10);
    std::vector<uint32_t> cig;
    int64_t reflen = 0;
    if (!(fl[5] == 1 && f[5][0] == '*')) {
        static const char ops[] = "MIDNSHP=X";
        uint64_t n = 0;
        for (size_t k = 0; k < fl[5]; ++k) {
            const char c = f[5][k];
            if (c >= '0' && c <= '9') { n = n * 10 + (uint64_t)(c - '0'); continue; }
            const char *w = strchr(ops, c);
            if (!w || !c) return false;
            const uint32_t op = (uint32_t)(w - ops);
            cig.push_back((uint32_t)(n << 4) | op);
            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) reflen += (int64_t)n;
            n = 0;
        }
    }
    const bool no_seq = fl[9] == 1 && f[9][0] == '*';
    const size_t l_seq = no_seq ? 0 : fl[9];
    const size_t at = o.size();
    put32(o, 0);                                          // block_size, patched below
    put32(o, (uint32_t)rid);
    put32(o, (uint32_t)pos0);
    o.push_back((unsigned char)(fl[0] + 1));
    o.push_back((unsigned char)mapq);
    put16(o, (uint32_t)reg2bin(pos0, pos0 + std::max<int64_t>(reflen, 1)));
    put16(o, (uint32_t)cig.size());
    put16(o, (uint32_t)flag);
    put32(o, (uint32_t)l_seq);
    put32(o, (uint32_t)nid);
    put32(o, (uint32_t)pnext0);
    put32(o, (uint32_t)tlen);
    o.insert(o.end(), f[0], f[0] + fl[0]);
    o.push_back(0);
    for (uint32_t c : cig) put32(o, c);
    if (!no_seq) {
        static const struct Code { unsigned char t[256]; Code() { memset(t, 15, 256); const char *s = "=ACMGRSVTWYHKDBN"; for (int i = 0; i < 16; ++i) t[(unsigned char)s[i]] = (unsigned char)i; } } code;
        for (size_t k = 0; k < l_seq; k += 2) {
            const unsigned hi = code.t[(unsigned char)f[9][k]], lo = k + 1 < l_seq ? code.t[(unsigned char)f[9][k + 1]] : 0;
            o.push_back((unsigned char)((hi << 4) | lo));
        }
        if (fl[10] == 1 && f[10][0] == '*') o.insert(o.end(), l_seq, (unsigned char)0xff);
        else {
            if (fl[10] != l_seq) return false;
            for (size_t k = 0; k < l_seq; ++k) o.push_back((unsigned char)(f[10][k] - 33));
        }
    }
    for (int k = 11; k < nf; ++k) {                       // TAG:TYPE:VALUE
        if (fl[k] < 5 || f[k][2] != ':' || f[k][4] != ':') return false;
        const char t = f[k][3];
        const char *v = f[k] + 5;
        const size_t vl = fl[k] - 5;
        o.push_back((unsigned char)f[k][0]);
        o.push_back((unsigned char)f[k][1]);
        if (t == 'i') {                                   // the smallest type that holds the value, as samtools chooses
            const long long x = strtoll(v, nullptr, 10);
            if (x >= 0 && x <= 255) { o.push_back('C'); o.push_back((unsigned char)x); }
            else if (x >= -128 && x <= 127) { o.push_back('c'); o.push_back((unsigned char)(int8_t)x); }
            else if (x >= 0 && x <= 65535) { o.push_back('S'); put16(o, (uint32_t)x); }
            else if (x >= -32768 && x <= 32767) { o.push_back('s'); put16(o, (uint32_t)(uint16_t)(int16_t)x); }
            else if (x >= 0 && x <= 4294967295ll) { o.push_back('I'); put32(o, (uint32_t)x); }
            else if (x >= -2147483648ll && x <= 2147483647ll) { o.push_back('i'); put32(o, (uint32_t)(int32_t)x); }
            else return false;
        } else if (t == 'Z' || t == 'H') { o.push_back((unsigned char)t); o.insert(o.end(), v, v + vl); o.push_back(0); }
        else if (t == 'A') { if (vl != 1) return false; o.push_back('A'); o.push_back((unsigned char)v[0]); }
        else if (t == 'f') { o.push_back('f'); const float x = strtof(v, nullptr); uint32_t u; memcpy(&u, &x, 4); put32(o, u); }
        else if (t == 'B') {
            if (vl < 1) return false;
            const char st = v[0];
            o.push_back('B');
            o.push_back((unsigned char)st);
            const size_t cnt_at = o.size();
            put32(o, 0);
            uint32_t cnt = 0;
            const char *q = v + 1, *ve = v + vl;
            while (q < ve) {
                if (*q != ',') return false;
                ++q;
                char *e2;
                if (st == 'f') { const float x = strtof(q, &e2); uint32_t u; memcpy(&u, &x, 4); put32(o, u); }
                else {
                    const long long x = strtoll(q, &e2, 10);
                    if (st == 'c' || st == 'C') o.push_back((unsigned char)x);
                    else if (st == 's' || st == 'S') put16(o, (uint32_t)(uint16_t)x);
                    else if (st == 'i' || st == 'I') put32(o, (uint32_t)x);
                    else return false;
                }
                if (e2 == q) return false;
                q = e2;
                ++cnt;
            }
            o[cnt_at] = cnt & 255; o[cnt_at + 1] = (cnt >> 8) & 255; o[cnt_at + 2] = (cnt >> 16) & 255; o[cnt_at + 3] = cnt >> 24;
        } else return false;
    }
    const uint32_t bs = (uint32_t)(o.size() - at - 4);
    o[at] = bs & 255; o[at + 1] = (bs >> 8) & 255; o[at + 2] = (bs >> 16) & 255; o[at + 3] = bs >> 24;
    return true;
}

}   // namespace

extern "C" int hgx_write_bam(const char *path, const char *sam, size_t n_bytes, const char *ref_names, const int32_t *ref_lens,
                             int32_t n_refs, int32_t sort_by_coordinate, int32_t n_threads) {
    HARGCHK(path && (sam || n_bytes == 0) && n_refs >= 0 && (n_refs == 0 || (ref_names && ref_lens)));
    try {
        if (n_threads <= 0) n_threads = hgx_default_threads();
        std::vector<std::string> refs;
        {
            const char *p = ref_names;
            for (int i = 0; i < n_refs; ++i) {
                const char *e = strchr(p, '\n');
                if (!e) e = p + strlen(p);
                refs.emplace_back(p, e);
                p = *e ? e + 1 : e;
            }
        }
        std::vector<SamRec> recs;
        for (const char *p = sam, *end = sam + n_bytes; p < end;) {
            const char *e = (const char *)memchr(p, '\n', (size_t)(end - p));
            if (!e) e = end;
            size_t len = (size_t)(e - p);
            if (len && p[len - 1] == '\r') --len;
            if (len && *p != '@') recs.push_back(SamRec{p, (uint32_t)len, 0, 0});
            p = e + 1;
        }
        if (sort_by_coordinate) {
            par_for(n_threads, recs.size(), [&](int, size_t b, size_t e) {
                for (size_t i = b; i < e; ++i) {
                    const char *t1 = (const char *)memchr(recs[i].p, '\t', recs[i].len);
                    const char *t2 = t1 ? (const char *)memchr(t1 + 1, '\t', recs[i].len - (size_t)(t1 + 1 - recs[i].p)) : nullptr;
                    const char *t3 = t2 ? (const char *)memchr(t2 + 1, '\t', recs[i].len - (size_t)(t2 + 1 - recs[i].p)) : nullptr;
                    int32_t rid = 0x7fffffff;                              // unplaced records last, as samtools sort puts them
                    if (t3)
                        for (size_t k = 0; k < refs.size(); ++k)
                            if (refs[k].size() == (size_t)(t3 - t2 - 1) && memcmp(refs[k].data(), t2 + 1, refs[k].size()) == 0) { rid = (int32_t)k; break; }
                    recs[i].rid = rid;
                    recs[i].pos0 = t3 ? (int32_t)strtol(t3 + 1, nullptr, 10) - 1 : 0;
                }
            });
            std::stable_sort(recs.begin(), recs.end(), [](const SamRec &a, const SamRec &b) {
                return a.rid != b.rid ? a.rid < b.rid : a.pos0 < b.pos0;
            });
        }
        // header
        std::vector<unsigned char> head;
        head.insert(head.end(), {'B', 'A', 'M', 1});
        std::string text = sort_by_coordinate ? "@HD\tVN:1.6\tSO:coordinate\n" : "";
        for (int i = 0; i < n_refs; ++i) text += "@SQ\tSN:" + refs[i] + "\tLN:" + std::to_string(ref_lens[i]) + "\n";
        put32(head, (uint32_t)text.size());
        head.insert(head.end(), text.begin(), text.end());
        put32(head, (uint32_t)n_refs);
        for (int i = 0; i < n_refs; ++i) {
            put32(head, (uint32_t)refs[i].size() + 1);
            head.insert(head.end(), refs[i].begin(), refs[i].end());
            head.push_back(0);
            put32(head, (uint32_t)ref_lens[i]);
        }
        // records, encoded per worker range
        const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_threads, recs.size() / 2000 + 1));
        std::vector<std::vector<unsigned char>> enc(T);
        std::vector<int> bad(T, 0);
        par_for(T, recs.size(), [&](int t, size_t b, size_t e) {
            enc[t].reserve((e - b) * 260);
            for (size_t i = b; i < e; ++i)
                if (!encode_record(recs[i].p, recs[i].len, refs, enc[t])) { bad[t] = 1; return; }
        });
        for (int v : bad) if (v) { hgx_set_error("malformed SAM record: cannot be written as BAM"); return HGX_EPARSE; }
        // one byte stream -> BGZF blocks
        size_t total = head.size();
        std::vector<size_t> off(T + 1, 0);
        for (int t = 0; t < T; ++t) { off[t] = total; total += enc[t].size(); }
        Bytes raw;
        raw.alloc(total);
        memcpy(raw.data(), head.data(), head.size());
        par_for(T, (size_t)T, [&](int, size_t b, size_t e) { for (size_t t = b; t < e; ++t) if (!enc[t].empty()) memcpy(raw.data() + off[t], enc[t].data(), enc[t].size()); });
        const size_t BLK = 0xff00;
        const size_t n_blocks = (total + BLK - 1) / BLK;
        std::vector<std::vector<unsigned char>> comp(n_blocks);
        std::vector<int> zbad(std::max(1, n_threads), 0);
        par_for(n_threads, n_blocks, [&](int t, size_t b, size_t e) {
            for (size_t k = b; k < e; ++k) {
                const unsigned char *src = raw.data() + k * BLK;
                const size_t n = std::min(BLK, total - k * BLK);
                std::vector<unsigned char> &o = comp[k];
                o.resize(18 + compressBound((uLong)n) + 8);
                z_stream zs;
                memset(&zs, 0, sizeof zs);
                if (deflateInit2(&zs, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { zbad[t] = 1; return; }
                zs.next_in = const_cast<unsigned char *>(src);
                zs.avail_in = (uInt)n;
                zs.next_out = o.data() + 18;
                zs.avail_out = (uInt)(o.size() - 18);
                const int rc = deflate(&zs, Z_FINISH);
                const size_t clen = zs.total_out;
                deflateEnd(&zs);
                if (rc != Z_STREAM_END || 18 + clen + 8 > 65536) { zbad[t] = 1; return; }
                static const unsigned char hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
                memcpy(o.data(), hdr, 16);
                const uint32_t bsize = (uint32_t)(18 + clen + 8 - 1);
                o[16] = bsize & 255; o[17] = bsize >> 8;
                const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), src, (uInt)n);
                unsigned char *tail = o.data() + 18 + clen;
                tail[0] = crc & 255; tail[1] = (crc >> 8) & 255; tail[2] = (crc >> 16) & 255; tail[3] = crc >> 24;
                tail[4] = n & 255; tail[5] = (n >> 8) & 255; tail[6] = (n >> 16) & 255; tail[7] = (unsigned char)(n >> 24);
                o.resize(18 + clen + 8);
            }
        });
        for (int v : zbad) if (v) { hgx_set_error("BGZF deflate failed"); return HGX_EINVAL; }
        FILE *fo = fopen(path, "wb");
        if (!fo) { hgx_set_error("cannot create %s", path); return HGX_EINVAL; }
        bool ok = true;
        for (auto &c : comp) ok = ok && fwrite(c.data(), 1, c.size(), fo) == c.size();
        static const unsigned char eof_block[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        ok = ok && fwrite(eof_block, 1, sizeof eof_block, fo) == sizeof eof_block;
        ok = (fclose(fo) == 0) && ok;
        if (!ok) { hgx_set_error("short write on %s", path); return HGX_EINVAL; }
        return HGX_OK;
    } catch (const std::exception &e) {
        hgx_set_error("hgx_write_bam: %s", e.what());
        return HGX_ENOMEM;
    }
}
