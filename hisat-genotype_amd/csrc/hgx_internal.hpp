// hgx_internal.hpp -- host-side data model shared by the front-end translation units.
#pragma once
#include <algorithm>
#include <array>
#include <atomic>
#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "hgx.h"

extern "C" void hgx_set_error(const char *fmt, ...);
extern "C" const char *hgx_test_switch(const char *name);

#define HARGCHK(cond)                                                                  \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            hgx_set_error("invalid argument: %s (%s:%d)", #cond, __FILE__, __LINE__);  \
            return HGX_EINVAL;                                                         \
        }                                                                              \
    } while (0)

// One equivalent spelling around a deletion: coordinates + variant indices
// ("529-hv8-hv22-606" of typing_common.py:1421 with ids as indices).
struct AltHt {
    int32_t left, right;
    std::vector<int32_t> vars;
    bool operator==(const AltHt &o) const { return left == o.left && right == o.right && vars == o.vars; }
};
struct AltEntry {
    AltHt key;
    std::vector<AltHt> alts;   // set semantics, insertion order irrelevant
};

struct hgx_locus {
    int32_t base_kind = 0;
    std::string backbone;
    int32_t V = 0;
    std::vector<int32_t> pos, right, len;
    std::vector<uint8_t> type, linked;
    std::vector<char> base;
    std::vector<std::string> name, ins;
    std::unordered_map<std::string, int32_t> name_to_var;
    std::vector<int32_t> hv_index;               // "hv<n>" -> variant index (-1 if absent); fast path of the Zs id lookup
    int32_t A = 0, a_pad = 0, n_words = 1, w64 = 0;
    std::vector<int32_t> link_off, link_allele;
    std::vector<int32_t> maxright;               // prefix max of right ends (core:393-401)
    std::vector<std::array<int32_t, 2>> exons;
    std::vector<uint8_t> exonic;
    std::vector<int32_t> av_off, av_var;         // allele -> variants in gene_var_list order (core:476-487)
    std::vector<int32_t> rep_of;                 // allele -> representative allele or -1 (core:86-115)
    std::vector<int32_t> grp_off, grp_member;    // CSR over representatives: members of allele a's exon group (empty unless a is a representative)
    std::vector<uint64_t> exon_mask, gene_mask;
    std::vector<uint32_t> link_bits;             // [n_words][a_pad]
    std::vector<uint32_t> linked_bits;           // [n_words] variants present in Links
    std::vector<int32_t> allele_len, name_rank;
    // alternatives (common:1424-1657), sorted by anchor position like Alts_left_list / Alts_right_list
    std::vector<AltEntry> alts_left, alts_right;
    bool alts_built = false;
    // the device front end's tables of this locus (hgx_front.hip), made on first use: owned here, freed through fe_dev_free
    mutable void *fe_dev = nullptr;
    mutable void (*fe_dev_free)(void *) = nullptr;
    hgx_locus() = default;
    hgx_locus(const hgx_locus &) = delete;
    hgx_locus &operator=(const hgx_locus &) = delete;
    ~hgx_locus() { if (fe_dev && fe_dev_free) fe_dev_free(fe_dev); }

    bool carries(int32_t allele, int32_t v) const {
        return (link_bits[(size_t)(v >> 5) * a_pad + allele] >> (v & 31)) & 1u;
    }
};

// typing_common.py:406-422 on the position column of the (static) variant list
inline int32_t lower_bound_pos(const std::vector<int32_t> &pos, int32_t key) {
    int32_t low = 0, high = (int32_t)pos.size();
    while (low < high) {
        int32_t m = (low + high) / 2;
        if (pos[m] < key) low = m + 1;
        else high = m;
    }
    return low;
}

// Distinct-piece table: open addressing over piece ids, keys compared against the batch's own mask pool
// (no allocation per lookup: the front-end interns a few pieces per read pair).
struct PieceTable {
    std::vector<int32_t> slot;     // piece id or -1
    size_t used = 0;
    static uint64_t hash(uint16_t lo, uint16_t nw, const uint32_t *m) {
        uint64_t h = 1469598103934665603ull ^ lo ^ ((uint64_t)nw << 16);
        for (int i = 0; i < 2 * (int)nw; ++i) { h ^= m[i]; h *= 1099511628211ull; h ^= h >> 29; }
        return h;
    }
    void clear() { slot.clear(); used = 0; }
};

struct TraceRec {
    std::string text;
};

// Host block pool (hgx_host.cpp): the ingestion path allocates and drops several buffers of hundreds of MB per sample; handing
// them back to the kernel (munmap) and faulting fresh pages in costs more than the work done on them.  Blocks >= 1 MB are kept
// and reused (best fit within 2x); hgx_pool_trim() releases them.
void *hgx_host_alloc(size_t bytes);
void hgx_host_free(void *p);
void hgx_host_pool_trim();
template <class T>
struct HostPoolAlloc {
    typedef T value_type;
    HostPoolAlloc() = default;
    template <class U> HostPoolAlloc(const HostPoolAlloc<U> &) {}
    T *allocate(size_t n) { return (T *)hgx_host_alloc(n * sizeof(T)); }
    void deallocate(T *p, size_t) { hgx_host_free(p); }
    template <class U> bool operator==(const HostPoolAlloc<U> &) const { return true; }
    template <class U> bool operator!=(const HostPoolAlloc<U> &) const { return false; }
};
typedef std::basic_string<char, std::char_traits<char>, HostPoolAlloc<char>> PString;
// vectors of a batch: MBs each, handed back with every sample.  With 30+ threads in the process an munmap is a round of TLB
// shoot-down interrupts: destroying one batch took 5-6 ms of a 65 ms file -> result call before its arrays came from the pool.
template <class T> using PVec = std::vector<T, HostPoolAlloc<T>>;
// ... and for tables that are overwritten right after resize(): elements are default-initialised (no zero fill: value-initialising
// the 24 MB line table of a 1 M-record file was 2-3 ms on one thread)
template <class T>
struct HostPoolRawAlloc : HostPoolAlloc<T> {
    template <class U> struct rebind { typedef HostPoolRawAlloc<U> other; };
    HostPoolRawAlloc() = default;
    template <class U> HostPoolRawAlloc(const HostPoolRawAlloc<U> &) {}
    template <class U> void construct(U *p) { ::new ((void *)p) U; }
    template <class U, class... Args> void construct(U *p, Args &&...args) { ::new ((void *)p) U(std::forward<Args>(args)...); }
};
template <class T> using RawVec = std::vector<T, HostPoolRawAlloc<T>>;

struct hgx_batch {
    PVec<hgx_piece> pieces;
    PVec<uint32_t> masks;
    PVec<int32_t> pair_off{0};
    PVec<uint32_t> pair_ref;
    int32_t n_reads = 0;
    PieceTable table;
    std::vector<TraceRec> trace;
    std::vector<uint8_t> nt_set;        // [L] 4-bit masks
    std::vector<uint32_t> counts;       // [L][6]
};

// piece "left-ids-right" -> index of the distinct piece in the batch (creates it if new). < 0 on error.
int64_t hgx_intern_piece(hgx_batch &b, const hgx_locus &loc, int32_t left, int32_t right, const int32_t *ids, int32_t n_ids);

// Persistent host worker pool (hgx_host.cpp): body(worker) runs on `n` threads (worker 0 = the caller) and the call returns when
// all are done.  The helpers below cut [0, n_items) into contiguous ranges / hand out task indices dynamically.
void hgx_run_workers(int n, const std::function<void(int)> &body);
int hgx_default_threads();          // host threads for a parallel phase: hardware threads, capped by the cgroup CPU quota (x2)
template <class F>
inline void hgx_par_ranges(int n_threads, size_t n_items, F fn) {          // fn(thread, begin, end)
    n_threads = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(n_threads, 1), n_items));
    if (n_threads == 1) { fn(0, (size_t)0, n_items); return; }
    hgx_run_workers(n_threads, [&](int t) { fn(t, n_items * (size_t)t / n_threads, n_items * (size_t)(t + 1) / n_threads); });
}
template <class F>
inline void hgx_par_tasks(int n_threads, size_t n_tasks, F fn) {           // fn(thread, task): tasks pulled from a shared counter
    n_threads = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(n_threads, 1), n_tasks));
    if (n_threads == 1) { for (size_t k = 0; k < n_tasks; ++k) fn(0, k); return; }
    std::atomic<size_t> next{0};
    hgx_run_workers(n_threads, [&](int t) { for (size_t k; (k = next.fetch_add(1)) < n_tasks;) fn(t, k); });
}

// The alignment reader's internal result (hgx_bam.cpp -> hgx_sam.cpp): the records as a line table, stable-sorted by QNAME, over
// buffers this object owns.  p[len] is a byte the parser may overwrite (the line's terminator).
struct hgx_line { char *p; uint32_t len, klen; uint64_t key; };     // klen = QNAME length, key = its first 8 bytes, big endian
// BGZF container (hgx_inflate.hip: inflate on the device, one wavefront per block)
struct hgx_bgzf_block { size_t in_off, in_len, out_off, out_len; uint32_t crc; };
int hgx_bgzf_scan(const unsigned char *data, size_t n, std::vector<hgx_bgzf_block> &blocks, size_t *total_out);
int hgx_bgzf_scan_par(const unsigned char *data, size_t n, std::vector<hgx_bgzf_block> &blocks, size_t *total_out, int n_threads);   // (the same, ranges on several threads)
// a BAM whose record walk, region filter and name sort are left to the device front end (hgx_front.hip: k_bam_*): the reader stops
// after the inflate and the header
struct hgx_bam_deferred {
    bool on = false;
    bool on_device = false;              // the inflated stream exists only in the caller's device buffer (it inflated the blocks itself)
    size_t body0 = 0;                    // offset of the first record's block_size in the inflated stream
    bool filtered = false;               // ONE region was given: per reference what it keeps
    std::vector<uint8_t> ref_action;     // [n_ref] 0 = drop, 1 = keep, 2 = keep where [pos0, end0] overlaps [left0, right0]
    int64_t left0 = 0, right0 = 0;
    // SAM TEXT left to the device (round 5): the line table -- newline scan, header / blank lines dropped, region filter on RNAME /
    // POS / CIGAR, QNAME order check and sort -- is made by kernels too.  `filtered`: region_whole keeps a whole reference,
    // region_name (non-empty) the records overlapping [left0, right0] on it (a samtools region string reads both ways).
    bool text = false;
    std::string region_whole, region_name;
};
struct hgx_align_lines {
    char *raw = nullptr;                   // SAM text as read, or the inflated BAM stream (pooled block), or null
    std::vector<PString> chunks;           // text decoded from BAM records (hgx_read_alignments only)
    RawVec<hgx_line> lines;
    // BAM records handed over in binary (hgx_parse_alignment_file): lines[i].p = the record's read name (record start + 32,
    // NUL-terminated), lines[i].len = its block_size; ref_names = the header's reference sequences
    bool binary = false;
    std::vector<std::string> ref_names;
    size_t raw_bytes = 0;
    // called as soon as bytes the line table will point into are complete -- a big SAM text in a few consecutive ranges while it is
    // read, a BAM stream once, when it is inflated and before the record walk and the name sort: the device front end uploads
    // [begin, end) of the n_bytes at `raw` there
    std::function<void(const char *raw, size_t n_bytes, size_t begin, size_t end)> on_raw;
    // in: the caller can walk / filter / sort BAM records itself (the device front end) when the inflated stream has at least
    // defer_min_bytes and at most one region was asked for; out: `deferred.on` -- `lines` is empty then
    bool defer_walk = false;
    bool defer_text = false;             // ... and SAM text without a line table (the device front end scans the lines itself)
    size_t defer_min_bytes = 0;
    hgx_bam_deferred deferred;
    // in: with defer_walk -- the caller can also inflate the BGZF blocks itself (hgx_inflate.hip): called with the file's bytes and
    // block table once the BAM header (inflated on the host) says the stream qualifies; returns 0 when the payload now lies in
    // the caller's device buffer (deferred.on_device: `raw` stays NULL, raw_bytes = the payload's size), else the host inflates
    std::function<int(const unsigned char *data, size_t n, const std::vector<hgx_bgzf_block> &blocks, size_t total)> inflate_dev;
    // in, optional: called with the file's bytes as soon as they are read and look like BGZF -- the caller may start sending them
    // while the host still hops through the container and inflates the header (inflate_dev then finds them on their way)
    std::function<void(const unsigned char *data, size_t n)> comp_early;
    // in, optional: called before the reader gives the file's bytes back when comp_early has seen them (its copy must be complete)
    std::function<void()> comp_sync;
    hgx_align_lines() = default;
    hgx_align_lines(const hgx_align_lines &) = delete;
    hgx_align_lines &operator=(const hgx_align_lines &) = delete;
    ~hgx_align_lines() { hgx_host_free(raw); }
};
int hgx_read_alignment_lines(const char *path, const char *regions, int n_threads, hgx_align_lines &out, bool keep_binary = false);
int hgx_deferred_for_regions(const char *regions, bool text, size_t body0, const std::vector<std::string> &refs, hgx_bam_deferred &d);

// find-or-insert a piece given its word range and (MP,P) mask words
uint32_t hgx_intern_masks(hgx_batch &b, uint16_t lo, uint16_t nw, const uint32_t *m);
void hgx_finalize_batch(hgx_batch &b, int n_threads = 1);
void hgx_canonical_piece_order(hgx_batch &b, int n_threads, std::vector<uint32_t> &new_id);
int hgx_batch_merge(hgx_batch **out, const hgx_batch *const *batches, int32_t n, int32_t *pair_base);
// alternatives tables (defined in hgx_sam.cpp)
int hgx_build_alternatives(hgx_locus &loc);

// ---- device front end (hgx_front_core.hpp, hgx_front.hip, hgx_front_host.cpp) ------------------------------------------------
#include "hgx_front_core.hpp"
// host copies of the tables an FeLocus points into (the device path uploads them once per locus and device)
struct hgx_front_tables {
    std::vector<int32_t> exons, name_off;
    std::vector<char> name_pool;
    std::vector<int32_t> alt_anchor[2], alt_key_off[2], alt_str_off[2], alt_list_off[2], alt_ht_off, alt_ints;
    std::vector<char> alt_chars;
    bool usable = false;               // false: the locus has something the device path does not take (see hgx_front_tables_build)
    std::string why;
};
int hgx_front_tables_build(hgx_locus &L, hgx_front_tables &T);
// keep_trace on the device route: the trace records the decode wrote (FePools::trace_pool) -> one line per kept record, in stream
// order, spelled as hgx_sam.cpp spells its own ("cmp_list2 \t cmp_left \t cmp_right \t left alts \t right alts"); novel
// variants are numbered nv<k> in the order the kept records meet them (typing_core.py:404-431)
void hgx_front_trace_lines(const hgx_locus &L, const uint32_t *rec_info, size_t n_rec, const uint8_t *state, const uint32_t *trace_off,
                           const int32_t *pool, std::vector<std::string> &out);
FeLocus hgx_front_view(const hgx_locus &L, const hgx_front_tables &T);       // pointers into L and T (host side)

// What the host stages (split, filters, key grouping) hand to the device stages: the distinct keys that count into the pileup
// or are decoded, in stream order of their first records, with their text; and the records that passed the filters.
// The three arrays live in staging memory obtained from `alloc` (pinned for the device path).
struct hgx_front_alloc { void *(*alloc)(size_t); void (*release)(void *); };
// while alive, this thread's hgx_host_alloc calls of >= min_bytes come from `a` (hgx_host_free gives such blocks back to it)
struct hgx_big_alloc_scope {
    hgx_front_alloc old;
    size_t old_min;
    hgx_big_alloc_scope(hgx_front_alloc a, size_t min_bytes);
    ~hgx_big_alloc_scope();
};
struct hgx_front_input {
    hgx_front_alloc mem{nullptr, nullptr};
    FeKey *keys = nullptr; size_t n_keys = 0;
    char *text = nullptr; size_t n_text = 0;
    uint32_t *rec_info = nullptr; size_t n_rec = 0;
    size_t n_slots = 0;                // distinct keys that are decoded
    // CODIS D18S51 (codis_choose_pairs / interdist_exchange): the histogram of this stream's inner distances (HGX_INTERDIST_BINS
    // counters; the key route's host stages fill it, the record route counts on the device and leaves it empty)
    bool want_interdist = false;
    std::vector<int64_t> interdist_hist;
    bool text_borrowed = false;        // `text` belongs to someone else (the record route reads the file's bytes in place)
    hgx_front_input() = default;
    hgx_front_input(const hgx_front_input &) = delete;
    hgx_front_input &operator=(const hgx_front_input &) = delete;
    ~hgx_front_input() { if (mem.release) { mem.release(keys); if (!text_borrowed) mem.release(text); mem.release(rec_info); } }
};
// > 0: the device front end declines this input (code = an FE_E_* value negated, or one of the HGX_FE_DECLINE_* below)
#define HGX_FE_DECLINE_OPTS 1          // choose_pairs / inter-distance exchange (and keep_trace in a many-task pass): host only
#define HGX_FE_DECLINE_RECORD 2        // a record the reference would raise on
#define HGX_FE_DECLINE_LOCUS 3         // tables the device path does not take
#define HGX_FE_DECLINE_SIZE 4          // more records / keys / text than its 32-bit offsets hold
#define HGX_FE_DECLINE_COLLISION 5     // two different pieces with one 64-bit content key
#define HGX_FE_DECLINE_SMALL 6         // too few records for a dozen launches to pay
// Sharded loci (hgx_parse_opts.pileup_exchange / _dev): every rank must communicate exactly ONCE per parse, whichever route finishes
// it.  The device front end wraps the caller's host callback in this: once a summed table exists (the device route exchanged and
// then declined, or the wrapper itself ran the caller's callback) the host stages get that table instead of a second exchange.
struct hgx_pileup_share {
    int (*orig)(void *, uint32_t *, int64_t) = nullptr;
    void *orig_ctx = nullptr;
    bool have_sum = false;
    std::vector<uint32_t> sum;                 // [L * 6]
    static int trampoline(void *self, uint32_t *counts, int64_t n) {
        hgx_pileup_share *s = (hgx_pileup_share *)self;
        if (s->have_sum) {
            if ((size_t)n != s->sum.size()) return 1;
            std::copy(s->sum.begin(), s->sum.end(), counts);
            return 0;
        }
        if (!s->orig) return 1;
        const int rc = s->orig(s->orig_ctx, counts, n);
        if (rc == 0) { s->sum.assign(counts, counts + n); s->have_sum = true; }
        return rc;
    }
};
// The same for the inter-distance histogram of a sharded CODIS D18S51 sample (hgx_parse_opts.interdist_exchange): the device route
// exchanges after its pileup exchange (the order of the host stages); if a later stage declines, the host stages get the summed
// histogram from here instead of a second exchange.
struct hgx_interdist_share {
    int (*orig)(void *, int64_t *, int64_t) = nullptr;
    void *orig_ctx = nullptr;
    bool have_sum = false;
    std::vector<int64_t> sum;                  // [HGX_INTERDIST_BINS]
    static int trampoline(void *self, int64_t *hist, int64_t n) {
        hgx_interdist_share *s = (hgx_interdist_share *)self;
        if (s->have_sum) {
            if ((size_t)n != s->sum.size()) return 1;
            std::copy(s->sum.begin(), s->sum.end(), hist);
            return 0;
        }
        if (!s->orig) return 1;
        const int rc = s->orig(s->orig_ctx, hist, n);
        if (rc == 0) { s->sum.assign(hist, hist + n); s->have_sum = true; }
        return rc;
    }
};
// Element len / 2 of the sorted distances (typing_common.py:1258-1262) out of their histogram (HGX_INTERDIST_BINS counters); -1 when
// there is no distance.  Returns 1 when that element lies in an edge bin: outside the range the histogram resolves.
int hgx_interdist_median(const int64_t *hist, long long *expected);
// The host stages call `run` after key grouping; it returns HGX_OK with *declined = 0 when the device stages produced the result
// (which the hook's owner holds: the parse functions then return *out = NULL), or *declined = the reason -- the host stages
// then finish the job and `declined` says why.
struct hgx_front_hook {
    hgx_front_alloc mem{nullptr, nullptr};
    std::function<int(hgx_locus &, const hgx_front_input &, const hgx_parse_opts &, int *declined)> run;
    size_t min_records = 0;            // `run` would decline fewer records as too small: the host stages do not even build its input then
    int declined = 0;
    // The RECORD route, tried first when set: the device takes the records themselves (fields, filters, key grouping as kernels
    // over the SAM text / the inflated BAM stream it was sent through `on_raw`), the host stages do not run at all.  `lines`: the
    // name-ordered line table of the reader; raw / raw_bytes: the bytes it points into.  *declined != 0: the host stages run
    // (and `run` gets its chance after them).
    // `def` != NULL: a BAM stream whose records have not been walked (lines == NULL, n == 0): the device does that too.
    std::function<int(hgx_locus &, const char *raw, size_t raw_bytes, const hgx_line *lines, size_t n, bool binary, const hgx_parse_opts &,
                      int *declined, const hgx_bam_deferred *def)> records;
    bool defer_walk = false;           // the hook's owner takes unwalked BAM streams
    bool defer_text = false;           // ... and SAM text without a line table
    size_t defer_min_bytes = 0;
    std::function<int(const unsigned char *data, size_t n, const std::vector<hgx_bgzf_block> &blocks, size_t total)> inflate_dev;   // ... and deflated ones
    std::function<void(const unsigned char *data, size_t n)> comp_early;             // (the deflated bytes, before the container is looked at)
    std::function<void()> comp_sync;                                                 // (... and the wait for that copy, before the bytes are released)
    std::function<void(const char *raw, size_t n_bytes, size_t begin, size_t end)> on_raw;
    int declined_records = 0;
};
int hgx_parse_sam_hook(hgx_batch **out, const hgx_locus *loc, const char *sam, size_t n_bytes, const hgx_parse_opts *opts, hgx_front_hook *hook);
int hgx_parse_alignment_file_hook(hgx_batch **out, const hgx_locus *loc, const char *path, const char *regions, const hgx_parse_opts *opts,
                                  hgx_front_hook *hook);
// the BGZF-compressed BAM files of many tasks, read but not inflated (the device front end inflates, walks, filters and sorts):
// data = the file's bytes (hgx_host_alloc: the caller frees), ok = a BAM whose header (inflated here) was understood and whose region
// list has at most one entry; on_task(t) runs on the reader's thread when task t's bytes are in memory
struct hgx_bgzf_task {
    unsigned char *data = nullptr;
    size_t n = 0;
    std::vector<hgx_bgzf_block> blocks;
    size_t total = 0;
    hgx_bam_deferred def;
    bool ok = false;
};
int hgx_bgzf_tasks_read(std::vector<hgx_bgzf_task> &tasks, const char *const *paths, const char *const *regions, int n_tasks, int n_threads,
                        const hgx_front_alloc *mem, const std::function<void(int)> &on_task);
// ---- the record streams of MANY tasks of one locus as one stream (hgx_many_create_files / _sams: one device pass for all) ------
// what a many-task pass reports per task
struct hgx_front_totals { std::vector<uint32_t> reads, pairs, pieces; std::vector<uint64_t> refs; };
struct hgx_many_streams {
    int n_tasks = 0;
    std::unique_ptr<hgx_align_lines[]> al;       // per task: the reader's line table (and its bytes, when it read a file)
    std::vector<const char *> raw;               // per task: the bytes the lines point into (a file's, or the caller's SAM text)
    std::vector<size_t> raw_bytes;
    std::vector<size_t> base, line_base;         // [n_tasks + 1]: where a task's bytes / lines start in the concatenation (bytes 64-aligned)
    bool binary = false;                         // BAM records (all tasks alike: a mix is refused with `mixed`)
    bool mixed = false;
};
// reads (paths[t], regions[t] or regions NULL) or walks (sams[t], sam_bytes[t]) every task's stream, tasks side by side on the
// host's threads; big reader blocks come from `mem` (pinned staging) when given
// when given; on_task(t), when given, runs on the task's reader thread as soon as raw[t] / raw_bytes[t] are known and sets
// base[t] itself (the device front end reserves a range of its text buffer there and starts the upload); otherwise the tasks'
// bytes are laid out one after the other
int hgx_many_read(hgx_many_streams &ms, const char *const *paths, const char *const *regions, const char *const *sams, const size_t *sam_bytes,
                  int n_tasks, int n_threads, const hgx_front_alloc *mem, const std::function<int(int task)> &on_task = nullptr);
void hgx_many_lines(const hgx_many_streams &ms, FeLine *dst, int n_threads);     // the concatenated line table: (offset, length, task)
struct hgx_dbatch;
// the device pass over them (hgx_front.hip): *declined != 0 -> nothing made, the caller goes task by task through the host stages
int hgx_front_many_dev(hgx_dbatch **out, hgx_front_totals *tot, const hgx_locus *loc, const char *const *paths, const char *const *regions,
                       const char *const *sams, const size_t *sam_bytes, int n_tasks, const hgx_parse_opts *opts, void *stream, int *declined);
#ifdef HGX_LAB
int hgx_front_emulate(hgx_batch **out, hgx_locus &L, const hgx_front_input &in, const hgx_parse_opts &opts, int *declined, int n_tasks = 1,
                      hgx_front_totals *many = nullptr);
int hgx_front_emulate_records(hgx_batch **out, hgx_locus &L, const char *raw, size_t raw_bytes, const FeLine *lines, size_t n, bool binary,
                              const hgx_parse_opts &opts, int *declined, int n_tasks = 1, hgx_front_totals *many = nullptr);
#endif
