// hgx_type.hip -- the per-locus body of typing() behind ONE C-ABI call (hgx_type_batch / hgx_type_dbatch / hgx_type_file).
//
// Replaces typing_core.py:1589-1789 for one locus: Gene_counts and their ranking (core:1187-1190, 1650-1651), the exon-level
// classes and EM #1 (core:1732-1737), the choice of exon_alleles (core:1739-1749), the hand-off Gene_cmpt2 + EM #2 with allele
// lengths (core:1752-1782), the combination of the two results and the tie order of the reference's stable sorts.  It only
// ORCHESTRATES the device entry points of hgx.h (piece compatibility, pair classes, grouping, dedup, allele counts, EM) on
// two side streams plus the caller's stream; no arithmetic of the hot path happens on the host.
//
// Concurrency inside one call (HLA-like loci, >= 4096 pairs): the gene-level side (per-pair rows -> dedup -> Gene_counts ->
// ranking) runs on its own host thread and low-priority stream beside the exon-level grouping, dedup and EM #1 on a
// high-priority stream; both start from an event recorded behind hgx_piece_compat on the caller's stream (device-side
// dependency, the host runs ahead).  Stream pairs are recycled through a process-wide free list, so callers may come from
// short-lived threads (several samples in flight per GPU).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "hgx_common.hpp"
#include "hgx_emx.hpp"
#include "hgx_internal.hpp"

extern "C" int hgx_em_set_fast(int on);
extern "C" int hgx_em_last_order(int32_t *order_host, int32_t n);

struct hgx_gate { std::mutex mu; };

namespace {

struct EmOut {
    int32_t n_classes = 0, n_iter = 0, remove_low = 0, use_length = 0;
    bool exact = false;              // abundances are the reference's bit for bit (single-wavefront EM in the reference's order)
    std::vector<int32_t> allele;
    std::vector<double> prob;
};

constexpr double TIE_REL_TOL = 1e-11;

// The reference's `sorted(..., key=prob, reverse=True)` (a STABLE sort: equal abundances keep dict insertion order).  Alleles the
// data cannot tell apart come out of the reference's EM bit-identical; on the GPU they agree to ~1e-14 only (same arithmetic,
// another summation order), so abundances within a relative 1e-11 of the run's first count as tied and keep insertion order.
// When the values ARE the reference's (`exact`), the plain stable sort is the reference's order and no tolerance applies.
void stable_desc(std::vector<int32_t> &allele, std::vector<double> &prob, bool exact) {
    const double tol = exact ? 0.0 : TIE_REL_TOL;
    const size_t n = allele.size();
    std::vector<size_t> idx(n);
    std::iota(idx.begin(), idx.end(), (size_t)0);
    std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return prob[a] > prob[b]; });
    std::vector<size_t> order;
    order.reserve(n);
    for (size_t i = 0; i < n;) {
        size_t j = i + 1;
        const double top = prob[idx[i]];
        while (j < n && top - prob[idx[j]] <= tol * std::fabs(top)) ++j;
        const size_t at = order.size();
        order.insert(order.end(), idx.begin() + i, idx.begin() + j);
        std::sort(order.begin() + at, order.end());                 // the tied run, back in insertion order
        i = j;
    }
    std::vector<int32_t> a2(n);
    std::vector<double> p2(n);
    for (size_t k = 0; k < n; ++k) { a2[k] = allele[order[k]]; p2[k] = prob[order[k]]; }
    allele.swap(a2);
    prob.swap(p2);
}

// [[allele, prob]] of an EM result: survivors in the insertion order of the reference's dict (first class containing the
// allele, then its place in that class' sorted key: common:1300-1305), then the reference's stable descending sort.
void sorted_result(const std::vector<double> &prob, const std::vector<int32_t> &first, const int32_t *name_rank, int32_t A, EmOut &o) {
    std::vector<int32_t> present;
    for (int32_t a = 0; a < A; ++a) if (prob[a] >= 0.0) present.push_back(a);
    std::sort(present.begin(), present.end(), [&](int32_t a, int32_t b) {
        if (first[a] != first[b]) return first[a] < first[b];
        if (name_rank[a] != name_rank[b]) return name_rank[a] < name_rank[b];
        return a < b;
    });
    o.allele = present;
    o.prob.resize(present.size());
    for (size_t k = 0; k < present.size(); ++k) o.prob[k] = prob[present[k]];
    stable_desc(o.allele, o.prob, o.exact);
}

// ---- stream sets --------------------------------------------------------------------------------------------------------
// A typing call runs on three streams: the caller's (scoring), an EM stream (the exon level's chain of short dependent launches: the
// critical path) and a gene-side stream (chip-filling kernels beside it).  Callers that run side by side (samples in flight, the
// loci of class I) each hold a set.
// What the hardware does with them (measured on MI355X / ROCm 7.2, tools/stream_probe.py and tools/stream_conflicts.py, round 6):
//   * the runtime keeps up to four hardware queues PER PRIORITY and hands a new stream the least-used one of its priority; two
//     streams on one queue run strictly one behind the other (two 150 us spin kernels: 324 us instead of 173);
//   * beyond that, queues fall into four LANES in creation order across all priorities (creation i and i + 4 share one): two CHAINS
//     of short dependent kernels on one lane take 415 us instead of 180 -- each launch waits for the other chain's -- while a chain
//     beside a few big kernels on its lane loses almost nothing (180 -> 200-230 us).
// So what must not share a lane is the EM chains of callers in flight: with two samples in flight a step costs 2.4 ms per sample
// when their EM streams are four creations apart and 1.6 ms when they are not.  Rounds 4-5 arranged that by CREATION ORDER with
// placeholder streams -- right for a fresh process, wrong as soon as the caller (torch, RCCL, its own streams) had created a
// different number of streams first; the driver saw class I at 4.45 ms in one round and 5.03 in the next.
// Round 6 MEASURES: a candidate stream runs a chain of eight 10 us one-wavefront kernels alone and interleaved with the same chain
// on a representative of every lane known so far; the lane where its chain takes > 1.5x as long is its lane.  A set's EM stream is
// the candidate on the lane with the fewest EM streams, its gene-side stream a candidate on a lane without EM streams (or at least
// not its own); up to four candidates each (a creation moves the runtime on by one lane), the others are destroyed again.  State
// per DEVICE.  Test switch streams = "unplaced": first candidates, no probes.
struct StreamSet { int dev = -1; hipStream_t em = nullptr, gene = nullptr; hipEvent_t fork = nullptr; int cls_em = -1, cls_gene = -1, id = 0; };
struct DevStreams {
    std::vector<StreamSet> free;          // taken and given back at the front: a caller alone always gets the same set
    std::vector<hipStream_t> reps;        // one kept stream per lane seen so far
    std::vector<int> em_in_class;         // EM streams of this device's sets per lane
    std::vector<int> caller_in_class;     // callers' own placed streams per lane (hgx_stream_create_placed: front-end chains run there)
    std::map<hipStream_t, int> caller_cls;  // ... and which lane each of them is on (hgx_stream_destroy gives the lane back)
    int n_sets = 0;
    double probe_ms = 0.0;                // time spent placing (hgx_stream_sets_info reports it)
    int probes = 0;
};
std::mutex g_ss_mu;
std::map<int, DevStreams> g_ss;           // by device

__global__ void k_ss_spin(long long ticks, long long *stamp) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
    if (stamp) { stamp[0] = t0; stamp[1] = wall_clock64(); }
}
constexpr long long SS_SPIN_TICKS = 15000;            // 150 us of the 100 MHz constant clock (the queue probe of the diagnostics)
constexpr int SS_CHAIN = 8;
static long long *ss_stamps() {
    static thread_local long long *stamps = nullptr;                       // pinned, the kernels write it directly
    if (!stamps && hipHostMalloc((void **)&stamps, 64 * sizeof(long long), hipHostMallocDefault) != hipSuccess) stamps = nullptr;
    return stamps;
}
// microseconds (the kernels' own clock) stream a's chain of SS_CHAIN 10 us kernels takes, alone (b == nullptr) or with the same chain
// launched on b in between: best of `tries`
static double ss_chain_us(DevStreams &D, hipStream_t a, hipStream_t b, int tries) {
    long long *st = ss_stamps();
    if (!st) return 0.0;
    double best = 1e30;
    for (int k = 0; k < tries; ++k) {
        (void)hipStreamSynchronize(a);
        if (b) (void)hipStreamSynchronize(b);
        for (int i = 0; i < SS_CHAIN; ++i) {
            hipLaunchKernelGGL(k_ss_spin, dim3(1), dim3(64), 0, a, 1000LL, st + 2 * i);
            if (b) hipLaunchKernelGGL(k_ss_spin, dim3(1), dim3(64), 0, b, 1000LL, st + 32 + 2 * i);
        }
        (void)hipStreamSynchronize(a);
        if (b) (void)hipStreamSynchronize(b);
        D.probes++;
        best = std::min(best, (double)(st[2 * SS_CHAIN - 1] - st[0]) / 100.0);
    }
    return best;
}
// true = kernels on a and b run strictly one behind the other (the same hardware queue), by the kernels' own clock stamps: any try
// that shows the two 150 us spins side by side settles it (diagnostics; the placement looks at lanes, which include queues)
static bool ss_same_queue(DevStreams &D, hipStream_t a, hipStream_t b) {
    long long *stamps = ss_stamps();
    if (!stamps) return false;
    for (int k = 0; k < 3; ++k) {
        (void)hipStreamSynchronize(a);
        (void)hipStreamSynchronize(b);
        stamps[0] = stamps[1] = stamps[2] = stamps[3] = 0;
        hipLaunchKernelGGL(k_ss_spin, dim3(1), dim3(64), 0, a, SS_SPIN_TICKS, stamps);
        hipLaunchKernelGGL(k_ss_spin, dim3(1), dim3(64), 0, b, SS_SPIN_TICKS, stamps + 2);
        (void)hipStreamSynchronize(a);
        (void)hipStreamSynchronize(b);
        D.probes++;
        if (stamps[2] < stamps[1] && stamps[0] < stamps[3]) return false;    // the two intervals overlap: different queues
    }
    return true;
}
// lane of stream s among D.reps (-1: none of them: a lane nobody of ours is on yet)
static int ss_class_of(DevStreams &D, hipStream_t s) {
    if (D.reps.empty()) return -1;
    const double alone = ss_chain_us(D, s, nullptr, 2);
    for (size_t c = 0; c < D.reps.size(); ++c) {
        double beside = ss_chain_us(D, s, D.reps[c], 1);              // (one try when the answer is clear: ~1.0x or ~2.3x)
        if (beside > 1.25 * alone && beside < 1.9 * alone) beside = std::min(beside, ss_chain_us(D, s, D.reps[c], 2));
        if (beside > 1.5 * alone) return (int)c;
    }
    return -1;
}
static int make_gene_stream(hipStream_t *out, int least) {
    // The gene side's chip-filling kernels (k_pair_classes_x2, k_verify_ht) run beside the EM chain of the exon level; a 1 024-thread
    // workgroup of an EM pass is only placed on a CU that has sixteen free wave slots and 135 KB of LDS, so with the gene side free to
    // take every CU the first EM passes wait for it to drain.  HGX_GENE_CUS=n confines the gene-side stream to n CUs (CU mask).
    const char *gcu = getenv("HGX_GENE_CUS");
    const int n_gene_cus = gcu ? atoi(gcu) : 0;
    if (n_gene_cus > 0 && n_gene_cus < 256) {
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int b = 0; b < n_gene_cus; ++b) mask[b >> 5] |= 1u << (b & 31);
        HIPCHK(hipExtStreamCreateWithCUMask(out, 8, mask));
    } else
        HIPCHK(hipStreamCreateWithPriority(out, hipStreamNonBlocking, least));
    return HGX_OK;
}
// One more set for device `dev`.  The EM chain gets the highest priority, the overlapped side work the lowest.
static int make_stream_set(int dev, DevStreams &D, StreamSet &set) {
    const auto t0 = std::chrono::steady_clock::now();
    int least = 0, greatest = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    const bool probe = !hgx_switch_has("streams", "unplaced");
    set = StreamSet();
    set.dev = dev;
    std::vector<hipStream_t> spare;
    struct Spare { std::vector<hipStream_t> &v; ~Spare() { for (hipStream_t x : v) (void)hipStreamDestroy(x); } } spare_guard{spare};
    auto keep_class = [&](hipStream_t s, int cls) -> int {                // a kept stream on a new lane becomes its representative
        if (cls >= 0) return cls;
        D.reps.push_back(s);
        D.em_in_class.push_back(0);
        D.caller_in_class.push_back(0);
        return (int)D.reps.size() - 1;
    };
    // EM stream: the lane fewest EM streams of this device are on (a new lane counts as unused)
    // (the runtime has four queues per priority, each on the lane it was created on: if the process' high-priority queues sit on
    // two lanes only -- it depends on what was created between them -- no high-priority candidate ever reaches a third one; the EM
    // stream then takes the normal or the low priority: a lane of its own is worth more than the priority, tools/stream_ab.sh)
    int best_cls = -2, best_load = 1 << 30;
    const int prios[3] = {greatest, (least + greatest) / 2, least};
    // (once all four lanes are known, the best a candidate can do is the least-loaded lane's count: no point in trying further for 0)
    int target = 0;
    if (D.reps.size() >= 4) { target = 1 << 30; for (size_t c = 0; c < D.reps.size(); ++c) target = std::min(target, 16 * D.em_in_class[c] + std::min(15, D.caller_in_class[c])); }
    for (int pk = 0; pk < (probe && D.n_sets > 0 ? 3 : 1) && best_load > target; ++pk)
        for (int k = 0; k < (probe && D.n_sets > 0 ? 4 : 1); ++k) {
            hipStream_t c = nullptr;
            HIPCHK(hipStreamCreateWithPriority(&c, hipStreamNonBlocking, prios[pk]));
            const int cls = probe ? ss_class_of(D, c) : -1;
            const int load = cls < 0 ? 0 : 16 * D.em_in_class[(size_t)cls] + std::min(15, D.caller_in_class[(size_t)cls]);   // EM chains first, then callers' chains
            if (load < best_load) {
                if (set.em) spare.push_back(set.em);
                set.em = c; best_cls = cls; best_load = load;
            } else spare.push_back(c);
            if (best_load <= target) break;
        }
    set.cls_em = probe ? keep_class(set.em, best_cls) : -1;
    if (probe) D.em_in_class[(size_t)set.cls_em]++;
    // gene-side stream: a lane without EM streams; failing that, any lane but this set's own EM lane
    int g_cls = -2, g_score = -1;
    for (int k = 0; k < (probe ? 4 : 1); ++k) {
        hipStream_t c = nullptr;
        { const int rc_ = make_gene_stream(&c, least); if (rc_) return rc_; }
        const int cls = probe ? ss_class_of(D, c) : -1;
        const int score = !probe ? 2 : (cls < 0 || D.em_in_class[(size_t)cls] == 0) ? 2 : (cls != set.cls_em ? 1 : 0);
        if (score > g_score) {
            if (set.gene) spare.push_back(set.gene);
            set.gene = c; g_cls = cls; g_score = score;
        } else spare.push_back(c);
        if (g_score == 2) break;
    }
    set.cls_gene = probe ? keep_class(set.gene, g_cls) : -1;
    HIPCHK(hipEventCreateWithFlags(&set.fork, hipEventDisableTiming));
    set.id = D.n_sets++;
    D.probe_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return HGX_OK;
}
}   // namespace
// A stream for a CALLER that runs chains of short kernels of its own beside other callers' (a worker thread's main stream: the device
// front end of its sample or locus is such a chain): created on the lane with the fewest chains so far -- EM streams of the stream
// sets and other callers' placed streams.  Plain hgx_stream_create_prio streams land wherever the runtime's creation order puts them.
extern "C" int hgx_stream_create_placed(void **st, int high_priority) {
    ARGCHK(st != nullptr);
    *st = nullptr;
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    int least = 0, greatest = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    std::lock_guard<std::mutex> g(g_ss_mu);
    DevStreams &D = g_ss[dev];
    const auto t0 = std::chrono::steady_clock::now();
    const bool probe = !hgx_switch_has("streams", "unplaced");
    std::vector<hipStream_t> spare;
    struct Spare { std::vector<hipStream_t> &v; ~Spare() { for (hipStream_t x : v) (void)hipStreamDestroy(x); } } spare_guard{spare};
    hipStream_t best = nullptr;
    int best_cls = -2, best_load = 1 << 30;
    int target = 0;
    if (D.reps.size() >= 4) { target = 1 << 30; for (size_t c = 0; c < D.reps.size(); ++c) target = std::min(target, D.em_in_class[c] + D.caller_in_class[c]); }
    for (int k = 0; k < (probe ? 4 : 1); ++k) {
        hipStream_t c = nullptr;
        HIPCHK(hipStreamCreateWithPriority(&c, hipStreamNonBlocking, high_priority ? greatest : least));
        const int cls = probe ? ss_class_of(D, c) : -1;
        const int load = cls < 0 ? 0 : D.em_in_class[(size_t)cls] + D.caller_in_class[(size_t)cls];
        if (load < best_load) {
            if (best) spare.push_back(best);
            best = c; best_cls = cls; best_load = load;
        } else spare.push_back(c);
        if (best_load <= target) break;
    }
    if (probe) {
        if (best_cls < 0) { D.reps.push_back(best); D.em_in_class.push_back(0); D.caller_in_class.push_back(0); best_cls = (int)D.reps.size() - 1; }
        D.caller_in_class[(size_t)best_cls]++;
        D.caller_cls[best] = best_cls;
    }
    D.probe_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    *st = (void *)best;
    return HGX_OK;
}
// hgx_stream_destroy asks: a placed caller stream gives its lane back; one that REPRESENTS a lane (later probes run beside it) is kept
// alive by the library instead of being destroyed -- true = do not destroy
bool hgx_ss_forget_stream(void *stream) {
    std::lock_guard<std::mutex> g(g_ss_mu);
    for (auto &kv : g_ss) {
        DevStreams &D = kv.second;
        auto it = D.caller_cls.find((hipStream_t)stream);
        if (it == D.caller_cls.end()) continue;
        if (D.caller_in_class[(size_t)it->second] > 0) D.caller_in_class[(size_t)it->second]--;
        D.caller_cls.erase(it);
        for (hipStream_t r : D.reps) if (r == (hipStream_t)stream) return true;
        return false;
    }
    return false;
}
namespace {
int acquire_streams(StreamSet &s) {
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> g(g_ss_mu);
    DevStreams &D = g_ss[dev];
    if (!D.free.empty()) { s = D.free.front(); D.free.erase(D.free.begin()); return HGX_OK; }
    // creating (and placing) a set takes milliseconds: done once, then recycled
    StreamSet made;
    const int rc = make_stream_set(dev, D, made);
    if (rc) {                                                                // nothing half-made is handed out or leaked
        if (made.em) (void)hipStreamDestroy(made.em);
        if (made.gene) (void)hipStreamDestroy(made.gene);
        if (made.fork) (void)hipEventDestroy(made.fork);
        return rc;
    }
    s = made;
    return HGX_OK;
}
void release_streams(const StreamSet &s) {
    if (s.dev < 0) return;
    std::lock_guard<std::mutex> g(g_ss_mu);
    DevStreams &D = g_ss[s.dev];
    // sets go back in the order they were made (the first, best-placed ones in front)
    auto it = D.free.begin();
    while (it != D.free.end() && it->id < s.id) ++it;
    D.free.insert(it, s);
}

struct GateHold {                    // a held gate that is released exactly once
    hgx_gate *g = nullptr;
    explicit GateHold(hgx_gate *gate) : g(gate) { if (g) g->mu.lock(); }
    void release() { if (g) { g->mu.unlock(); g = nullptr; } }
    ~GateHold() { release(); }
};

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// hgx_type_opts.em_fast as the library uses it inside: 0 = the reference's order (one-workgroup problems), 1 = table lookups, -1 = the
// reference's order at every size.  The option's 0 means "the default of the entry point": the reference's order for the one-task
// calls, table lookups for the many-task calls (the throughput API); 2 asks for the reference's order there too.
inline int em_mode_one(int em_fast) { return em_fast == 2 ? 0 : em_fast; }
inline int em_mode_many(int em_fast) { return em_fast == 0 ? 1 : (em_fast == 2 ? 0 : em_fast); }

struct EmFastScope {                 // hgx_type_opts.em_fast for the EMs of this call (this thread)
    int old;
    explicit EmFastScope(int on) : old(hgx_em_set_fast(on)) {}
    ~EmFastScope() { hgx_em_set_fast(old); }
};

}   // namespace

struct hgx_typing {
    int32_t n_reads = 0, n_pairs = 0, n_pieces = 0, n_alleles = 0;
    int64_t n_refs = 0;
    mutable std::vector<int32_t> counted;        // alleles with a non-zero Gene_count, in the reference's print order
    std::vector<int64_t> cnt;            // Gene_counts per allele index [n_alleles]
    // hgx_type_many leaves the ranking (core:1650-1651: a sort of every counted allele) to the first caller that asks for it
    mutable std::vector<int32_t> first_pair;     // per allele: first pair of the first class containing it (the tie order)
    mutable bool ranked = true;
    mutable std::mutex rank_mu;
    void ensure_ranked() const {
        std::lock_guard<std::mutex> g(rank_mu);
        if (ranked) return;
        const int32_t A = (int32_t)cnt.size();
        counted.clear();
        for (int32_t a = 0; a < A; ++a) if (cnt[a] > 0) counted.push_back(a);
        const int64_t *c = cnt.data();
        const int32_t *fp = first_pair.data();
        std::sort(counted.begin(), counted.end(), [&](int32_t a, int32_t b) {
            if (c[a] != c[b]) return c[a] > c[b];
            if (fp[a] != fp[b]) return fp[a] < fp[b];
            return a < b;
        });
        first_pair.clear();
        first_pair.shrink_to_fit();
        ranked = true;
    }
    std::vector<EmOut> em;
    EmOut gene_prob;                     // final Gene_prob (allele, prob), sorted
    hgx_classes *exon_cl = nullptr, *gene_cl = nullptr;      // kept alive with keep_classes
    double t_em = 0.0;
    char err_thread[512] = "";
    ~hgx_typing() { hgx_classes_destroy(exon_cl); hgx_classes_destroy(gene_cl); }
};

// ---- device batch ---------------------------------------------------------------------------------------------------
extern "C" int hgx_dbatch_destroy(hgx_dbatch *d) {
    if (!d) return HGX_OK;
    hgx_pool_free(d->d_pieces); hgx_pool_free(d->d_masks); hgx_pool_free(d->d_pair_off); hgx_pool_free(d->d_pair_ref);
    hgx_pool_free(d->d_counts); hgx_pool_free(d->d_nt_set);
    delete d;
    return HGX_OK;
}

extern "C" int hgx_dbatch_create(hgx_dbatch **out, const hgx_batch *b, void *stream) {
    ARGCHK(out && b);
    *out = nullptr;
    hipStream_t st = (hipStream_t)stream;
    hgx_dbatch *d = new hgx_dbatch();
    d->n_pieces = (int32_t)b->pieces.size();
    d->n_pairs = (int32_t)b->pair_off.size() - 1;
    d->n_reads = b->n_reads;
    d->n_refs = (int64_t)b->pair_ref.size();
    d->n_mask_u32 = (int64_t)b->masks.size();
    for (const auto &p : b->pieces) d->sum_piece_words += p.n_words;
    for (uint32_t r : b->pair_ref) d->n_gene_refs += r >> 31;
    for (const auto &t : b->trace) d->trace.push_back(t.text);
    auto up = [&](void **dst, const void *src, size_t bytes) -> int {
        *dst = hgx_pool_alloc(std::max<size_t>(bytes, 16));
        if (!*dst) { hgx_set_error("device allocation of %zu bytes failed", bytes); return HGX_ENOMEM; }
        if (bytes) HIPCHK(hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, st));
        return HGX_OK;
    };
    int rc = up((void **)&d->d_pieces, b->pieces.data(), b->pieces.size() * sizeof(hgx_piece));
    if (!rc) rc = up((void **)&d->d_masks, b->masks.data(), b->masks.size() * 4);
    if (!rc) rc = up((void **)&d->d_pair_off, b->pair_off.data(), b->pair_off.size() * 4);
    if (!rc) rc = up((void **)&d->d_pair_ref, b->pair_ref.data(), b->pair_ref.size() * 4);
    if (!rc && hipStreamSynchronize(st) != hipSuccess) { hgx_set_error("upload of the piece batch failed"); rc = HGX_EHIP; }
    if (rc) { hgx_dbatch_destroy(d); return rc; }
    *out = d;
    return HGX_OK;
}

extern "C" int hgx_dbatch_dims(const hgx_dbatch *d, int32_t *n_pieces, int32_t *n_pairs, int64_t *n_refs, int32_t *n_reads,
                               int64_t *sum_piece_words, int64_t *n_gene_refs) {
    ARGCHK(d);
    if (n_pieces) *n_pieces = d->n_pieces;
    if (n_pairs) *n_pairs = d->n_pairs;
    if (n_refs) *n_refs = d->n_refs;
    if (n_reads) *n_reads = d->n_reads;
    if (sum_piece_words) *sum_piece_words = d->sum_piece_words;
    if (n_gene_refs) *n_gene_refs = d->n_gene_refs;
    return HGX_OK;
}

extern "C" int hgx_gate_create(hgx_gate **g) { ARGCHK(g); *g = new hgx_gate(); return HGX_OK; }
extern "C" int hgx_gate_destroy(hgx_gate *g) { delete g; return HGX_OK; }

// ---- the per-locus body ---------------------------------------------------------------------------------------------
namespace {

struct GeneSide {                    // what the gene-level side hands back
    hgx_classes *gcl = nullptr;
    std::vector<int32_t> counted;
    std::vector<int64_t> cnt;
    int rc = HGX_OK;
    char err[512] = "";
};

int gene_rank(hgx_classes *gcl, int32_t A, int32_t a_pad, hipStream_t st, GeneSide &g);

// Gene_counts (core:1187-1190) and their print order (core:1650-1651): dict insertion order of Gene_counts = (first pair
// that counted the allele, Gene_names order), then the reference's stable descending sort on the count.
int gene_side(const hgx_index *ix, const hgx_dbatch *db, const uint64_t *compat, uint64_t *gene_bits, uint64_t *gene_hash,
              bool rows_ready, const hgx_type_opts *opts, hipStream_t st, GeneSide &g) {
    int32_t A = 0, a_pad = 0;
    int rc = hgx_index_dims(ix, &A, &a_pad, nullptr, nullptr);
    if (rc) return rc;
#ifdef HGX_LAB
    if (!rows_ready && hgx_test_switch("fused")) {
        // OPT-IN (measured slower): rows claimed / verified against their class' representative by the wavefront
        // that computes them -- no row per pair in memory, no insert pass, no verify pass (hgx_pair_classes_dedup); a key
        // collision falls through to the two-call form
        rc = hgx_pair_classes_dedup_ev(&g.gcl, ix, compat, db->d_pair_off, db->d_pair_ref, db->n_pairs, HGX_LEVEL_GENE, gene_bits, st,
                                       opts->ev_pairs_begin, opts->ev_pairs_end);
        if (rc == HGX_OK) return gene_rank(g.gcl, A, a_pad, st, g);
        if (rc != HGX_ECOLLISION) return rc;
    }
#endif
    if (!rows_ready) {
        if (opts->ev_pairs_begin) HIPCHK(hipEventRecord((hipEvent_t)opts->ev_pairs_begin, st));
        rc = hgx_pair_classes(ix, compat, db->d_pair_off, db->d_pair_ref, db->n_pairs, nullptr, gene_bits, nullptr, gene_hash, st);
        if (rc) return rc;
        if (opts->ev_pairs_end) HIPCHK(hipEventRecord((hipEvent_t)opts->ev_pairs_end, st));
    }
    rc = hgx_dedup_classes(&g.gcl, gene_bits, gene_hash, nullptr, db->n_pairs, a_pad, nullptr, st);
    if (rc) return rc;
    return gene_rank(g.gcl, A, a_pad, st, g);
}

int gene_rank(hgx_classes *gcl, int32_t A, int32_t a_pad, hipStream_t st, GeneSide &g) {
    int rc;
    std::vector<int64_t> cnt((size_t)a_pad);
    std::vector<int32_t> first((size_t)a_pad);
    rc = hgx_allele_counts_on(gcl, cnt.data(), first.data(), st);
    if (rc) return rc;
    int32_t C = 0;
    hgx_classes_dims(gcl, &C, nullptr);
    std::vector<int64_t> fr((size_t)std::max(C, 1));                     // first pair of every class
    rc = hgx_classes_to_host(gcl, nullptr, nullptr, fr.data());
    if (rc) return rc;
    g.cnt.assign(cnt.begin(), cnt.begin() + A);
    for (int32_t a = 0; a < A; ++a) if (cnt[a] > 0) g.counted.push_back(a);
    std::sort(g.counted.begin(), g.counted.end(), [&](int32_t a, int32_t b) {
        if (cnt[a] != cnt[b]) return cnt[a] > cnt[b];
        const int64_t ia = fr[first[a]], ib = fr[first[b]];
        if (ia != ib) return ia < ib;
        return a < b;
    });
    return HGX_OK;
}

int run_em(hgx_classes *cl, const hgx_locus *loc, int32_t remove_low, const int32_t *lengths, hipStream_t st, hgx_typing *t) {
    const int32_t A = loc->A;
    std::vector<double> prob((size_t)A);
    std::vector<int32_t> first((size_t)A);
    int32_t n_iter = 0, C = 0;
    const double t0 = now_s();
    int rc = hgx_classes_set_allele_rank(cl, loc->name_rank.data(), A);    // small problems then sum in the reference's own order
    if (!rc) rc = hgx_em_ordered(cl, A, remove_low, lengths, prob.data(), first.data(), &n_iter, st);
    t->t_em += now_s() - t0;
    if (rc) return rc;
    hgx_classes_dims(cl, &C, nullptr);
    EmOut o;
    o.exact = hgx_em_last_exact() != 0;
    o.n_classes = C; o.n_iter = n_iter; o.remove_low = remove_low ? 1 : 0; o.use_length = lengths ? 1 : 0;
    std::vector<int32_t> order((size_t)A);
    if (o.exact && hgx_em_last_order(order.data(), A)) {
        // the returned dict's own insertion order (its positions are distinct): the sort key instead of (first class, name order)
        for (int32_t a = 0; a < A; ++a) first[a] = prob[a] >= 0.0 ? order[a] : -1;
    }
    sorted_result(prob, first, loc->name_rank.data(), A, o);
    t->em.push_back(std::move(o));
    return HGX_OK;
}

// exon_alleles (core:1739-1749): the members of the exon groups of the leading representatives of EM #1's result
bool exon_alleles_of(const EmOut &e1, const hgx_locus *loc, std::vector<uint8_t> &in_exon, double &psum) {
    in_exon.assign((size_t)loc->A, 0);
    psum = 0.0;
    bool any = false;
    for (size_t i = 0; i < e1.allele.size(); ++i) {
        const int32_t a = e1.allele[i];
        const double p = e1.prob[i];
        if (i >= 10 && p < 0.03) break;
        const int32_t g0 = loc->grp_off[a], g1 = loc->grp_off[a + 1];       // members of a's exon group (precomputed per locus)
        if (g1 - g0 <= 1) continue;
        psum += p;
        for (int32_t k = g0; k < g1; ++k) { in_exon[loc->grp_member[k]] = 1; any = true; }
    }
    return any;
}
// Gene_combined_prob (core:1771-1782): exon-level survivors outside exon_alleles, then EM #2's result scaled by exon_prob_sum
void combine_levels(hgx_typing *t, EmOut &&e2, const std::vector<uint8_t> &in_exon, double psum) {
    const EmOut &e1 = t->em[0];
    EmOut comb;                                                   // dict order: exon-level survivors, then EM #2's
    for (size_t i = 0; i < e1.allele.size(); ++i)
        if (!in_exon[e1.allele[i]]) { comb.allele.push_back(e1.allele[i]); comb.prob.push_back(e1.prob[i]); }
    for (size_t i = 0; i < e2.allele.size(); ++i) { comb.allele.push_back(e2.allele[i]); comb.prob.push_back(e2.prob[i] * psum); }
    stable_desc(comb.allele, comb.prob, e1.exact && e2.exact);      // products p2 * psum in the reference's own order
    t->em.push_back(std::move(e2));
    t->gene_prob.allele = comb.allele;
    t->gene_prob.prob = comb.prob;
}

// EM #1 on the exon-level classes, exon_alleles, hand-off and EM #2 on the gene classes, combination (core:1732-1782).
// `gene_ready` delivers the gene-level class set (and the counts in `t`) when the exon-level EM is done -- the gene side may
// still be running beside it until then.
template <class GeneReady>
int finish_hla(hgx_typing *t, const hgx_locus *loc, hgx_classes *ecl, const hgx_type_opts *opts, hipStream_t em_stream, hipStream_t stream,
               GeneReady gene_ready) {
    const int32_t A = loc->A;
    const int w64 = loc->a_pad / 64;
    int rc;
    const bool prof = getenv("HGX_TYPE_PROFILE") != nullptr;
    const double tp0 = now_s();
    rc = run_em(ecl, loc, opts->remove_low, nullptr, em_stream, t);                          // core:1732-1737
    if (rc) return rc;
    const double tp1 = now_s();
    hgx_classes *gcl = nullptr;
    rc = gene_ready(&gcl);
    if (rc) return rc;
    const double tp2 = now_s();
    const EmOut &e1 = t->em[0];
    std::vector<uint8_t> in_exon;
    double psum = 0.0;
    const bool any = exon_alleles_of(e1, loc, in_exon, psum);
    t->gene_prob = e1;
    if (any) {                                                                               // core:1752-1782
        std::vector<uint64_t> mask((size_t)w64, 0);
        for (int32_t a = 0; a < A; ++a) if (in_exon[a]) mask[a >> 6] |= 1ull << (a & 63);
        std::vector<double> prob2((size_t)A);
        std::vector<int32_t> first2((size_t)A);
        int32_t it2 = 0, ncls2 = 0;
        const double t0 = now_s();
        rc = hgx_classes_set_allele_rank(gcl, loc->name_rank.data(), A);
        // Gene_cmpt2 (gene classes filtered to exon_alleles, merged) and EM #2 in one call
        if (!rc) rc = hgx_em_masked(gcl, mask.data(), A, 1, loc->allele_len.data(), prob2.data(), first2.data(), &it2, &ncls2, stream);
        t->t_em += now_s() - t0;
        if (rc) return rc;
        if (prof) fprintf(stderr, "[finish_hla] EM#1 %.1f us | wait gene %.1f | exon_alleles+mask %.1f | EM#2 call %.1f\n", (tp1 - tp0) * 1e6,
                          (tp2 - tp1) * 1e6, (t0 - tp2) * 1e6, (now_s() - t0) * 1e6);
        EmOut e2;
        e2.exact = hgx_em_last_exact() != 0;
        e2.n_classes = ncls2; e2.n_iter = it2; e2.remove_low = 1; e2.use_length = 1;
        sorted_result(prob2, first2, loc->name_rank.data(), A, e2);
        combine_levels(t, std::move(e2), in_exon, psum);
    }
    return HGX_OK;
}

// non-HLA bases (core:1784-1789): the EM on the gene classes, no pruning, no lengths; a single class is the reference's quirk Q3
int finish_other(hgx_typing *t, const hgx_locus *loc, hgx_classes *gcl, hipStream_t stream) {
    int32_t C = 0;
    hgx_classes_dims(gcl, &C, nullptr);
    if (C == 1) {
        hgx_set_error("'dict_keys' object is not subscriptable (reference quirk Q3, typing_core.py:1787)");
        return HGX_ETYPE;
    }
    if (C > 1) {
        const int rc = run_em(gcl, loc, 0, nullptr, stream, t);
        if (rc) return rc;
        t->gene_prob = t->em[0];
    }
    return HGX_OK;
}

int type_impl(hgx_typing *t, const hgx_locus *loc, const hgx_index *ix, const hgx_dbatch *db, const hgx_type_opts *opts,
              hipStream_t stream, StreamSet &ss, GateHold &gate) {
    int32_t A = 0, a_pad = 0;
    int rc = hgx_index_dims(ix, &A, &a_pad, nullptr, nullptr);
    if (rc) return rc;
    ARGCHK(A == loc->A && a_pad == loc->a_pad);
    ARGCHK((int32_t)loc->name_rank.size() == A && (int32_t)loc->allele_len.size() == A);
    const int w64 = a_pad / 64;
    const bool hla = loc->base_kind == HGX_BASE_HLA;
    const int32_t n_pairs = db->n_pairs;
    const bool by_list = hla && !opts->per_pair_exon;
    bool overlap = opts->overlap < 0 ? stream == nullptr : opts->overlap != 0;
    overlap = overlap && hla && n_pairs >= 4096;

    // Order of declaration = reverse order of clean-up on ANY way out: the gene-side thread is joined first, then every stream
    // that may still have kernels queued is drained, then handles and buffers go back to the pool.
    DevBuf b_compat, b_gbits, b_ghash, b_ebits, b_ehash;
    GeneSide gs;
    hgx_classes *ecl = nullptr;
    hgx_groups *groups = nullptr;
    struct Handles {
        GeneSide &g; hgx_classes *&ecl; hgx_groups *&groups;
        ~Handles() { hgx_groups_destroy(groups); hgx_classes_destroy(ecl); hgx_classes_destroy(g.gcl); }
    } handles{gs, ecl, groups};
    struct Drain {
        hipStream_t a, b, c;
        ~Drain() { (void)hipStreamSynchronize(a); if (b) (void)hipStreamSynchronize(b); if (c) (void)hipStreamSynchronize(c); }
    } drain{stream, overlap ? ss.em : nullptr, overlap ? ss.gene : nullptr};
    std::thread worker;
    struct Join {
        std::thread &w;
        ~Join() { if (w.joinable()) w.join(); }
    } join{worker};

    ALLOC(b_compat, (size_t)std::max(db->n_pieces, 1) * w64 * 8);
    ALLOC(b_gbits, (size_t)std::max(n_pairs, 1) * w64 * 8);
    ALLOC(b_ghash, (size_t)std::max(n_pairs, 1) * 8);
    if (hla) {                       // exon-level rows: per pair (HGX_NO_SIG) or scratch for the rows per distinct ref list
        ALLOC(b_ebits, (size_t)std::max(n_pairs, 1) * w64 * 8);
        ALLOC(b_ehash, (size_t)std::max(n_pairs, 1) * 8);
    }
    uint64_t *compat = b_compat.as<uint64_t>();

    if (opts->ev_compat_begin) HIPCHK(hipEventRecord((hipEvent_t)opts->ev_compat_begin, stream));
    if (by_list || !hla) {
        rc = hgx_piece_compat(ix, db->d_pieces, db->d_masks, db->n_pieces, compat, stream);
        if (rc) return rc;
        if (opts->ev_compat_end) HIPCHK(hipEventRecord((hipEvent_t)opts->ev_compat_end, stream));
    }
    bool rows_ready = false;
    if (!by_list) {                  // per-pair rows of every level right away
        if (hla) {
            rc = hgx_score_pairs(ix, db->d_pieces, db->d_masks, db->n_pieces, db->d_pair_off, db->d_pair_ref, n_pairs, compat,
                                 b_ebits.as<uint64_t>(), b_gbits.as<uint64_t>(), b_ehash.as<uint64_t>(), b_ghash.as<uint64_t>(), stream);
            if (opts->ev_compat_end) HIPCHK(hipEventRecord((hipEvent_t)opts->ev_compat_end, stream));
        } else {
            rc = hgx_pair_classes(ix, compat, db->d_pair_off, db->d_pair_ref, n_pairs, nullptr, b_gbits.as<uint64_t>(), nullptr,
                                  b_ghash.as<uint64_t>(), stream);
        }
        if (rc) return rc;
        rows_ready = true;
    }

    hipStream_t em_stream = stream, gene_stream = stream;
    const bool prof = getenv("HGX_TYPE_PROFILE") != nullptr;
    const double tq0 = now_s();
    double tq1 = tq0;

    if (overlap) {
        em_stream = ss.em;
        gene_stream = ss.gene;
        // scoring is complete before either side reads its output: a device-side dependency, the host keeps running ahead
        HIPCHK(hipEventRecord(ss.fork, stream));
        HIPCHK(hipStreamWaitEvent(gene_stream, ss.fork, 0));
        if (by_list) {
            // grouping the pairs by exon-level ref list does not read the piece bitsets: queued on the EM stream right away
            // it runs BESIDE hgx_piece_compat; the stream is ordered behind the scoring only further down
            rc = hgx_group_pairs(&groups, db->d_pair_off, db->d_pair_ref, n_pairs, HGX_LEVEL_EXON, em_stream);
            if (rc) return rc;
        }
        int dev = 0;
        HIPCHK(hipGetDevice(&dev));
        worker = std::thread([&, dev] {
            if (hipSetDevice(dev) != hipSuccess) { gs.rc = HGX_EHIP; snprintf(gs.err, sizeof(gs.err), "hipSetDevice failed on the gene-side thread"); return; }
            gs.rc = gene_side(ix, db, compat, b_gbits.as<uint64_t>(), b_ghash.as<uint64_t>(), rows_ready, opts, gene_stream, gs);
            if (gs.rc) snprintf(gs.err, sizeof(gs.err), "%s", hgx_last_error());
        });
        if (groups) {
            int64_t ng = 0;
            rc = hgx_groups_dims(groups, &ng, nullptr);          // host wait for the grouping alone
            if (rc) return rc;
        }
        tq1 = now_s();
        HIPCHK(hipStreamWaitEvent(em_stream, ss.fork, 0));
    } else {
        gs.rc = gene_side(ix, db, compat, b_gbits.as<uint64_t>(), b_ghash.as<uint64_t>(), rows_ready, opts, stream, gs);
        if (gs.rc) return gs.rc;
    }
    auto finish_gene = [&]() -> int {
        if (worker.joinable()) worker.join();
        if (gs.rc) { hgx_set_error("%s", gs.err[0] ? gs.err : "gene-level side failed"); return gs.rc; }
        t->counted = gs.counted;
        t->cnt = gs.cnt;
        return HGX_OK;
    };

    if (hla) {
        if (by_list) {
            if (groups) rc = hgx_level_classes_grouped(&ecl, ix, compat, db->d_pair_off, db->d_pair_ref, groups, b_ebits.as<uint64_t>(),
                                                       b_ehash.as<uint64_t>(), em_stream);
            else rc = hgx_level_classes(&ecl, ix, compat, db->d_pair_off, db->d_pair_ref, n_pairs, HGX_LEVEL_EXON, b_ebits.as<uint64_t>(),
                                        b_ehash.as<uint64_t>(), em_stream);
        } else {
            rc = hgx_dedup_classes(&ecl, b_ebits.as<uint64_t>(), b_ehash.as<uint64_t>(), nullptr, n_pairs, a_pad, nullptr, em_stream);
        }
        if (rc) return rc;
        if (prof) fprintf(stderr, "[type_impl] queue + wait for the grouping %.1f us | exon-level classes (rows, dedup) %.1f us\n", (tq1 - tq0) * 1e6,
                          (now_s() - tq1) * 1e6);
        gate.release();              // several samples in flight: the bandwidth-bound front of this one is through
        rc = finish_hla(t, loc, ecl, opts, em_stream, stream, [&](hgx_classes **g) { const int r = finish_gene(); *g = gs.gcl; return r; });
        if (rc) return rc;
    } else {
        rc = finish_gene();
        if (rc) return rc;
        gate.release();
        rc = finish_other(t, loc, gs.gcl, stream);
        if (rc) return rc;
    }
    if (opts->keep_classes) { t->gene_cl = gs.gcl; gs.gcl = nullptr; t->exon_cl = ecl; ecl = nullptr; }
    return HGX_OK;
}

}   // namespace

extern "C" int hgx_typing_destroy(hgx_typing *t) { delete t; return HGX_OK; }

// diagnostic (tools/stream_probe.py): n fresh streams (priority per stream: 1 = highest, 0 = lowest), the time two 150 us spin kernels
// take back to back on every pair of them -- ~150 us side by side, ~300 one behind the other; us[n][n], the diagonal = one kernel alone
extern "C" int hgx_stream_probe_matrix(int32_t n, const int32_t *high_prio, double *us) {
    ARGCHK(n > 0 && n <= 64 && us);
    int least = 0, greatest = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    std::vector<hipStream_t> st((size_t)n, nullptr);
    struct Free { std::vector<hipStream_t> &v; ~Free() { for (hipStream_t x : v) if (x) (void)hipStreamDestroy(x); } } fr{st};
    for (int i = 0; i < n; ++i) HIPCHK(hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, high_prio && high_prio[i] ? greatest : least));
    for (int rep = 0; rep < 2; ++rep)
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                (void)hipStreamSynchronize(st[i]);
                (void)hipStreamSynchronize(st[j]);
                const auto t0 = std::chrono::steady_clock::now();
                hipLaunchKernelGGL(k_ss_spin, dim3(1), dim3(64), 0, st[i], SS_SPIN_TICKS, (long long *)nullptr);
                if (i != j) hipLaunchKernelGGL(k_ss_spin, dim3(1), dim3(64), 0, st[j], SS_SPIN_TICKS, (long long *)nullptr);
                (void)hipStreamSynchronize(st[i]);
                (void)hipStreamSynchronize(st[j]);
                const double t = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                us[(size_t)i * n + j] = rep == 0 ? t : std::min(us[(size_t)i * n + j], t);
            }
    return HGX_OK;
}

// diagnostic: do kernels on these two streams run one behind the other (1) or side by side (0)?  (tools/stream_conflicts.py)
extern "C" int hgx_stream_probe_pair(void *a, void *b, int32_t *same_queue) {
    ARGCHK(same_queue);
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    DevStreams tmp;
    *same_queue = ss_same_queue(tmp, (hipStream_t)a, (hipStream_t)b) ? 1 : 0;
    return HGX_OK;
}
// diagnostic: a chain of 16 one-wavefront 10 us kernels on `light`, alone (us[0]) and (us[1]) beside `other` running mode 0: ONE
// launch of 131 072 tiny workgroups (dispatch pressure), mode 1: the same chain, launches interleaved (what the placement measures)
__global__ void k_ss_tiny(unsigned *sink) { if (threadIdx.x == 1u << 20) *sink = 1; }
extern "C" int hgx_stream_probe_chain(void *light, void *other, int32_t mode, double *us) {
    ARGCHK(us && light && other && (mode == 0 || mode == 1));
    long long *stamps = ss_stamps();
    if (!stamps) { hgx_set_error("pinned allocation failed"); return HGX_ENOMEM; }
    hipStream_t l = (hipStream_t)light, h = (hipStream_t)other;
    for (int with = 0; with < 2; ++with) {
        double best = 1e30;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipStreamSynchronize(l);
            (void)hipStreamSynchronize(h);
            if (with && mode == 0) hipLaunchKernelGGL(k_ss_tiny, dim3(131072), dim3(64), 0, h, (unsigned *)nullptr);
            for (int k = 0; k < 16; ++k) {
                hipLaunchKernelGGL(k_ss_spin, dim3(1), dim3(64), 0, l, 1000LL, stamps + 2 * k);
                if (with && mode == 1) hipLaunchKernelGGL(k_ss_spin, dim3(1), dim3(64), 0, h, 1000LL, stamps + 32 + 2 * (k & 7));
            }
            (void)hipStreamSynchronize(l);
            (void)hipStreamSynchronize(h);
            best = std::min(best, (double)(stamps[31] - stamps[0]) / 100.0);
        }
        us[with] = best;
    }
    return HGX_OK;
}
// ... and the streams of the sets that are free at the moment (em, gene per set), in hand-out order
extern "C" int hgx_stream_sets_streams(void **streams, int32_t cap, int32_t *n_sets) {
    ARGCHK(streams && n_sets);
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> g(g_ss_mu);
    DevStreams &D = g_ss[dev];
    *n_sets = 0;
    for (size_t i = 0; i < D.free.size() && (int32_t)(2 * i + 1) < cap; ++i) { streams[2 * i] = D.free[i].em; streams[2 * i + 1] = D.free[i].gene; (*n_sets)++; }
    return HGX_OK;
}

// what the stream placement of the current device looks like (bench.py prints it; tests assert on it)
extern "C" int hgx_stream_sets_info(int32_t *n_sets, int32_t *n_classes, int32_t *n_probes, double *probe_ms, int32_t *classes, int32_t cap) {
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> g(g_ss_mu);
    DevStreams &D = g_ss[dev];
    if (n_sets) *n_sets = D.n_sets;
    if (n_classes) *n_classes = (int32_t)D.reps.size();
    if (n_probes) *n_probes = D.probes;
    if (probe_ms) *probe_ms = D.probe_ms;
    if (classes)                                    // (em class, gene class) of the sets that are free at the moment, in hand-out order
        for (size_t i = 0; i < D.free.size() && (int32_t)(2 * i + 1) < cap; ++i) { classes[2 * i] = D.free[i].cls_em; classes[2 * i + 1] = D.free[i].cls_gene; }
    return HGX_OK;
}

extern "C" int hgx_type_dbatch(hgx_typing **out, const hgx_locus *loc, const hgx_index *ix, const hgx_dbatch *db,
                               const hgx_type_opts *opts, void *stream) {
    ARGCHK(out && loc && ix && db && opts);
    *out = nullptr;
    hgx_typing *t = new hgx_typing();
    t->n_reads = db->n_reads; t->n_pairs = db->n_pairs; t->n_pieces = db->n_pieces; t->n_refs = db->n_refs; t->n_alleles = loc->A;
    if (db->n_reads <= 0) { *out = t; return HGX_OK; }                                           // core:1589-1590
    GateHold gate(opts->gate);
    EmFastScope em_mode(em_mode_one(opts->em_fast));
    StreamSet ss;
    int rc = acquire_streams(ss);
    if (!rc) rc = type_impl(t, loc, ix, db, opts, (hipStream_t)stream, ss, gate);
    release_streams(ss);
    if (rc) { delete t; return rc; }
    *out = t;
    return HGX_OK;
}

extern "C" int hgx_type_classes(hgx_typing **out, const hgx_locus *loc, hgx_classes *exon_cl, hgx_classes *gene_cl, int32_t n_reads,
                                int32_t n_pairs, const hgx_type_opts *opts, void *stream) {
    ARGCHK(out && loc && gene_cl && opts);
    *out = nullptr;
    const bool hla = loc->base_kind == HGX_BASE_HLA;
    ARGCHK(!hla || exon_cl);
    ARGCHK(gene_cl->a_pad == loc->a_pad && (!exon_cl || exon_cl->a_pad == loc->a_pad));
    ARGCHK((int32_t)loc->name_rank.size() == loc->A && (int32_t)loc->allele_len.size() == loc->A);
    hgx_typing *t = new hgx_typing();
    t->n_reads = n_reads; t->n_pairs = n_pairs; t->n_alleles = loc->A;
    if (n_reads <= 0) { *out = t; return HGX_OK; }
    hipStream_t st = (hipStream_t)stream;
    EmFastScope em_mode(em_mode_one(opts->em_fast));
    GeneSide gs;
    int rc = gene_rank(gene_cl, loc->A, loc->a_pad, st, gs);
    if (!rc) {
        t->counted = gs.counted;
        t->cnt = gs.cnt;
        if (hla) rc = finish_hla(t, loc, exon_cl, opts, st, st, [&](hgx_classes **g) { *g = gene_cl; return HGX_OK; });
        else rc = finish_other(t, loc, gene_cl, st);
    }
    if (rc) { delete t; return rc; }
    *out = t;
    return HGX_OK;
}

extern "C" int hgx_type_batch(hgx_typing **out, const hgx_locus *loc, const hgx_index *ix, const hgx_batch *batch,
                              const hgx_type_opts *opts, void *stream) {
    ARGCHK(out && batch);
    *out = nullptr;
    hgx_dbatch *db = nullptr;
    int rc = hgx_dbatch_create(&db, batch, stream);
    if (rc) return rc;
    rc = hgx_type_dbatch(out, loc, ix, db, opts, stream);
    hgx_dbatch_destroy(db);
    return rc;
}

extern "C" int hgx_type_file(hgx_typing **out, const hgx_locus *loc, const hgx_index *ix, const char *path, const char *regions,
                             const hgx_parse_opts *popts, const hgx_type_opts *opts, void *stream) {
    ARGCHK(out && loc && ix && path && popts && opts);
    *out = nullptr;
    hgx_dbatch *db = nullptr;
    const bool prof = getenv("HGX_PARSE_PROFILE") != nullptr;
    const double t0 = now_s();
    // host stages (read, tokenise, filters, key grouping) + device stages (pileup, decode, piece table, pair protocol): the batch is
    // born in HBM; inputs the kernels decline are finished on the host and uploaded (hgx_front.hip)
    int rc = hgx_parse_alignment_file_dev(&db, loc, path, regions, popts, stream);
    if (rc) return rc;
    const double t1 = now_s();
    rc = hgx_type_dbatch(out, loc, ix, db, opts, stream);
    const double t2 = now_s();
    hgx_dbatch_destroy(db);
    if (prof) fprintf(stderr, "[hgx_type_file] front end %.1f ms (incl. releasing the reader's buffers), GPU path + result %.1f ms, batch destroy %.1f ms\n",
                      (t1 - t0) * 1e3, (t2 - t1) * 1e3, (now_s() - t2) * 1e3);
    return rc;
}

extern "C" int hgx_typing_dims(const hgx_typing *t, int32_t *n_reads, int32_t *n_pairs, int32_t *n_pieces, int64_t *n_refs,
                               int32_t *n_counted, int32_t *n_em, int32_t *n_gene_prob, double *em_seconds) {
    ARGCHK(t);
    if (n_reads) *n_reads = t->n_reads;
    if (n_pairs) *n_pairs = t->n_pairs;
    if (n_pieces) *n_pieces = t->n_pieces;
    if (n_refs) *n_refs = t->n_refs;
    if (n_counted) { t->ensure_ranked(); *n_counted = (int32_t)t->counted.size(); }
    if (n_em) *n_em = (int32_t)t->em.size();
    if (n_gene_prob) *n_gene_prob = (int32_t)t->gene_prob.allele.size();
    if (em_seconds) *em_seconds = t->t_em;
    return HGX_OK;
}

extern "C" int hgx_typing_counts(const hgx_typing *t, int32_t *ranked_allele, int64_t *count_per_allele) {
    ARGCHK(t);
    if (ranked_allele) t->ensure_ranked();
    if (ranked_allele && !t->counted.empty()) memcpy(ranked_allele, t->counted.data(), t->counted.size() * 4);
    if (count_per_allele && !t->cnt.empty()) memcpy(count_per_allele, t->cnt.data(), t->cnt.size() * 8);
    return HGX_OK;
}

static int em_out(const EmOut &e, int32_t *n_classes, int32_t *n_iter, int32_t *remove_low, int32_t *use_length, int32_t *n_result,
                  int32_t *allele, double *prob) {
    if (n_classes) *n_classes = e.n_classes;
    if (n_iter) *n_iter = e.n_iter;
    if (remove_low) *remove_low = e.remove_low;
    if (use_length) *use_length = e.use_length;
    if (n_result) *n_result = (int32_t)e.allele.size();
    if (allele && !e.allele.empty()) memcpy(allele, e.allele.data(), e.allele.size() * 4);
    if (prob && !e.prob.empty()) memcpy(prob, e.prob.data(), e.prob.size() * 8);
    return HGX_OK;
}

extern "C" int hgx_typing_em(const hgx_typing *t, int32_t k, int32_t *n_classes, int32_t *n_iter, int32_t *remove_low,
                             int32_t *use_length, int32_t *n_result, int32_t *allele, double *prob) {
    ARGCHK(t && k >= 0 && k < (int32_t)t->em.size());
    return em_out(t->em[k], n_classes, n_iter, remove_low, use_length, n_result, allele, prob);
}

extern "C" int hgx_typing_gene_prob(const hgx_typing *t, int32_t *allele, double *prob) {
    ARGCHK(t);
    return em_out(t->gene_prob, nullptr, nullptr, nullptr, nullptr, nullptr, allele, prob);
}

extern "C" int hgx_typing_classes(const hgx_typing *t, int32_t level, const hgx_classes **out) {
    ARGCHK(t && out && (level == HGX_LEVEL_EXON || level == HGX_LEVEL_GENE));
    *out = level == HGX_LEVEL_EXON ? t->exon_cl : t->gene_cl;
    return HGX_OK;
}

// =====================================================================================================================
// Many tasks of ONE locus behind one launch chain (hgx_type_many).  The reference's unit of scale is many samples x loci
// (/root/reference/hisatgenotype:613-665 Pool.apply_async(genotyping_locus ...), typing_core.py:370 locus loop): with one launch
// chain per task a 384-task panel is 43 000 launches and the GPU idles between them.  Here the tasks' piece batches are merged
// (hgx_many_create), scored and de-duplicated by the same kernels as one task with the dedup keeping the tasks apart, and the
// per-task remainder -- Gene_counts, EM #1, the hand-off, EM #2 -- runs with a task dimension (hgx_many.hip, hgx_emx.hip).
// Results are the per-task path's: same class tables, same counts and orders, the same doubles out of the EMs (both run in
// the reference's own order of operations).
// =====================================================================================================================
int hgx_many_class_tasks(const hgx_classes *cl, const uint32_t *pair_seg, int32_t n_tasks, int32_t *per_task_dev, hipStream_t st);
int hgx_many_counts(const hgx_classes *gcl, const int32_t *cls_off_dev, const int32_t *pair_base_dev, int32_t n_tasks,
                    int64_t *out_cnt_dev, int32_t *out_first_pair_dev, hipStream_t st);
int hgx_many_fill_seg(const int32_t *pair_base_dev, int32_t n_tasks, uint32_t *seg_dev, hipStream_t st);

struct hgx_many {
    int32_t n_tasks = 0, A = 0, a_pad = 0;
    hgx_dbatch *db = nullptr;                        // the merged batch
    std::vector<int32_t> pair_base, n_reads, n_pieces;
    std::vector<int64_t> n_refs;
    uint32_t *d_pair_seg = nullptr;                  // [n_pairs] task of every pair
    int32_t *d_pair_base = nullptr;                  // [n_tasks + 1]
    int32_t *d_rank = nullptr;                       // [a_pad] name order of the alleles
    double *d_len = nullptr;                         // [a_pad] allele lengths
    void *h_pinned = nullptr;                        // staging of the tasks' Gene_counts
    size_t h_pinned_bytes = 0;
    std::atomic<int> in_use{0};                      // the staging block serves one call at a time
};

extern "C" int hgx_many_destroy(hgx_many *m) {
    if (!m) return HGX_OK;
    hgx_dbatch_destroy(m->db);
    hgx_pool_free(m->d_pair_seg); hgx_pool_free(m->d_pair_base); hgx_pool_free(m->d_rank); hgx_pool_free(m->d_len);
    if (m->h_pinned) (void)hipHostFree(m->h_pinned);
    delete m;
    return HGX_OK;
}

namespace {
// the tables a many-task batch needs besides the merged batch itself: task of every pair, the alleles' name order and lengths
int many_finish(hgx_many *m, const hgx_locus *loc, hipStream_t st) {
    const int32_t n_tasks = m->n_tasks, n_pairs = m->db->n_pairs;
    m->d_pair_seg = (uint32_t *)hgx_pool_alloc((size_t)std::max(n_pairs, 1) * 4);
    m->d_pair_base = (int32_t *)hgx_pool_alloc((size_t)(n_tasks + 1) * 4);
    m->d_rank = (int32_t *)hgx_pool_alloc((size_t)loc->a_pad * 4);
    m->d_len = (double *)hgx_pool_alloc((size_t)loc->a_pad * 8);
    if (!m->d_pair_seg || !m->d_pair_base || !m->d_rank || !m->d_len) { hgx_set_error("device allocation failed"); return HGX_ENOMEM; }
    std::vector<int32_t> rank((size_t)loc->a_pad, 0);
    std::vector<double> len((size_t)loc->a_pad, 1.0);
    for (int32_t a = 0; a < loc->A; ++a) { rank[a] = loc->name_rank[a]; len[a] = (double)loc->allele_len[a]; }
    for (int32_t a = loc->A; a < loc->a_pad; ++a) rank[a] = a;       // (never read: padding alleles occur in no class)
    bool ok = hipMemcpyAsync(m->d_pair_base, m->pair_base.data(), (size_t)(n_tasks + 1) * 4, hipMemcpyHostToDevice, st) == hipSuccess;
    ok = ok && hipMemcpyAsync(m->d_rank, rank.data(), rank.size() * 4, hipMemcpyHostToDevice, st) == hipSuccess;
    ok = ok && hipMemcpyAsync(m->d_len, len.data(), len.size() * 8, hipMemcpyHostToDevice, st) == hipSuccess;
    ok = ok && hgx_many_fill_seg(m->d_pair_base, n_tasks, m->d_pair_seg, st) == HGX_OK;
    ok = ok && hipStreamSynchronize(st) == hipSuccess;
    if (!ok) { hgx_set_error("upload of the merged batch tables failed"); return HGX_EHIP; }
    return HGX_OK;
}
}   // namespace

extern "C" int hgx_many_create(hgx_many **out, const hgx_locus *loc, const hgx_batch *const *batches, int32_t n_tasks, void *stream) {
    ARGCHK(out && loc && n_tasks >= 0 && (n_tasks == 0 || batches));
    *out = nullptr;
    ARGCHK((int32_t)loc->name_rank.size() == loc->A && (int32_t)loc->allele_len.size() == loc->A);
    hipStream_t st = (hipStream_t)stream;
    hgx_many *m = new hgx_many();
    m->n_tasks = n_tasks; m->A = loc->A; m->a_pad = loc->a_pad;
    m->pair_base.assign((size_t)n_tasks + 1, 0);
    for (int t = 0; t < n_tasks; ++t) {
        ARGCHK(batches[t]);
        m->n_reads.push_back(batches[t]->n_reads);
        m->n_pieces.push_back((int32_t)batches[t]->pieces.size());
        m->n_refs.push_back((int64_t)batches[t]->pair_ref.size());
    }
    hgx_batch *merged = nullptr;
    int rc = hgx_batch_merge(&merged, batches, n_tasks, m->pair_base.data());
    if (!rc) rc = hgx_dbatch_create(&m->db, merged, stream);
    hgx_batch_destroy(merged);
    if (!rc) rc = many_finish(m, loc, st);
    if (rc) { hgx_many_destroy(m); return rc; }
    *out = m;
    return HGX_OK;
}

namespace {
// alignment streams -> many-task batch: one pass of the device front end over all tasks, or -- where it declines (a record the
// reference would raise on, options it leaves to the host, too little work) -- the host front end task by task, then the merge
int many_from_streams(hgx_many **out, const hgx_locus *loc, const char *const *paths, const char *const *regions, const char *const *sams,
                      const size_t *sam_bytes, int32_t n_tasks, const hgx_parse_opts *opts, void *stream) {
    *out = nullptr;
    ARGCHK((int32_t)loc->name_rank.size() == loc->A && (int32_t)loc->allele_len.size() == loc->A);
    hipStream_t st = (hipStream_t)stream;
    hgx_dbatch *db = nullptr;
    hgx_front_totals tot;
    int declined = 0;
    int rc = n_tasks > 0 ? hgx_front_many_dev(&db, &tot, loc, paths, regions, sams, sam_bytes, n_tasks, opts, stream, &declined) : HGX_OK;
    if (rc) return rc;
    if (n_tasks > 0 && !declined && db) {
        hgx_many *m = new hgx_many();
        m->n_tasks = n_tasks; m->A = loc->A; m->a_pad = loc->a_pad;
        m->db = db;
        m->pair_base.assign((size_t)n_tasks + 1, 0);
        for (int t = 0; t < n_tasks; ++t) {
            m->pair_base[(size_t)t + 1] = m->pair_base[t] + (int32_t)tot.pairs[t];
            m->n_reads.push_back((int32_t)tot.reads[t]);
            m->n_pieces.push_back((int32_t)tot.pieces[t]);
            m->n_refs.push_back((int64_t)tot.refs[t]);
        }
        if (m->pair_base[(size_t)n_tasks] != db->n_pairs) {
            hgx_many_destroy(m);
            hgx_set_error("device front end: the tasks' pair counts do not add up to the batch");
            return HGX_EHIP;
        }
        rc = many_finish(m, loc, st);
        if (rc) { hgx_many_destroy(m); return rc; }
        *out = m;
        return HGX_OK;
    }
    // host front end, tasks side by side
    std::vector<hgx_batch *> bs((size_t)n_tasks, nullptr);
    std::vector<int> rcs((size_t)n_tasks, HGX_OK);
    std::vector<std::string> errs((size_t)n_tasks);
    const int n_threads = opts->n_threads > 0 ? opts->n_threads : hgx_default_threads();
    hgx_parse_opts po = *opts;
    po.n_threads = n_tasks > 0 ? std::max(1, n_threads / n_tasks) : 1;
    hgx_par_tasks(std::max(1, std::min(n_threads, n_tasks)), (size_t)n_tasks, [&](int, size_t t) {
        rcs[t] = paths ? hgx_parse_alignment_file(&bs[t], loc, paths[t], regions ? regions[t] : nullptr, &po)
                       : hgx_parse_sam(&bs[t], loc, sams[t], sam_bytes[t], &po);
        if (rcs[t]) errs[t] = hgx_last_error();
    });
    rc = HGX_OK;
    for (int t = 0; t < n_tasks && !rc; ++t)
        if (rcs[t]) { hgx_set_error("task %d: %s", t, errs[t].c_str()); rc = rcs[t]; }
    if (!rc) rc = hgx_many_create(out, loc, bs.data(), n_tasks, stream);
    for (hgx_batch *b : bs) hgx_batch_destroy(b);
    return rc;
}
}   // namespace

extern "C" int hgx_many_create_files(hgx_many **out, const hgx_locus *loc, const char *const *paths, const char *const *regions, int32_t n_tasks,
                                     const hgx_parse_opts *opts, void *stream) {
    ARGCHK(out && loc && opts && n_tasks >= 0 && (n_tasks == 0 || paths));
    for (int t = 0; t < n_tasks; ++t) ARGCHK(paths[t]);
    return many_from_streams(out, loc, paths, regions, nullptr, nullptr, n_tasks, opts, stream);
}

extern "C" int hgx_many_create_sams(hgx_many **out, const hgx_locus *loc, const char *const *sams, const size_t *n_bytes, int32_t n_tasks,
                                    const hgx_parse_opts *opts, void *stream) {
    ARGCHK(out && loc && opts && n_tasks >= 0 && (n_tasks == 0 || (sams && n_bytes)));
    for (int t = 0; t < n_tasks; ++t) ARGCHK(sams[t] || n_bytes[t] == 0);
    return many_from_streams(out, loc, nullptr, nullptr, sams, n_bytes, n_tasks, opts, stream);
}

// ONE task's batch that is already resident (hgx_parse_*_dev, hgx_alignment_parse_dev) as a many-task batch of one task: the loci of
// one sample go into hgx_type_many_loci this way -- the EMs of all of them in one launch.  Takes the batch over on success.
extern "C" int hgx_many_from_dbatch(hgx_many **out, const hgx_locus *loc, hgx_dbatch *db, void *stream) {
    ARGCHK(out && loc && db);
    *out = nullptr;
    ARGCHK((int32_t)loc->name_rank.size() == loc->A && (int32_t)loc->allele_len.size() == loc->A);
    hgx_many *m = new hgx_many();
    m->n_tasks = 1; m->A = loc->A; m->a_pad = loc->a_pad;
    m->db = db;
    m->pair_base = {0, db->n_pairs};
    m->n_reads.push_back(db->n_reads);
    m->n_pieces.push_back(db->n_pieces);
    m->n_refs.push_back(db->n_refs);
    const int rc = many_finish(m, loc, (hipStream_t)stream);
    if (rc) { m->db = nullptr; hgx_many_destroy(m); return rc; }          // (the caller keeps its batch)
    *out = m;
    return HGX_OK;
}

// the merged batch of a many-task batch (tests, tools: hgx_dbatch_to_host on it) and its per-task extents; arrays of n_tasks (+ 1
// for pair_base) entries or NULL
extern "C" int hgx_many_tasks(const hgx_many *m, const hgx_dbatch **db, int32_t *pair_base, int32_t *n_reads, int32_t *n_pieces, int64_t *n_refs) {
    ARGCHK(m);
    if (db) *db = m->db;
    for (int t = 0; t < m->n_tasks; ++t) {
        if (pair_base) pair_base[t] = m->pair_base[t];
        if (n_reads) n_reads[t] = m->n_reads[t];
        if (n_pieces) n_pieces[t] = m->n_pieces[t];
        if (n_refs) n_refs[t] = m->n_refs[t];
    }
    if (pair_base) pair_base[m->n_tasks] = m->pair_base[(size_t)m->n_tasks];
    return HGX_OK;
}

extern "C" int hgx_many_dims(const hgx_many *m, int32_t *n_tasks, int32_t *n_pieces, int32_t *n_pairs, int64_t *n_refs, int64_t *n_reads) {
    ARGCHK(m);
    if (n_tasks) *n_tasks = m->n_tasks;
    if (n_pieces) *n_pieces = m->db->n_pieces;
    if (n_pairs) *n_pairs = m->db->n_pairs;
    if (n_refs) *n_refs = m->db->n_refs;
    if (n_reads) { int64_t r = 0; for (int32_t x : m->n_reads) r += x; *n_reads = r; }
    return HGX_OK;
}

namespace {

// a task's classes inside the merged class table, as a class set of its own (nothing owned but what the EM builds lazily)
struct ClassesView {
    hgx_classes c;
    ClassesView(const hgx_classes *src, int32_t off, int32_t n) : c() {
        c.n_classes = n; c.a_pad = src->a_pad; c.w64 = src->w64; c.c64 = 0;
        c.d_bits = src->d_bits + (size_t)off * src->w64;
        c.d_count = src->d_count + off;
        c.d_first_row = src->d_first_row + off;
        c.d_bitsT = nullptr;
    }
    ~ClassesView() {
        hgx_pool_free(c.d_bitsT); hgx_pool_free(c.d_prow); hgx_pool_free(c.d_pcol); hgx_pool_free(c.d_act); hgx_pool_free(c.d_bitsC);
        hgx_pool_free(c.d_bitsTC); hgx_pool_free(c.d_wrow); hgx_pool_free(c.d_wcol); hgx_pool_free(c.d_setup0); hgx_pool_free(c.d_setup1);
        delete[] c.h_act;
        delete[] c.h_rank;
    }
};

// per-task class ranges of a merged class table: off[t] .. off[t + 1]
int class_offsets(const hgx_classes *cl, const hgx_many *m, int32_t *scratch_dev, std::vector<int32_t> &off, hipStream_t st) {
    const int n = m->n_tasks;
    off.assign((size_t)n + 1, 0);
    if (!cl || cl->n_classes == 0) return HGX_OK;
    int rc = hgx_many_class_tasks(cl, m->d_pair_seg, n, scratch_dev, st);
    if (rc) return rc;
    std::vector<int32_t> start((size_t)n);
    { int rc_ = hgx_d2h(start.data(), scratch_dev, (size_t)n * 4, st); if (rc_) return rc_; }
    { int rc_ = hgx_sync(st); if (rc_) return rc_; }
    // runs of classes, task after task: a task without classes starts where the next one does
    off[n] = cl->n_classes;
    int32_t last = -1;
    for (int t = n - 1; t >= 0; --t) {
        off[t] = start[t] >= 0 ? start[t] : off[t + 1];
        if (start[t] >= 0) { if (last >= 0 && start[t] > last) { hgx_set_error("class table of the merged batch is not ordered by task"); return HGX_EHIP; } last = start[t]; }
    }
    if (off[0] != 0) { hgx_set_error("class table of the merged batch does not start with the first task's classes"); return HGX_EHIP; }
    return HGX_OK;
}

// the result list of a batched EM from its records: dict insertion order (first class, then name order), then the reference's
// stable descending sort -- what sorted_result does from dense arrays
void em_out_from(const hgx_emx_job &J, const std::vector<hgx_emx_rec> &recs, const hgx_locus *loc, int32_t use_length, EmOut &o, bool by_dict_order) {
    o.exact = J.fast == 0;
    o.n_classes = J.n_classes; o.n_iter = J.n_iter; o.remove_low = J.remove_low ? 1 : 0; o.use_length = use_length;
    std::vector<hgx_emx_rec> r(recs.begin() + J.rec_off, recs.begin() + J.rec_off + J.n_rec);
    const int32_t *name_rank = loc->name_rank.data();
    std::sort(r.begin(), r.end(), [&](const hgx_emx_rec &x, const hgx_emx_rec &y) {
        if (by_dict_order && o.exact) return x.order < y.order;          // the returned dict's insertion order (distinct positions)
        if (x.first != y.first) return x.first < y.first;
        if (name_rank[x.allele] != name_rank[y.allele]) return name_rank[x.allele] < name_rank[y.allele];
        return x.allele < y.allele;
    });
    o.allele.resize(r.size());
    o.prob.resize(r.size());
    for (size_t k = 0; k < r.size(); ++k) { o.allele[k] = r[k].allele; o.prob[k] = r[k].prob; }
    stable_desc(o.allele, o.prob, o.exact);
}

}   // namespace

namespace {

// One locus' share of a hgx_type_many / hgx_type_many_loci call.  The phases of all loci are interleaved by the caller so that
// the EM jobs of EVERY locus go out in one launch (one workgroup per task: the launch is as wide as the panel, not as one locus).
struct ManyRun {
    const hgx_locus *loc = nullptr;
    const hgx_index *ix = nullptr;
    hgx_many *m = nullptr;
    const hgx_type_opts *opts = nullptr;
    hipStream_t st = nullptr;
    int32_t *rc_out = nullptr;
    int n = 0, A = 0, a_pad = 0, w64 = 0;
    bool hla = false, active = false;
    std::vector<hgx_typing *> res;
    DevBuf b_compat, b_gbits, b_ghash, b_ebits, b_ehash, b_pt, b_goff, b_cnt, b_fp, b_masks;
    hgx_classes *ecl = nullptr, *gcl = nullptr;
    hgx_groups *groups = nullptr;
    std::vector<int32_t> e_off, g_off;
    bool holds_many = false;
    int64_t *h_cnt = nullptr;
    int32_t *h_fp = nullptr;
    std::vector<std::vector<uint8_t>> in_exon;
    std::vector<double> psum;
    std::vector<uint64_t> masks;
    size_t job_lo = 0, job_hi = 0, job2_lo = 0, job2_hi = 0;     // this locus' ranges in the callers' job lists
    std::vector<int> job_task, job2_task;

    ~ManyRun() {
        if (st || active) (void)hipStreamSynchronize(st);
        if (holds_many) m->in_use.store(0);
        hgx_groups_destroy(groups);
        hgx_classes_destroy(ecl);
        hgx_classes_destroy(gcl);
        for (auto *t : res) delete t;
    }
    int fail_task(int t, int code) {                       // per-task error: reported through rc_out, or the whole call fails
        if (!rc_out) return code;
        rc_out[t] = code;
        delete res[t];
        res[t] = nullptr;
        return HGX_OK;
    }
    const hgx_classes *cl1() const { return hla ? ecl : gcl; }
    const std::vector<int32_t> &off1() const { return hla ? e_off : g_off; }

    int init(hgx_typing **out, int32_t *rc, const hgx_locus *loc_, const hgx_index *ix_, hgx_many *m_, const hgx_type_opts *o, hipStream_t s) {
        ARGCHK(out && loc_ && ix_ && m_ && o);
        loc = loc_; ix = ix_; m = m_; opts = o; st = s; rc_out = rc;
        n = m->n_tasks;
        for (int t = 0; t < n; ++t) { out[t] = nullptr; if (rc_out) rc_out[t] = HGX_OK; }
        int rc_ = hgx_index_dims(ix, &A, &a_pad, nullptr, nullptr);
        if (rc_) return rc_;
        ARGCHK(A == loc->A && a_pad == loc->a_pad && A == m->A && a_pad == m->a_pad);
        w64 = a_pad / 64;
        hla = loc->base_kind == HGX_BASE_HLA;
        res.assign((size_t)n, nullptr);
        for (int t = 0; t < n; ++t) {
            hgx_typing *ty = new hgx_typing();
            ty->n_reads = m->n_reads[t]; ty->n_pairs = m->pair_base[t + 1] - m->pair_base[t]; ty->n_pieces = m->n_pieces[t];
            ty->n_refs = m->n_refs[t]; ty->n_alleles = A;
            res[t] = ty;
        }
        active = n > 0 && m->db->n_pairs > 0;              // core:1589-1590: loci without reads are skipped
        return HGX_OK;
    }

    // Scoring + dedup of ALL tasks' pairs (the kernels of the one-task path) in two halves.  score_first() goes as far as the
    // classes EM #1 reads (HLA: the exon-level classes; other bases: the gene-level classes and the counts, i.e. everything);
    // score_rest() is the gene level of an HLA locus and every task's Gene_counts, which nothing needs before the hand-off to
    // EM #2: the caller runs it BESIDE the launch of EM #1 (as the one-task path overlaps its gene side).
    const uint64_t *compat = nullptr;
    int score_first() {
        if (!active) return HGX_OK;
        const hgx_dbatch *db = m->db;
        const int32_t n_pairs = db->n_pairs;
        ALLOC(b_compat, (size_t)std::max(db->n_pieces, 1) * w64 * 8);
        ALLOC(b_pt, (size_t)(2 * n + 2) * 4);
        compat = b_compat.as<uint64_t>();
        int rc = hgx_piece_compat(ix, db->d_pieces, db->d_masks, db->n_pieces, b_compat.as<uint64_t>(), st);
        if (rc) return rc;
        if (!hla) return score_gene();
        rc = hgx_group_pairs_seg(&groups, db->d_pair_off, db->d_pair_ref, n_pairs, HGX_LEVEL_EXON, m->d_pair_seg, st);
        if (rc) return rc;
        int64_t ng = 0;
        rc = hgx_groups_dims(groups, &ng, nullptr);
        if (rc) return rc;
        const size_t n_rows = ng > 0 ? (size_t)ng : (size_t)n_pairs;
        ALLOC(b_ebits, n_rows * w64 * 8);
        ALLOC(b_ehash, n_rows * 8);
        rc = hgx_level_classes_grouped_seg(&ecl, ix, compat, db->d_pair_off, db->d_pair_ref, groups, b_ebits.as<uint64_t>(),
                                           b_ehash.as<uint64_t>(), m->d_pair_seg, st);
        if (rc) return rc;
        return class_offsets(ecl, m, b_pt.as<int32_t>(), e_off, st);
    }
    int score_rest() { return active && hla ? score_gene() : HGX_OK; }
    int score_gene() {
        const hgx_dbatch *db = m->db;
        const int32_t n_pairs = db->n_pairs;
        ALLOC(b_gbits, (size_t)n_pairs * w64 * 8);
        ALLOC(b_ghash, (size_t)n_pairs * 8);
        int rc = hgx_pair_classes(ix, compat, db->d_pair_off, db->d_pair_ref, n_pairs, nullptr, b_gbits.as<uint64_t>(), nullptr,
                                  b_ghash.as<uint64_t>(), st);
        if (rc) return rc;
        rc = hgx_dedup_classes_seg(&gcl, b_gbits.as<uint64_t>(), b_ghash.as<uint64_t>(), n_pairs, a_pad, m->d_pair_seg, st);
        if (rc) return rc;
        rc = class_offsets(gcl, m, b_pt.as<int32_t>(), g_off, st);
        if (rc) return rc;
        ALLOC(b_goff, (size_t)(n + 1) * 4);
        ALLOC(b_cnt, (size_t)n * a_pad * 8);
        ALLOC(b_fp, (size_t)n * a_pad * 4);
        { int rc_ = hgx_h2d(b_goff.p, g_off.data(), (size_t)(n + 1) * 4, st); if (rc_) return rc_; }
        rc = hgx_many_counts(gcl, b_goff.as<int32_t>(), m->d_pair_base, n, b_cnt.as<int64_t>(), b_fp.as<int32_t>(), st);
        if (rc) return rc;
        if (m->in_use.exchange(1) != 0) { hgx_set_error("hgx_many: one hgx_type_many call at a time per merged batch"); return HGX_EINVAL; }
        holds_many = true;
        const size_t need = (size_t)n * a_pad * 12;
        if (m->h_pinned_bytes < need) {
            if (m->h_pinned) (void)hipHostFree(m->h_pinned);
            m->h_pinned = nullptr; m->h_pinned_bytes = 0;
            HIPCHK(hipHostMalloc(&m->h_pinned, need, hipHostMallocDefault));
            m->h_pinned_bytes = need;
        }
        h_cnt = (int64_t *)m->h_pinned;
        h_fp = (int32_t *)((char *)m->h_pinned + (size_t)n * a_pad * 8);
        HIPCHK(hipMemcpyAsync(h_cnt, b_cnt.p, (size_t)n * a_pad * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(h_fp, b_fp.p, (size_t)n * a_pad * 4, hipMemcpyDeviceToHost, st));
        return HGX_OK;
    }

    // EM #1 of every task (HLA: exon classes; other bases: gene classes, no pruning): jobs appended to the callers' list
    int em1_jobs(std::vector<hgx_emx_job> &jobs) {
        job_lo = jobs.size();
        if (active)
            for (int t = 0; t < n; ++t) {
                const int32_t C = off1()[t + 1] - off1()[t];
                if (!res[t] || res[t]->n_reads <= 0 || C == 0) continue;
                if (!hla && C == 1) {                           // core:1784-1787, quirk Q3
                    hgx_set_error("'dict_keys' object is not subscriptable (reference quirk Q3, typing_core.py:1787)");
                    const int rc = fail_task(t, HGX_ETYPE);
                    if (rc) return rc;
                    continue;
                }
                hgx_emx_job J{};
                J.bits = cl1()->d_bits + (size_t)off1()[t] * w64; J.count = cl1()->d_count + off1()[t]; J.rank = m->d_rank;
                J.C = C; J.w64 = w64; J.a_pad = a_pad; J.remove_low = hla ? (opts->remove_low ? 1 : 0) : 0;
                J.fast = em_mode_many(opts->em_fast) > 0 ? 1 : 0;
                J.any_size = em_mode_many(opts->em_fast) < 0 ? 1 : 0;
                J.prob = nullptr; J.first = nullptr; J.n_out = A;
                jobs.push_back(J);
                job_task.push_back(t);
            }
        job_hi = jobs.size();
        return HGX_OK;
    }

    // Gene_counts of every task from the staging block into the results (the scoring stream must be drained); the ranking itself
    // (core:1650-1651, a sort of every counted allele) is left to the first caller that asks for it (hgx_typing::ensure_ranked)
    bool counts_taken = false;
    void take_counts() {
        counts_taken = true;
        if (!active) return;
        for (int t = 0; t < n; ++t) {
            hgx_typing *ty = res[t];
            if (!ty || ty->n_reads <= 0) continue;
            const int64_t *cnt = h_cnt + t * (size_t)a_pad;
            const int32_t *fp = h_fp + t * (size_t)a_pad;
            ty->cnt.assign(cnt, cnt + A);
            ty->first_pair.assign(fp, fp + A);
            ty->ranked = false;
        }
        if (holds_many) { m->in_use.store(0); holds_many = false; }
    }

    // EM #1's results -> the tasks; exon_alleles and the hand-off jobs (core:1739-1766)
    int after_em1(const std::vector<hgx_emx_job> &jobs, const std::vector<hgx_emx_rec> &recs, std::vector<hgx_emx_job> &jobs2) {
        job2_lo = job2_hi = jobs2.size();
        if (!active) return HGX_OK;
        int rc = HGX_OK;
        for (size_t k = job_lo; k < job_hi; ++k) {
            const int t = job_task[k - job_lo];
            hgx_typing *ty = res[t];
            if (!ty) continue;
            const hgx_emx_job &J = jobs[k];
            if (J.status == 2) {
                hgx_set_error("EM: allele missing from the next estimate (the reference raises KeyError here, common:1365-1369)");
                rc = fail_task(t, HGX_EKEY);
                if (rc) return rc;
                continue;
            }
            if (J.status == 1) {                                // beyond the batched kernel's limits: this task's EM through the one-task path
                ClassesView v(cl1(), off1()[t], J.C);
                rc = run_em(&v.c, loc, J.remove_low, nullptr, st, ty);
                if (rc) { rc = fail_task(t, rc); if (rc) return rc; }
                continue;
            }
            EmOut o;
            em_out_from(J, recs, loc, 0, o, true);
            ty->em.push_back(std::move(o));
        }
        if (!counts_taken) take_counts();
        if (!hla) {
            for (int t = 0; t < n; ++t) if (res[t] && !res[t]->em.empty()) res[t]->gene_prob = res[t]->em[0];
            return HGX_OK;
        }
        in_exon.assign((size_t)n, {}); psum.assign((size_t)n, 0.0);
        masks.assign((size_t)n * w64, 0);
        ALLOC(b_masks, masks.size() * 8);
        for (int t = 0; t < n; ++t) {
            hgx_typing *ty = res[t];
            if (!ty || ty->em.empty()) continue;
            ty->gene_prob = ty->em[0];
            if (!exon_alleles_of(ty->em[0], loc, in_exon[t], psum[t])) continue;
            uint64_t *mk = &masks[(size_t)t * w64];
            for (int32_t a = 0; a < A; ++a) if (in_exon[t][a]) mk[a >> 6] |= 1ull << (a & 63);
            hgx_emx_job J{};
            J.bits = gcl->d_bits + (size_t)g_off[t] * w64; J.count = gcl->d_count + g_off[t]; J.rank = m->d_rank; J.len = m->d_len;
            J.mask = b_masks.as<uint64_t>() + (size_t)t * w64;
            J.C = g_off[t + 1] - g_off[t]; J.w64 = w64; J.a_pad = a_pad; J.remove_low = 1;
            J.prob = nullptr; J.first = nullptr; J.n_out = A;
            jobs2.push_back(J);
            job2_task.push_back(t);
        }
        job2_hi = jobs2.size();
        if (job2_hi > job2_lo) HIPCHK(hipMemcpyAsync(b_masks.p, masks.data(), masks.size() * 8, hipMemcpyHostToDevice, st));
        return HGX_OK;
    }

    // EM #2's results, the combination of the two levels (core:1771-1782); hands the results out
    int finish(const std::vector<hgx_emx_job> &jobs2, const std::vector<hgx_emx_rec> &recs2, hgx_typing **out) {
        int rc = HGX_OK;
        for (size_t k = job2_lo; k < job2_hi; ++k) {
            const int t = job2_task[k - job2_lo];
            hgx_typing *ty = res[t];
            const hgx_emx_job &J = jobs2[k];
            if (J.status == 2) {
                hgx_set_error("EM: allele missing from the next estimate (the reference raises KeyError here, common:1365-1369)");
                rc = fail_task(t, HGX_EKEY);
                if (rc) return rc;
                continue;
            }
            EmOut e2;
            if (J.status == 1) {                                // more than 64 alleles pass the filter (or too many merged classes)
                ClassesView v(gcl, g_off[t], J.C);
                int32_t it2 = 0, ncls2 = 0;
                std::vector<double> p2((size_t)A);
                std::vector<int32_t> f2((size_t)A);
                rc = hgx_classes_set_allele_rank(&v.c, loc->name_rank.data(), A);
                if (!rc) rc = hgx_em_masked(&v.c, &masks[(size_t)t * w64], A, 1, loc->allele_len.data(), p2.data(), f2.data(), &it2, &ncls2, st);
                if (rc) { rc = fail_task(t, rc); if (rc) return rc; continue; }
                e2.exact = hgx_em_last_exact() != 0;
                e2.n_classes = ncls2; e2.n_iter = it2; e2.remove_low = 1; e2.use_length = 1;
                sorted_result(p2, f2, loc->name_rank.data(), A, e2);
            } else {
                em_out_from(J, recs2, loc, 1, e2, false);      // (as the one-task path's hand-off kernel orders its result)
            }
            combine_levels(ty, std::move(e2), in_exon[t], psum[t]);
        }
        for (int t = 0; t < n; ++t) out[t] = res[t];
        res.clear();
        return HGX_OK;
    }
};

// The phases of all loci: the first half of the scoring side by side (one host thread and stream per locus: a dozen round trips
// each), ONE launch for the EM #1 of every task of every locus (longest first: the launch's makespan is the longest task's time
// plus what queues behind it) with the second half of the scoring (gene level of the HLA loci, Gene_counts -> results) beside it
// (measured on the 384-task panel: 13.7 -> 12.3 ms per call; the launch itself 6.5 -> 7.5 ms with 2.6 ms of scoring kernels on
// the same CUs; their streams' priority makes no difference), the hand-off set-up, ONE launch for every EM #2.
int run_many(std::vector<ManyRun> &runs, hgx_typing ***out, hipStream_t st) {
    const bool prof = getenv("HGX_TYPE_PROFILE") != nullptr;
    const bool rest_first = hgx_switch_has("many", "rest_first");       // the gene level BEFORE the launch of EM #1 (comparison)
    double tp[6];
    tp[0] = now_s();
    int rc = HGX_OK;
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    const size_t nr = runs.size();
    std::vector<StreamSet> sets(nr);
    struct SetsGuard { std::vector<StreamSet> &s; ~SetsGuard() { for (auto &x : s) release_streams(x); } } sets_guard{sets};
    std::vector<int> rcs(nr, HGX_OK);
    std::vector<std::string> errs(nr);
    for (size_t i = 0; i < nr; ++i) { rc = acquire_streams(sets[i]); if (rc) return rc; }
    HIPCHK(hipStreamSynchronize(st));
    // one phase of every locus on its own host thread and on the stream the caller names; joined and drained before returning
    auto side_by_side = [&](const std::function<int(ManyRun &)> &phase, const std::function<void()> &meanwhile) {
        std::vector<std::thread> th;
        for (size_t i = 0; i < nr; ++i)
            th.emplace_back([&, i] {
                if (hipSetDevice(dev) != hipSuccess) { rcs[i] = HGX_EHIP; errs[i] = "hipSetDevice failed on a scoring thread"; return; }
                ManyRun &r = runs[i];
                const hipStream_t main_st = r.st;
                r.st = sets[i].em;
                rcs[i] = phase(r);
                if (hgx_sync(r.st) != HGX_OK && rcs[i] == HGX_OK) rcs[i] = HGX_EHIP;
                if (rcs[i]) errs[i] = hgx_last_error();
                r.st = main_st;
            });
        if (meanwhile) meanwhile();
        for (auto &t : th) t.join();
        for (size_t i = 0; i < nr; ++i)
            if (rcs[i]) { hgx_set_error("%s", errs[i].c_str()); return rcs[i]; }
        return (int)HGX_OK;
    };
    rc = side_by_side([&](ManyRun &r) { int c = r.score_first(); if (!c && rest_first) c = r.score_rest(); return c; }, nullptr);
    if (rc) return rc;
    tp[1] = now_s();
    std::vector<hgx_emx_job> jobs, jobs2;
    std::vector<hgx_emx_rec> recs, recs2;
    for (auto &r : runs) { rc = r.em1_jobs(jobs); if (rc) return rc; }
    int rc_em = HGX_OK;
    std::string err_em;
    rc = side_by_side([&](ManyRun &r) {
                          int c = rest_first ? (int)HGX_OK : r.score_rest();
                          if (!c && hgx_sync(r.st) != HGX_OK) c = HGX_EHIP;
                          if (!c) r.take_counts();                                       // (host work behind the EM launch)
                          return c;
                      },
                      [&] { rc_em = hgx_emx_run(jobs.data(), (int)jobs.size(), st, &recs); if (rc_em) err_em = hgx_last_error(); tp[5] = now_s(); });
    if (rc_em) { hgx_set_error("%s", err_em.c_str()); return rc_em; }
    if (rc) return rc;
    tp[2] = now_s();
    for (auto &r : runs) { rc = r.after_em1(jobs, recs, jobs2); if (rc) return rc; }
    tp[3] = now_s();
    if (!jobs2.empty()) {
        HIPCHK(hipStreamSynchronize(st));                          // (the masks were copied from pageable memory)
        rc = hgx_emx_run(jobs2.data(), (int)jobs2.size(), st, &recs2);
        if (rc) return rc;
    }
    tp[4] = now_s();
    const double em_share = ((tp[5] - tp[1]) + (tp[4] - tp[3])) / std::max<size_t>(jobs.size(), 1);
    size_t n_tasks = 0;
    for (size_t i = 0; i < runs.size(); ++i) {
        for (auto *t : runs[i].res) if (t) t->t_em = em_share;
        n_tasks += runs[i].n;
        rc = runs[i].finish(jobs2, recs2, out[i]);
        if (rc) return rc;
    }
    if (prof)
        fprintf(stderr, "[hgx_type_many] %zu loci, %zu tasks: scoring up to EM #1's classes %.2f ms | EM #1 (%zu jobs) %.2f, with the rest of the scoring beside it %.2f | hand-off set-up %.2f | "
                        "EM #2 (%zu jobs) %.2f | results %.2f\n", runs.size(), n_tasks, (tp[1] - tp[0]) * 1e3, jobs.size(), (tp[5] - tp[1]) * 1e3, (tp[2] - tp[1]) * 1e3,
                (tp[3] - tp[2]) * 1e3, jobs2.size(), (tp[4] - tp[3]) * 1e3, (now_s() - tp[4]) * 1e3);
    return HGX_OK;
}

}   // namespace

extern "C" int hgx_type_many(hgx_typing **out, int32_t *rc_out, const hgx_locus *loc, const hgx_index *ix, hgx_many *m,
                             const hgx_type_opts *opts, void *stream) {
    ARGCHK(opts);
    EmFastScope em_mode(em_mode_many(opts->em_fast));        // (tasks that fall back to the one-task EM follow the same setting)
    std::vector<ManyRun> runs(1);
    int rc = runs[0].init(out, rc_out, loc, ix, m, opts, (hipStream_t)stream);
    if (rc) return rc;
    hgx_typing **outs[1] = {out};
    rc = run_many(runs, outs, (hipStream_t)stream);
    if (rc) for (int t = 0; t < m->n_tasks; ++t) { if (out[t]) { delete out[t]; out[t] = nullptr; } }
    return rc;
}

extern "C" int hgx_type_many_loci(int32_t n_loci, hgx_typing ***out, int32_t **rc_out, const hgx_locus *const *loci, const hgx_index *const *ixs,
                                  hgx_many *const *manies, const hgx_type_opts *opts, void *stream) {
    ARGCHK(n_loci >= 0 && (n_loci == 0 || (out && loci && ixs && manies)) && opts);
    // every locus is checked and its outputs are cleared BEFORE anything runs: the clean-up below may then visit any of them
    for (int i = 0; i < n_loci; ++i) {
        ARGCHK(out[i] && loci[i] && ixs[i] && manies[i]);
        for (int t = 0; t < manies[i]->n_tasks; ++t) out[i][t] = nullptr;
    }
    EmFastScope em_mode(em_mode_many(opts->em_fast));
    std::vector<ManyRun> runs((size_t)n_loci);
    int rc = HGX_OK;
    for (int i = 0; i < n_loci && !rc; ++i) rc = runs[i].init(out[i], rc_out ? rc_out[i] : nullptr, loci[i], ixs[i], manies[i], opts, (hipStream_t)stream);
    if (!rc) rc = run_many(runs, out, (hipStream_t)stream);
    if (rc)
        for (int i = 0; i < n_loci; ++i)
            for (int t = 0; t < manies[i]->n_tasks; ++t) { if (out[i][t]) { delete out[i][t]; out[i][t] = nullptr; } }
    return rc;
}

// The calls of many results at once (what a throughput run looks at): per result the reads, the single_abundance calls made and
// the first `k` alleles of the final Gene_prob with their abundances (-1 / 0.0 beyond the end of a shorter list).
extern "C" int hgx_typing_top(const hgx_typing *const *ts, int32_t n, int32_t k, int32_t *n_reads, int32_t *n_em, int32_t *allele, double *prob) {
    ARGCHK(n >= 0 && k >= 0 && (n == 0 || ts));
    for (int32_t i = 0; i < n; ++i) {
        const hgx_typing *t = ts[i];
        if (n_reads) n_reads[i] = t ? t->n_reads : 0;
        if (n_em) n_em[i] = t ? (int32_t)t->em.size() : 0;
        for (int32_t j = 0; j < k; ++j) {
            const bool have = t && (size_t)j < t->gene_prob.allele.size();
            if (allele) allele[(size_t)i * k + j] = have ? t->gene_prob.allele[j] : -1;
            if (prob) prob[(size_t)i * k + j] = have ? t->gene_prob.prob[j] : 0.0;
        }
    }
    return HGX_OK;
}
