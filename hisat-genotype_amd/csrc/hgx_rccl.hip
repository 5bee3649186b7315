// hgx_rccl.hip -- 8e: the three exchanges of a sharded locus and the index broadcast as C-ABI entry points on DEVICE buffers,
// for callers that hold an RCCL communicator (ncclComm_t).  SURVEY.md 8(b) lists hgx_index_broadcast(handle, root, rccl_comm);
// the reference has no such layer (its concurrency is one process per sample, /root/reference/hisatgenotype:613-665).
// librccl is loaded on first use (dlopen): libhgx itself does not link against it.
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "hgx_common.hpp"

namespace {

typedef int ncclResult_t;
typedef void *ncclComm_t;
enum { NCCL_UINT8 = 1, NCCL_INT32 = 2, NCCL_UINT32 = 3, NCCL_INT64 = 4, NCCL_SUM = 0 };   // rccl.h: ncclDataType_t / ncclRedOp_t

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    std::string why;              // why loading failed (dlopen's message, captured once)
};
Rccl g_rccl;
std::once_flag g_rccl_once;

int rccl_load() {
    std::call_once(g_rccl_once, [] {
        // (test switch rccl_lib = a path: the GPU suite's stand-in transport for ranks that share one GPU, tests/fake_rccl; set before
        // the first collective of the process -- the library is loaded once)
        const char *sw = hgx_test_switch("rccl_lib");
        if (sw && *sw) {
            g_rccl.lib = dlopen(sw, RTLD_NOW | RTLD_LOCAL);
            if (!g_rccl.lib) { if (const char *e = dlerror()) g_rccl.why = e; return; }
        }
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (g_rccl.lib) break;
            g_rccl.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (g_rccl.lib) break;
            if (const char *e = dlerror()) g_rccl.why = e;        // captured once, here: dlerror() clears itself when read
        }
        if (!g_rccl.lib) return;
        auto sym = [&](const char *n) { return dlsym(g_rccl.lib, n); };
        g_rccl.Broadcast = (decltype(g_rccl.Broadcast))sym("ncclBroadcast");
        g_rccl.AllReduce = (decltype(g_rccl.AllReduce))sym("ncclAllReduce");
        g_rccl.AllGather = (decltype(g_rccl.AllGather))sym("ncclAllGather");
        g_rccl.CommCount = (decltype(g_rccl.CommCount))sym("ncclCommCount");
        g_rccl.CommUserRank = (decltype(g_rccl.CommUserRank))sym("ncclCommUserRank");
        g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
        g_rccl.ok = g_rccl.Broadcast && g_rccl.AllReduce && g_rccl.AllGather && g_rccl.CommCount && g_rccl.CommUserRank;
        if (!g_rccl.ok) g_rccl.why = "symbols missing";
    });
    if (!g_rccl.ok) { hgx_set_error("librccl could not be loaded (%s)", g_rccl.why.empty() ? "not found" : g_rccl.why.c_str()); return HGX_EHIP; }
    return HGX_OK;
}
#define RCCLCHK(expr)                                                                                                   \
    do {                                                                                                                \
        const ncclResult_t r_ = (expr);                                                                                 \
        if (r_ != 0) {                                                                                                  \
            hgx_set_error("%s failed: %s", #expr, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "rccl error");     \
            return HGX_EHIP;                                                                                            \
        }                                                                                                               \
    } while (0)

std::atomic<uint64_t> g_stat_calls{0}, g_stat_send{0}, g_stat_recv{0};
inline void rccl_count(uint64_t sent, uint64_t received) { g_stat_calls += 1; g_stat_send += sent; g_stat_recv += received; }

}   // namespace

extern "C" int hgx_index_device_block(const hgx_index *ix, void **dev_block, size_t *bytes);

extern "C" int hgx_index_broadcast(hgx_index *ix, int32_t root, void *comm, void *stream) {
    ARGCHK(ix && comm && root >= 0);
    { int rc_ = rccl_load(); if (rc_) return rc_; }
    void *block = nullptr;
    size_t bytes = 0;
    { int rc_ = hgx_index_device_block(ix, &block, &bytes); if (rc_) return rc_; }
    RCCLCHK(g_rccl.Broadcast(block, block, bytes, NCCL_UINT8, root, (ncclComm_t)comm, (hipStream_t)stream));
    rccl_count(bytes, bytes);
    return HGX_OK;
}

extern "C" int hgx_allreduce_sum_u32(uint32_t *buf, size_t n, void *comm, void *stream) {
    ARGCHK(buf && comm);
    { int rc_ = rccl_load(); if (rc_) return rc_; }
    RCCLCHK(g_rccl.AllReduce(buf, buf, n, NCCL_UINT32, NCCL_SUM, (ncclComm_t)comm, (hipStream_t)stream));
    rccl_count(n * 4, n * 4);
    return HGX_OK;
}
extern "C" int hgx_allreduce_sum_i64(int64_t *buf, size_t n, void *comm, void *stream) {
    ARGCHK(buf && comm);
    { int rc_ = rccl_load(); if (rc_) return rc_; }
    RCCLCHK(g_rccl.AllReduce(buf, buf, n, NCCL_INT64, NCCL_SUM, (ncclComm_t)comm, (hipStream_t)stream));
    rccl_count(n * 8, n * 8);
    return HGX_OK;
}

// ---- the class-table exchange in three parts: pack, all-gather, unpack + merge ------------------------------------------------
// A rank's table travels as rows [bits (w64 words) | count], padded with zero rows to `cap` = the largest table of the group;
// the receive buffer holds the ranks' blocks in rank order.  The two halves around the collective are entry points of their
// own (a caller with another transport -- MPI, files -- can use them; tests/test_gpu_typing.py assembles receive buffers of
// world 2-5 from them to exercise the rank-order unpack without a second GPU).
extern "C" int hgx_classes_pack_rows(const hgx_classes *mine, int32_t a_pad, int32_t cap, void *dev_send, void *stream) {
    ARGCHK(dev_send && cap >= 1 && a_pad > 0 && a_pad % 512 == 0 && (!mine || (mine->a_pad == a_pad && mine->n_classes <= cap)));
    hipStream_t st = (hipStream_t)stream;
    const int w64 = a_pad / 64;
    const size_t pitch = (size_t)(w64 + 1) * 8;
    const int32_t my_c = mine ? mine->n_classes : 0;
    if (mine) hgx_classes_order_after(mine, st);
    HIPCHK(hipMemsetAsync(dev_send, 0, (size_t)cap * pitch, st));
    if (my_c > 0) {
        HIPCHK(hipMemcpy2DAsync(dev_send, pitch, mine->d_bits, (size_t)w64 * 8, (size_t)w64 * 8, (size_t)my_c, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpy2DAsync((char *)dev_send + (size_t)w64 * 8, pitch, mine->d_count, 8, 8, (size_t)my_c, hipMemcpyDeviceToDevice, st));
    }
    return HGX_OK;
}

// `dev_recv`: world blocks of cap rows; sizes[r] = rows of rank r that are real.  Rows are concatenated in rank order -- which
// is stream order of the ranks' pairs, so first-seen order survives -- and equal rows merged with summed counts.
extern "C" int hgx_classes_merge_gathered(hgx_classes **out, const void *dev_recv, const int32_t *sizes, int32_t world, int32_t cap,
                                          int32_t a_pad, void *stream) {
    ARGCHK(out && dev_recv && sizes && world >= 1 && cap >= 1 && a_pad > 0 && a_pad % 512 == 0);
    *out = nullptr;
    hipStream_t st = (hipStream_t)stream;
    const int w64 = a_pad / 64;
    const size_t pitch = (size_t)(w64 + 1) * 8;
    int64_t total = 0;
    for (int r = 0; r < world; ++r) {
        ARGCHK(sizes[r] >= 0 && sizes[r] <= cap);
        total += sizes[r];
    }
    DevBuf b_rows, b_w;
    // declared after the buffers = destroyed before them: an error return behind a queued copy drains the stream before the
    // buffers go back to the pool
    struct Drain { hipStream_t s; ~Drain() { (void)hipStreamSynchronize(s); } } drain{st};
    ALLOC(b_rows, (size_t)std::max<int64_t>(total, 1) * w64 * 8);
    ALLOC(b_w, (size_t)std::max<int64_t>(total, 1) * 8);
    int64_t at = 0;
    for (int r = 0; r < world; ++r) {
        if (sizes[r] <= 0) continue;
        const char *src = (const char *)dev_recv + (size_t)r * cap * pitch;
        HIPCHK(hipMemcpy2DAsync((char *)b_rows.p + (size_t)at * w64 * 8, (size_t)w64 * 8, src, pitch, (size_t)w64 * 8, (size_t)sizes[r],
                                hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpy2DAsync((char *)b_w.p + (size_t)at * 8, 8, src + (size_t)w64 * 8, pitch, 8, (size_t)sizes[r], hipMemcpyDeviceToDevice, st));
        at += sizes[r];
    }
    int rc = hgx_dedup_classes(out, b_rows.as<uint64_t>(), nullptr, b_w.as<int64_t>(), total, a_pad, nullptr, stream);
    if (rc) return rc;
    { int rc_ = hgx_sync(st); if (rc_) return rc_; }          // (the row / weight buffers are read by kernels the dedup queued)
    return HGX_OK;
}

extern "C" int hgx_classes_allgather(hgx_classes **out, const hgx_classes *mine, int32_t a_pad, void *comm, void *stream) {
    ARGCHK(out && comm && a_pad > 0 && a_pad % 512 == 0 && (!mine || mine->a_pad == a_pad));
    *out = nullptr;
    { int rc_ = rccl_load(); if (rc_) return rc_; }
    hipStream_t st = (hipStream_t)stream;
    int world = 0, rank = 0;
    RCCLCHK(g_rccl.CommCount((ncclComm_t)comm, &world));
    RCCLCHK(g_rccl.CommUserRank((ncclComm_t)comm, &rank));
    const int w64 = a_pad / 64;
    const int32_t my_c = mine ? mine->n_classes : 0;
    DevBuf b_sizes;
    ALLOC(b_sizes, (size_t)(world + 1) * 4);
    int32_t *d_sizes = b_sizes.as<int32_t>();
    { int rc_ = hgx_h2d(d_sizes + world, &my_c, 4, st); if (rc_) return rc_; }
    RCCLCHK(g_rccl.AllGather(d_sizes + world, d_sizes, 1, NCCL_INT32, (ncclComm_t)comm, st));
    std::vector<int32_t> sizes((size_t)world);
    { int rc_ = hgx_d2h(sizes.data(), d_sizes, (size_t)world * 4, st); if (rc_) return rc_; }
    { int rc_ = hgx_sync(st); if (rc_) return rc_; }
    int32_t cap = 1;
    for (int32_t c : sizes) cap = std::max(cap, c);
    const size_t pitch = (size_t)(w64 + 1) * 8;
    DevBuf b_send, b_recv;
    struct Drain { hipStream_t s; ~Drain() { (void)hipStreamSynchronize(s); } } drain{st};
    ALLOC(b_send, (size_t)cap * pitch);
    ALLOC(b_recv, (size_t)world * cap * pitch);
    { int rc_ = hgx_classes_pack_rows(mine, a_pad, cap, b_send.p, stream); if (rc_) return rc_; }
    RCCLCHK(g_rccl.AllGather(b_send.p, b_recv.p, (size_t)cap * (w64 + 1), NCCL_INT64, (ncclComm_t)comm, st));
    g_stat_calls += 2;
    g_stat_send += (uint64_t)cap * pitch + 4;
    g_stat_recv += (uint64_t)world * ((uint64_t)cap * pitch + 4);
    return hgx_classes_merge_gathered(out, b_recv.p, sizes.data(), world, cap, a_pad, stream);
}

// bytes this process handed to / received from the collectives above since the last reset (bench.py: exchange bytes per step)
extern "C" int hgx_rccl_stats(uint64_t *n_collectives, uint64_t *bytes_sent, uint64_t *bytes_received, int32_t reset) {
    if (n_collectives) *n_collectives = g_stat_calls.load();
    if (bytes_sent) *bytes_sent = g_stat_send.load();
    if (bytes_received) *bytes_received = g_stat_recv.load();
    if (reset) { g_stat_calls = 0; g_stat_send = 0; g_stat_recv = 0; }
    return HGX_OK;
}
