// fake_rccl.cpp -- TEST INFRASTRUCTURE, not product: a stand-in for librccl's transport for ranks that share ONE GPU (RCCL itself
// refuses two ranks on one device, and this pool's boxes have one GPU).  It exports the eight entry points libhgx and dist.RcclComm bind
// -- ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclCommCount, ncclCommUserRank, ncclGetErrorString, ncclAllReduce (sum),
// ncclAllGather, ncclBroadcast -- with RCCL's signatures and datatype codes (rccl.h), and moves the bytes between PROCESSES through files
// in /dev/shm and a shared-memory barrier: device -> host copy after a stream synchronisation, exchange, host -> device copy.  What it lets
// the GPU suite run with a world of 2-4 ranks is everything AROUND the transport: hgx_index_broadcast, hgx_allreduce_sum_u32 / _i64 and
// hgx_classes_allgather (sizes gather, padded gather, rank-order unpack, weighted merge) through the real C-ABI, dist.RcclComm,
// dist.type_locus_sharded's device-side exchanges and bench.py's `comm_kind: rccl` body.  Nothing under hisat-genotype_amd/ knows this
// file; the tests point the loader at it (test switch `rccl_lib`, dist.RcclComm.lib_path).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

extern "C" {
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
}

namespace {
struct Ctl { std::atomic<uint32_t> count, sense; };
struct Comm {
    int rank = 0, n = 1;
    uint32_t sense = 0;
    std::string base;
    Ctl *ctl = nullptr;
    int fd_ctl = -1;
    std::vector<unsigned char> host;
};
size_t dtype_size(int dt) {
    switch (dt) { case 0: case 1: return 1; case 2: case 3: case 7: return 4; case 4: case 5: case 8: return 8; case 6: return 2; default: return 0; }
}
void barrier(Comm *c) {
    if (c->n == 1) return;
    const uint32_t my = c->sense ^= 1u;
    if (c->ctl->count.fetch_add(1) + 1 == (uint32_t)c->n) { c->ctl->count.store(0); c->ctl->sense.store(my); }
    else while (c->ctl->sense.load() != my) sched_yield();
}
std::string file_of(const Comm *c, int r) { return c->base + ".r" + std::to_string(r); }
bool put(const Comm *c, const void *p, size_t n) {
    const int fd = open(file_of(c, c->rank).c_str(), O_CREAT | O_RDWR | O_TRUNC, 0600);
    if (fd < 0) return false;
    size_t done = 0;
    while (done < n) { const ssize_t w = pwrite(fd, (const char *)p + done, n - done, (off_t)done); if (w <= 0) { close(fd); return false; } done += (size_t)w; }
    close(fd);
    return true;
}
bool get(const Comm *c, int r, void *p, size_t n) {
    const int fd = open(file_of(c, r).c_str(), O_RDONLY);
    if (fd < 0) return false;
    size_t done = 0;
    while (done < n) { const ssize_t g = pread(fd, (char *)p + done, n - done, (off_t)done); if (g <= 0) { close(fd); return false; } done += (size_t)g; }
    close(fd);
    return true;
}
// every rank's `bytes` -> all[r] on every rank (the building block of the three collectives)
bool exchange(Comm *c, const void *dev_send, size_t bytes, hipStream_t st, std::vector<std::vector<unsigned char>> &all, int only_from = -1) {
    if (hipStreamSynchronize(st) != hipSuccess) return false;
    c->host.resize(bytes ? bytes : 1);
    if (bytes && hipMemcpy(c->host.data(), dev_send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return false;
    if ((only_from < 0 || only_from == c->rank) && !put(c, c->host.data(), bytes)) return false;
    barrier(c);
    all.assign((size_t)c->n, {});
    for (int r = 0; r < c->n; ++r) {
        if (only_from >= 0 && r != only_from) continue;
        all[r].resize(bytes ? bytes : 1);
        if (r == c->rank) memcpy(all[r].data(), c->host.data(), bytes);
        else if (bytes && !get(c, r, all[r].data(), bytes)) return false;
    }
    barrier(c);                                   // nobody rewrites its file before everybody has read it
    return true;
}
}   // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    memset(id->internal, 0, sizeof id->internal);
    FILE *f = fopen("/dev/urandom", "rb");
    unsigned long long a = 0, b = 0;
    if (f) { (void)!fread(&a, 8, 1, f); (void)!fread(&b, 8, 1, f); fclose(f); }
    snprintf(id->internal, sizeof id->internal, "/dev/shm/hgx_fake_rccl_%016llx%016llx", a, b ^ (unsigned long long)getpid());
    return 0;
}
ncclResult_t ncclCommInitRank(void **comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return 4;
    Comm *c = new Comm();
    c->rank = rank; c->n = nranks;
    c->base = std::string(id.internal, strnlen(id.internal, sizeof id.internal));
    c->fd_ctl = open((c->base + ".ctl").c_str(), O_CREAT | O_RDWR, 0600);
    if (c->fd_ctl < 0 || ftruncate(c->fd_ctl, 4096) != 0) { delete c; return 2; }
    void *m = mmap(nullptr, 4096, PROT_READ | PROT_WRITE, MAP_SHARED, c->fd_ctl, 0);
    if (m == MAP_FAILED) { delete c; return 2; }
    c->ctl = (Ctl *)m;
    barrier(c);                                   // every rank has the control block mapped
    *comm = c;
    return 0;
}
ncclResult_t ncclCommDestroy(void *comm) {
    Comm *c = (Comm *)comm;
    if (!c) return 0;
    unlink(file_of(c, c->rank).c_str());
    if (c->rank == 0) unlink((c->base + ".ctl").c_str());
    munmap(c->ctl, 4096);
    close(c->fd_ctl);
    delete c;
    return 0;
}
ncclResult_t ncclCommCount(const void *comm, int *count) { *count = ((const Comm *)comm)->n; return 0; }
ncclResult_t ncclCommUserRank(const void *comm, int *rank) { *rank = ((const Comm *)comm)->rank; return 0; }
const char *ncclGetErrorString(ncclResult_t r) { return r == 0 ? "no error" : "fake_rccl: the exchange through /dev/shm failed"; }

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t st) {
    Comm *c = (Comm *)comm;
    const size_t bytes = count * dtype_size(dtype);
    if (!dtype_size(dtype)) return 4;
    std::vector<std::vector<unsigned char>> all;
    if (!exchange(c, send, bytes, st, all)) return 1;
    for (int r = 0; r < c->n; ++r)
        if (bytes && hipMemcpy((char *)recv + (size_t)r * bytes, all[r].data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return 1;
    return 0;
}
ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, int dtype, int op, void *comm, hipStream_t st) {
    Comm *c = (Comm *)comm;
    const size_t es = dtype_size(dtype), bytes = count * es;
    if (op != 0 || !(dtype == 2 || dtype == 3 || dtype == 4 || dtype == 5)) return 4;       // sums of 32- / 64-bit integers: what libhgx asks for
    std::vector<std::vector<unsigned char>> all;
    if (!exchange(c, send, bytes, st, all)) return 1;
    std::vector<unsigned char> out(bytes ? bytes : 1, 0);
    for (int r = 0; r < c->n; ++r) {
        if (es == 4) { uint32_t *o = (uint32_t *)out.data(); const uint32_t *x = (const uint32_t *)all[r].data(); for (size_t i = 0; i < count; ++i) o[i] += x[i]; }
        else { uint64_t *o = (uint64_t *)out.data(); const uint64_t *x = (const uint64_t *)all[r].data(); for (size_t i = 0; i < count; ++i) o[i] += x[i]; }
    }
    if (bytes && hipMemcpy(recv, out.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return 1;
    return 0;
}
ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, int dtype, int root, void *comm, hipStream_t st) {
    Comm *c = (Comm *)comm;
    const size_t bytes = count * dtype_size(dtype);
    if (!dtype_size(dtype) || root < 0 || root >= c->n) return 4;
    std::vector<std::vector<unsigned char>> all;
    if (!exchange(c, send, bytes, st, all, root)) return 1;
    if (bytes && hipMemcpy(recv, all[root].data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return 1;
    return 0;
}
}
