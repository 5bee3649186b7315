// How fast do host threads fill pinned memory, and how fast does it travel?  hipHostMalloc vs malloc + hipHostRegister vs plain malloc.
// build: hipcc --offload-arch=gfx950 -O2 -pthread tools/pinned_probe.hip -o /tmp/pinned_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void fill(char *dst, const char *src, size_t n, int nt) {
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back([=] { size_t b = n * t / nt, e = n * (t + 1) / nt; memcpy(dst + b, src + b, e - b); });
    for (auto &x : th) x.join();
}
int main(int argc, char **argv) {
    const size_t n = 400u << 20;
    const int nt = argc > 1 ? atoi(argv[1]) : 32;
    char *src = (char *)malloc(n);
    memset(src, 7, n);
    void *dev = nullptr;
    hipMalloc(&dev, n);
    hipStream_t st;
    hipStreamCreate(&st);
    for (int mode = 0; mode < 4; ++mode) {
        char *p = nullptr;
        const char *name = "";
        double t0 = now();
        if (mode == 0) { name = "hipHostMalloc default"; hipHostMalloc((void **)&p, n, hipHostMallocDefault); }
        else if (mode == 1) { name = "hipHostMalloc numa-user"; hipHostMalloc((void **)&p, n, hipHostMallocNumaUser); }
        else if (mode == 2) { name = "malloc + hipHostRegister"; p = (char *)aligned_alloc(2u << 20, n); fill(p, src, n, nt); hipHostRegister(p, n, hipHostRegisterDefault); }
        else { name = "plain malloc (pageable)"; p = (char *)aligned_alloc(2u << 20, n); }
        const double t_alloc = now() - t0;
        double best_fill = 1e9, best_up = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            t0 = now();
            fill(p, src, n, nt);
            best_fill = std::min(best_fill, now() - t0);
            t0 = now();
            hipMemcpyAsync(dev, p, n, hipMemcpyHostToDevice, st);
            hipStreamSynchronize(st);
            best_up = std::min(best_up, now() - t0);
        }
        printf("%-26s alloc %7.1f ms | fill by %d threads %6.2f ms (%5.1f GB/s) | H2D %6.2f ms (%5.1f GB/s)\n", name, t_alloc, nt, best_fill, n / best_fill / 1e6,
               best_up, n / best_up / 1e6);
    }
    return 0;
}
