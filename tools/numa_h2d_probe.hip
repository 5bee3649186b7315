// Where should the pinned staging of the file -> result call live?  Host -> device rate of a 410 MB registered buffer whose pages sit on
// NUMA node k (mbind), on the node of whoever touched them first (1 thread / 32 threads), and in 4 back-to-back quarter copies as the
// SAM reader sends them.  Prints the box's nodes and the GPU's own node first.
// build: hipcc --offload-arch=gfx950 -O2 -pthread tools/numa_h2d_probe.hip -o /tmp/numa_h2d_probe
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static std::string slurp(const std::string &p) { std::ifstream f(p); std::string s; std::getline(f, s); return s; }
static void touch(char *p, size_t n, int nt) {
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back([=] { size_t b = n * t / nt, e = n * (t + 1) / nt; memset(p + b, 3 + t, e - b); });
    for (auto &x : th) x.join();
}
int main() {
    const size_t n = 410u << 20;
    int n_nodes = 0;
    for (; n_nodes < 16; ++n_nodes) {
        const std::string c = slurp("/sys/devices/system/node/node" + std::to_string(n_nodes) + "/cpulist");
        if (c.empty()) break;
        printf("node %d: cpus %s\n", n_nodes, c.c_str());
    }
    char bdf[64] = "";
    hipDeviceGetPCIBusId(bdf, sizeof(bdf), 0);
    for (char *q = bdf; *q; ++q) *q = (char)tolower(*q);
    printf("GPU 0 at %s, numa_node %s; this thread on cpu %d\n", bdf, slurp(std::string("/sys/bus/pci/devices/") + bdf + "/numa_node").c_str(), sched_getcpu());
    void *dev = nullptr;
    if (hipMalloc(&dev, n) != hipSuccess) return 1;
    hipStream_t st;
    hipStreamCreate(&st);
    auto run = [&](const char *name, int node, int touch_threads) {
        char *p = (char *)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (p == MAP_FAILED) { printf("%s: mmap failed\n", name); return; }
        if (node >= 0) {
            unsigned long mask = 1ul << node;
            if (syscall(SYS_mbind, p, n, 2 /* MPOL_BIND */, &mask, sizeof(mask) * 8, 0) != 0) { printf("%s: mbind failed (%s)\n", name, strerror(errno)); munmap(p, n); return; }
        }
        double t0 = now();
        touch(p, n, touch_threads);
        const double t_touch = now() - t0;
        t0 = now();
        if (hipHostRegister(p, n, hipHostRegisterDefault) != hipSuccess) { printf("%s: register failed\n", name); munmap(p, n); return; }
        const double t_reg = now() - t0;
        double best = 1e9, best4 = 1e9, fill = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            t0 = now();
            touch(p, n, 32);
            fill = std::min(fill, now() - t0);
            t0 = now();
            hipMemcpyAsync(dev, p, n, hipMemcpyHostToDevice, st);
            hipStreamSynchronize(st);
            best = std::min(best, now() - t0);
            t0 = now();
            for (int q = 0; q < 4; ++q) hipMemcpyAsync((char *)dev + n / 4 * q, p + n / 4 * q, n / 4, hipMemcpyHostToDevice, st);
            hipStreamSynchronize(st);
            best4 = std::min(best4, now() - t0);
        }
        printf("%-34s touch %6.1f ms register %6.1f ms | refill by 32 threads %5.2f ms | H2D one copy %5.2f ms (%5.1f GB/s) | four quarters %5.2f ms (%5.1f GB/s)\n",
               name, t_touch, t_reg, fill, best, n / best / 1e6, best4, n / best4 / 1e6);
        hipHostUnregister(p);
        munmap(p, n);
    };
    run("first touch by this thread", -1, 1);
    run("first touch by 32 threads", -1, 32);
    for (int k = 0; k < n_nodes; ++k) {
        char nm[64];
        snprintf(nm, sizeof(nm), "bound to node %d (32 threads)", k);
        run(nm, k, 32);
    }
    return 0;
}
