// What does a host round trip cost on this box?  A dependent chain "kernel -> 16 bytes to the host -> host decides -> next kernel",
// 2 000 links, three ways: (A) hipMemcpyAsync into pinned memory + hipStreamSynchronize (what hgx_d2h / hgx_sync do), (B) a one-wave
// kernel that stores the words into mapped pinned memory and then a sequence number, the host polling that number, (C) the producing
// kernel storing words and sequence number itself.  Prints microseconds per link.
//   hipcc --offload-arch=gfx950 -O3 tools/sync_probe.hip -o /tmp/sync_probe && /tmp/sync_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <immintrin.h>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_work(uint32_t *d, uint32_t v) { if (threadIdx.x == 0) { d[0] = v; d[1] = v + 1; d[2] = v + 2; d[3] = v + 3; } }
__global__ void k_to_host(const uint32_t *d, volatile uint32_t *h, volatile uint32_t *flag, uint32_t seq) {
    if (threadIdx.x < 4) h[threadIdx.x] = d[threadIdx.x];
    __threadfence_system();
    if (threadIdx.x == 0) __hip_atomic_store((uint32_t *)flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_work_to_host(uint32_t *d, volatile uint32_t *h, volatile uint32_t *flag, uint32_t v) {
    if (threadIdx.x == 0) { d[0] = v; h[0] = v; h[1] = v + 1; h[2] = v + 2; h[3] = v + 3; }
    __threadfence_system();
    if (threadIdx.x == 0) __hip_atomic_store((uint32_t *)flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    hipStream_t st;
    CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    uint32_t *d, *h, *hd;
    CHK(hipMalloc(&d, 64));
    CHK(hipHostMalloc((void **)&h, 4096, hipHostMallocMapped));
    CHK(hipHostGetDevicePointer((void **)&hd, h, 0));
    volatile uint32_t *flag = h + 16;
    const int N = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        for (int i = 1; i <= N; ++i) {
            k_work<<<1, 64, 0, st>>>(d, (uint32_t)i);
            CHK(hipMemcpyAsync(h, d, 16, hipMemcpyDeviceToHost, st));
            CHK(hipStreamSynchronize(st));
            if (h[0] != (uint32_t)i) { fprintf(stderr, "A: wrong value\n"); return 1; }
        }
        const double a = (now() - t0) / N * 1e6;
        *flag = 0;
        t0 = now();
        for (int i = 1; i <= N; ++i) {
            k_work<<<1, 64, 0, st>>>(d, (uint32_t)i);
            k_to_host<<<1, 64, 0, st>>>(d, hd, hd + 16, (uint32_t)i);
            while (__atomic_load_n((uint32_t *)flag, __ATOMIC_ACQUIRE) != (uint32_t)i) _mm_pause();
            if (h[0] != (uint32_t)i) { fprintf(stderr, "B: wrong value\n"); return 1; }
        }
        const double b = (now() - t0) / N * 1e6;
        CHK(hipStreamSynchronize(st));
        *flag = 0;
        t0 = now();
        for (int i = 1; i <= N; ++i) {
            k_work_to_host<<<1, 64, 0, st>>>(d, hd, hd + 16, (uint32_t)i);
            while (__atomic_load_n((uint32_t *)flag, __ATOMIC_ACQUIRE) != (uint32_t)i) _mm_pause();
            if (h[3] != (uint32_t)i + 3) { fprintf(stderr, "C: wrong value\n"); return 1; }
        }
        const double c = (now() - t0) / N * 1e6;
        CHK(hipStreamSynchronize(st));
        // the floor: the same launches with no host wait in between
        t0 = now();
        for (int i = 1; i <= N; ++i) k_work<<<1, 64, 0, st>>>(d, (uint32_t)i);
        CHK(hipStreamSynchronize(st));
        const double f = (now() - t0) / N * 1e6;
        printf("per link: (A) copy + stream sync %.1f us | (B) copy kernel + polled flag %.1f us | (C) producer stores + polled flag %.1f us | back-to-back launches %.1f us\n", a, b, c, f);
    }
    return 0;
}
