// Host round-trip cost of "kernel -> small D2H -> hipStreamSynchronize" under the device scheduling flags.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ void k_touch(double *p) { p[0] += 1.0; }
int main(int argc, char **argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const unsigned flags[4] = {hipDeviceScheduleAuto, hipDeviceScheduleSpin, hipDeviceScheduleYield, hipDeviceScheduleBlockingSync};
    if (hipSetDeviceFlags(flags[mode]) != hipSuccess) printf("setflags failed\n");
    double *d, *h;
    hipMalloc(&d, 64); hipHostMalloc(&h, 64); hipMemset(d, 0, 64);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (int pinned = 0; pinned < 2; ++pinned) {
        double stack[8];
        double *dst = pinned ? h : stack;
        for (int i = 0; i < 50; ++i) { hipLaunchKernelGGL(k_touch, dim3(1), dim3(1), 0, st, d); hipMemcpyAsync(dst, d, 8, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); }
        auto t0 = std::chrono::steady_clock::now();
        const int N = 2000;
        for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(k_touch, dim3(1), dim3(1), 0, st, d); hipMemcpyAsync(dst, d, 8, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); }
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        printf("mode %d (%s) %s dst: %.1f us per kernel+D2H+sync\n", mode, mode == 0 ? "auto" : mode == 1 ? "spin" : mode == 2 ? "yield" : "blocking", pinned ? "pinned" : "pageable", us);
    }
    return 0;
}
