// Layout probe for v_mfma_i32_16x16x64_i8 on gfx950 (exact integer data, asymmetric operands).
// Hypothesis (by analogy with the bf16 16x16x32 form): lane l holds A[row = l&15][k = 16*(l>>4) + j] and
// B[k = 16*(l>>4) + j][col = l&15] in byte j (j = 0..15) of its 128-bit operand; D[row = 4*(l>>4) + reg][col = l&15].
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void k(const int8_t* A /*16x64 row-major*/, const int8_t* B /*64x16 row-major*/, int* D /*16x16*/) {
    const int l = threadIdx.x;
    union { v4i v; int8_t b[16]; } a, b;
    for (int j = 0; j < 16; ++j) {
        a.b[j] = A[(l & 15) * 64 + 16 * (l >> 4) + j];
        b.b[j] = B[(16 * (l >> 4) + j) * 16 + (l & 15)];
    }
    v4i acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a.v, b.v, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = acc[r];
}

int main() {
    std::vector<int8_t> A(16 * 64), B(64 * 16);
    srand(1);
    for (auto& x : A) x = (int8_t)(rand() % 7 - 3);
    for (auto& x : B) x = (int8_t)(rand() % 11 - 5);
    std::vector<int> ref(256, 0), got(256);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int kk = 0; kk < 64; ++kk) ref[i * 16 + j] += A[i * 64 + kk] * B[kk * 16 + j];
    int8_t *dA, *dB; int* dD;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dD, 1024);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(got.data(), dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += got[i] != ref[i];
    printf("mfma_i32_16x16x64_i8 layout hypothesis: %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);
    return bad != 0;
}
