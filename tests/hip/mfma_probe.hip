// Stand-alone check of the matrix-core operand/result lane maps that
// tgt_amd/csrc/common.hpp assumes (run on the GPU box: hipcc + ./a.out).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void k_bf16(const float* A, const float* B, float* Cc) {   // A[32][16], B[16][32], C[32][32]
    int l = threadIdx.x, r = l & 31, hi = l >> 5;
    bf16x8 a, b;
    for (int t = 0; t < 8; ++t) { a[t] = (__bf16)A[r * 16 + 8 * hi + t]; b[t] = (__bf16)B[(8 * hi + t) * 32 + r]; }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int q = 0; q < 16; ++q) Cc[((q & 3) + 8 * (q >> 2) + 4 * hi) * 32 + r] = c[q];
}
__global__ void k_f32(const float* A, const float* B, float* Cc) {    // K = 16 via 8 x (32x32x2)
    int l = threadIdx.x, r = l & 31, hi = l >> 5;
    f32x16 c = {0};
    for (int t = 0; t < 8; ++t)
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * 16 + 8 * hi + t], B[(8 * hi + t) * 32 + r], c, 0, 0, 0);
    for (int q = 0; q < 16; ++q) Cc[((q & 3) + 8 * (q >> 2) + 4 * hi) * 32 + r] = c[q];
}
int main() {
    float hA[512], hB[512], hC[1024], ref[1024];
    srand(1);
    for (int i = 0; i < 512; ++i) { hA[i] = (rand() % 17 - 8) / 4.f; hB[i] = (rand() % 13 - 6) / 2.f; }
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { float s = 0; for (int k = 0; k < 16; ++k) s += hA[m * 16 + k] * hB[k * 32 + n]; ref[m * 32 + n] = s; }
    float *dA, *dB, *dC;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 4096);
    hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
    int bad = 0;
    for (int which = 0; which < 2; ++which) {
        hipMemset(dC, 0, 4096);
        if (which == 0) hipLaunchKernelGGL(k_bf16, dim3(1), dim3(64), 0, 0, dA, dB, dC);
        else hipLaunchKernelGGL(k_f32, dim3(1), dim3(64), 0, 0, dA, dB, dC);
        hipMemcpy(hC, dC, 4096, hipMemcpyDeviceToHost);
        double err = 0;
        for (int i = 0; i < 1024; ++i) err = fmax(err, fabs(hC[i] - ref[i]));
        printf("MFMA_PROBE %s max_err=%g %s\n", which ? "f32_32x32x2" : "bf16_32x32x16", err, err < 1e-3 ? "OK" : "MISMATCH");
        if (err >= 1e-3) { bad = 1; for (int m = 0; m < 4; ++m) { for (int n = 0; n < 8; ++n) printf("%7.2f/%7.2f ", hC[m * 32 + n], ref[m * 32 + n]); printf("\n"); } }
    }
    return bad;
}
