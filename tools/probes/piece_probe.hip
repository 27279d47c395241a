// What does a workgroup that uses only PART of every 128-byte line cost?  (node attention: one pair's E row is 64 heads x 2 bytes
// = one 128-byte line; a workgroup of HG heads uses HG x 2 bytes of it and 64 / HG workgroups on the same XCD use the rest.)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/piece_probe.hip -o tools/probes/piece_probe && tools/probes/piece_probe
//
// A tensor of L lines is read once (and, in the second half of the table, an equally large one written once) by workgroups of 512
// threads.  A "set" = 768 consecutive lines (16 queries x 48 keys); a workgroup = (set, piece p of P bytes): every thread moves 16
// bytes, P / 16 threads per line.  The 128 / P workgroups of a set sit next to each other on ONE XCD (block index & 7 = XCD, the
// launch order of csrc/node_attention16.hip), so that a line comes from HBM once and the rest are L2 hits.  P = 128 is the fully
// coalesced reference.  Printed: useful TB/s (bytes of the tensor / time).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int kThreads = 512, kSetLines = 768;

template <int P, bool WRITE>
__global__ void __launch_bounds__(kThreads) piece_kernel(const char* __restrict__ src, char* __restrict__ dst, int sets, float* out) {
    constexpr int G = 128 / P, TPL = P / 16;               // workgroups per set, threads per line
    const int x = blockIdx.x & 7, t = blockIdx.x >> 3;
    const int nt = ((sets - x + 7) >> 3) * G;
    if (t >= nt) return;
    const int set = (t / G) * 8 + x, p = t % G;
    const char* s = src + (size_t)set * kSetLines * 128 + p * P;
    char* d = dst + (size_t)set * kSetLines * 128 + p * P;
    uint32_t acc = 0;
    constexpr int kChunks = kSetLines * TPL, kIters = (kChunks + kThreads - 1) / kThreads;
    uint4 v[kIters < 4 ? kIters : 4];
#pragma unroll
    for (int base = 0; base < kIters; base += 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = (base + i) * kThreads + threadIdx.x;
            if (base + i < kIters && c < kChunks) v[i] = *reinterpret_cast<const uint4*>(s + (size_t)(c / TPL) * 128 + (c % TPL) * 16);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = (base + i) * kThreads + threadIdx.x;
            if (base + i < kIters && c < kChunks) {
                if (WRITE) *reinterpret_cast<uint4*>(d + (size_t)(c / TPL) * 128 + (c % TPL) * 16) = v[i];
                else acc += v[i].x ^ v[i].w;
            }
        }
    }
    if (!WRITE && acc == 0x12345u) out[0] = 1.f;
}

template <int P, bool WRITE>
static void run(const char* src, char* dst, int sets, float* out) {
    constexpr int G = 128 / P;
    const int grid = ((sets + 7) / 8) * 8 * G;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((piece_kernel<P, WRITE>), dim3(grid), dim3(kThreads), 0, 0, src, dst, sets, out);
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((piece_kernel<P, WRITE>), dim3(grid), dim3(kThreads), 0, 0, src, dst, sets, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)sets * kSetLines * 128 * (WRITE ? 2 : 1);
    printf("piece %3d B  %s  %7.1f us  %6.2f TB/s useful\n", P, WRITE ? "read+write" : "read only ", ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
}

int main() {
    const int sets = 3072;                                  // 3072 x 768 x 128 B = 302 MB (config 4: 128 graphs x 3 query blocks x (E | G) x ...)
    const size_t bytes = (size_t)sets * kSetLines * 128;
    char *src, *dst; float* out;
    hipMalloc(&src, bytes); hipMalloc(&dst, bytes); hipMalloc(&out, 4);
    hipMemset(src, 1, bytes); hipMemset(dst, 0, bytes);
    run<128, false>(src, dst, sets, out); run<64, false>(src, dst, sets, out); run<32, false>(src, dst, sets, out); run<16, false>(src, dst, sets, out);
    run<128, true>(src, dst, sets, out); run<64, true>(src, dst, sets, out); run<32, true>(src, dst, sets, out); run<16, true>(src, dst, sets, out);
    hipDeviceSynchronize();
    return 0;
}
