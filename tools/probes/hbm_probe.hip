// Practical HBM ceiling on this box: read-only, write-only and copy kernels over 1 GiB buffers.
// hipcc --offload-arch=gfx950 -O3 tools/probes/hbm_probe.hip -o tools/probes/hbm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_read(const uint4* __restrict__ a, size_t n, uint4* sink) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    uint4 acc = {0, 0, 0, 0};
    for (; i < n; i += st) { uint4 v = a[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    if (acc.x == 0x12345 && acc.y == 77) *sink = acc;
}
__global__ void k_write(uint4* __restrict__ a, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    uint4 v = {1, 2, 3, 4};
    for (; i < n; i += st) a[i] = v;
}
__global__ void k_copy(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) b[i] = a[i];
}
// one 16-byte chunk per thread, no loop (the shape of an elementwise kernel)
__global__ void k_copy1(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i];
}
int main() {
    const size_t bytes = 1ull << 30, n = bytes / 16;
    uint4 *a, *b;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto fn, double moved) {
        for (int i = 0; i < 3; ++i) fn();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) fn();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
        printf("%-28s %8.3f ms  %7.1f GB/s\n", name, ms, moved / ms / 1e6);
    };
    for (int grid : {2048, 4096, 8192, 16384, 65536}) {
        char nm[64];
        snprintf(nm, 64, "read  grid=%d x256", grid);  run(nm, [&] { k_read<<<grid, 256>>>(a, n, b); }, bytes);
        snprintf(nm, 64, "write grid=%d x256", grid);  run(nm, [&] { k_write<<<grid, 256>>>(b, n); }, bytes);
        snprintf(nm, 64, "copy  grid=%d x256", grid);  run(nm, [&] { k_copy<<<grid, 256>>>(a, b, n); }, 2.0 * bytes);
    }
    run("copy1 one chunk/thread", [&] { k_copy1<<<(unsigned)(n / 256), 256>>>(a, b, n); }, 2.0 * bytes);
    run("hipMemcpyDtoD", [&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }, 2.0 * bytes);
    return 0;
}
