// In-kernel closing sums against a tiny second kernel ("reduce inside the producing kernel").
//   hipcc --offload-arch=gfx950 -O3 tools/probes/fanin_probe.hip -o tools/probes/fanin_probe && tools/probes/fanin_probe
//
// A persistent "producer" (256 workgroups x 1024 threads, one per CU, like the row kernels of csrc/edge_gemm.hip) streams its share
// of a tensor (read + write, so that the chip is busy the way it is at the end of an LN_BWD launch) and ends with a per-workgroup
// partial row of C floats (C = 768: dgamma | dbeta | bias-gradient column sums of the LayerNorm-backward epilogue; C = 1600 x 4
// rows per workgroup-quad: the triplet backward's per-graph column sums).  The P partial rows have to become ONE row of C floats,
// in a fixed order (bit-identical from run to run).  Three ways:
//   A  two launches: the producer writes its row with plain stores; a tiny kernel with the structure of tgt_sum_planes
//      (csrc/params.hip: 8 plane slices x 32 float4 columns per block, 4 loads in flight) adds the rows.  Cost = the kernel
//      boundary + the tiny kernel.  This is what the repo does.
//   B  last arriver (cdna_hip_programming.md Guideline 16, counter form): rows written write-through (sc1), every wave drains,
//      one relaxed agent-scope ticket per workgroup; the workgroup that draws P-1 makes ONE agent-scope acquire and adds all P rows
//      (all 1024 threads: 32 plane slices x 32 float4 columns per pass, LDS fold), alone, while the rest of the chip is idle.
//   C  grid barrier + everybody reduces a slice: after the ticket every workgroup spins (relaxed sc1 loads + s_sleep) until all P
//      have arrived, acquires once and adds its C / P columns (needs all P workgroups resident: one per CU).
// Measured (MI355X, profiles/r06l_fanin_probe.txt): 768 KB of partials: A +11.4-12.1 us over the producer alone, B +19.5-20.6, C +12.9-13.4;
// 1.6 MB: A +12.4, B +35-36, C +16.5-17; 6.4 MB: A +13.4, B +115, C +30.5.  The tiny second launch is the cheapest form.
// Printed: microseconds per iteration of (producer + reduction) for A, B, C and of the producer alone, so that the price of each
// form of the closing sum is (form - alone).  The sums are checked against each other (each form has its own fixed order).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef __attribute__((address_space(1))) unsigned gu32;

template <int MODE>   // 0: no partials (producer alone), 1: A (plain partial rows), 2: B (last arriver), 3: C (barrier + slices)
__global__ void __launch_bounds__(1024) producer(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16,
                                                 float* part, int C, unsigned* ticket, float* out) {
    const int tid = threadIdx.x, P = gridDim.x;
    // the streaming body: this workgroup's contiguous share
    const size_t per = n16 / P, base = (size_t)blockIdx.x * per;
    float acc = 0.f;
    for (size_t i = tid; i < per; i += 1024) {
        uint4 v = src[base + i];
        acc += __uint_as_float(v.x & 0x3f800000u);
        dst[base + i] = v;
    }
    if (MODE == 0) { if (acc == 123.f) out[0] = acc; return; }
    // the workgroup's partial row: C floats (a function of the block and the column, plus what was read: keeps the loads alive)
    float* row = part + (size_t)blockIdx.x * C;
    for (int c = tid; c < C; c += 1024) {
        const float v = (float)((blockIdx.x * 131 + c * 7) % 1000) * 1e-3f + acc * 0.f;
        if (MODE == 1) row[c] = v;
        else __hip_atomic_store(row + c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // write-through (sc1): no release fence needed
    }
    if (MODE == 1) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave drains
    __syncthreads();
    __shared__ unsigned drawn;
    if (tid == 0) drawn = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (MODE == 2) {
        if (drawn != (unsigned)(P - 1)) return;
        if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
        // all 1024 threads: 32 plane slices x 32 float4 columns per pass, four loads in flight per thread, the slices meet in LDS
        // in a fixed order (the structure of tgt_sum_planes, csrc/params.hip, inside ONE workgroup)
        __shared__ float4 red[32][32];
        const int cl = tid & 31, sl = tid >> 5, nv = C / 4;
        const float4* pv = reinterpret_cast<const float4*>(part);
        for (int cb = 0; cb < nv; cb += 32) {
            const int col = cb + cl;
            float4 s0 = {0, 0, 0, 0}, s1 = s0;
            if (col < nv) {
                for (int q = sl; q + 32 < P; q += 64) {
                    const float4 a = pv[(size_t)q * nv + col], b = pv[(size_t)(q + 32) * nv + col];
                    s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w; s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
                }
                s0.x += s1.x; s0.y += s1.y; s0.z += s1.z; s0.w += s1.w;
            }
            red[sl][cl] = s0;
            __syncthreads();
            if (sl == 0 && col < nv) {
                float4 t = red[0][cl];
                for (int k = 1; k < 32; ++k) { t.x += red[k][cl].x; t.y += red[k][cl].y; t.z += red[k][cl].z; t.w += red[k][cl].w; }
                reinterpret_cast<float4*>(out)[col] = t;
            }
            __syncthreads();
        }
        if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // (re-armed for the next launch)
        return;
    }
    // MODE 3: wait for everybody (one lane polls, relaxed, with a bound), one acquire, then this workgroup's slice of the columns
    if (tid == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)P && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(2);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    // float4 column j of the result belongs to workgroup j % P: thread p < P loads row p's piece, fixed-order fold in LDS
    __shared__ float4 red3[256];
    const int nv = C / 4;
    const float4* pv = reinterpret_cast<const float4*>(part);
    for (int j = blockIdx.x; j < nv; j += P) {
        if (tid < 256) red3[tid] = tid < P ? pv[(size_t)tid * nv + j] : make_float4(0, 0, 0, 0);
        __syncthreads();
        for (int w = 128; w; w >>= 1) {
            if (tid < w) { red3[tid].x += red3[tid + w].x; red3[tid].y += red3[tid + w].y; red3[tid].z += red3[tid + w].z; red3[tid].w += red3[tid + w].w; }
            __syncthreads();
        }
        if (tid == 0) reinterpret_cast<float4*>(out)[j] = red3[0];
        __syncthreads();
    }
}

// the structure of tgt_sum_planes (csrc/params.hip): 256 threads = 8 plane slices x 32 float4 columns, 4 loads in flight
__global__ void __launch_bounds__(256) reduce_rows(const float* __restrict__ part, int P, int C, float* __restrict__ out) {
    __shared__ float4 red[8][32];
    const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5, nv = C / 4, col = blockIdx.x * 32 + cl;
    const float4* pv = reinterpret_cast<const float4*>(part);
    float4 s0 = {0, 0, 0, 0}, s1 = s0, s2 = s0, s3 = s0;
    if (col < nv) {
        for (int p = sl; p + 24 < P; p += 32) {
            const float4 a = pv[(size_t)p * nv + col], b = pv[(size_t)(p + 8) * nv + col], c = pv[(size_t)(p + 16) * nv + col], d = pv[(size_t)(p + 24) * nv + col];
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w; s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
            s2.x += c.x; s2.y += c.y; s2.z += c.z; s2.w += c.w; s3.x += d.x; s3.y += d.y; s3.z += d.z; s3.w += d.w;
        }
        s0.x += s1.x + (s2.x + s3.x); s0.y += s1.y + (s2.y + s3.y); s0.z += s1.z + (s2.z + s3.z); s0.w += s1.w + (s2.w + s3.w);
    }
    red[sl][cl] = s0;
    __syncthreads();
    if (sl == 0 && col < nv) {
        float4 t = red[0][cl];
        for (int k = 1; k < 8; ++k) { t.x += red[k][cl].x; t.y += red[k][cl].y; t.z += red[k][cl].z; t.w += red[k][cl].w; }
        reinterpret_cast<float4*>(out)[col] = t;
    }
}

int main() {
    const int P = 256;
    const size_t bytes = 128ull << 20, n16 = bytes / 16;          // one pass over a 134 MB tensor each way (one E)
    uint4 *src, *dst;
    hipMalloc(&src, bytes); hipMalloc(&dst, bytes);
    hipMemset(src, 1, bytes);
    unsigned* ticket; hipMalloc(&ticket, 4); hipMemset(ticket, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int C : {768, 1600, 6400}) {
        float *part, *outA, *outB, *outC;
        hipMalloc(&part, (size_t)P * C * 4); hipMalloc(&outA, C * 4); hipMalloc(&outB, C * 4); hipMalloc(&outC, C * 4);
        auto timeit = [&](auto fn) {
            for (int i = 0; i < 5; ++i) fn();
            hipEventRecord(e0);
            for (int i = 0; i < 50; ++i) fn();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            return ms / 50 * 1e3f;
        };
        const float t0 = timeit([&] { hipLaunchKernelGGL(producer<0>, dim3(P), dim3(1024), 0, 0, src, dst, n16, part, C, ticket, outA); });
        const float tA = timeit([&] {
            hipLaunchKernelGGL(producer<1>, dim3(P), dim3(1024), 0, 0, src, dst, n16, part, C, ticket, outA);
            hipLaunchKernelGGL(reduce_rows, dim3((C / 4 + 31) / 32), dim3(256), 0, 0, part, P, C, outA);
        });
        const float tB = timeit([&] { hipLaunchKernelGGL(producer<2>, dim3(P), dim3(1024), 0, 0, src, dst, n16, part, C, ticket, outB); });
        const float tC = timeit([&] {
            hipMemsetAsync(ticket, 0, 4, 0);
            hipLaunchKernelGGL(producer<3>, dim3(P), dim3(1024), 0, 0, src, dst, n16, part, C, ticket, outC);
        });
        hipMemsetAsync(ticket, 0, 4, 0);
        std::vector<float> a(C), b(C), c(C);
        hipMemcpy(a.data(), outA, C * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), outB, C * 4, hipMemcpyDeviceToHost);
        hipMemcpy(c.data(), outC, C * 4, hipMemcpyDeviceToHost);
        int badB = 0, badC = 0;
        for (int i = 0; i < C; ++i) { badB += (a[i] - b[i]) > 1e-3f * a[i] || (b[i] - a[i]) > 1e-3f * a[i]; badC += (a[i] - c[i]) > 1e-3f * a[i] || (c[i] - a[i]) > 1e-3f * a[i]; }
        printf("P=%d rows of C=%5d floats (%6.1f KB of partials): producer alone %7.2f us | A two launches %7.2f (+%5.2f) | "
               "B last arriver %7.2f (+%5.2f)%s | C barrier+slices %7.2f (+%5.2f)%s\n", P, C, P * C * 4 / 1024.0, t0, tA, tA - t0, tB, tB - t0,
               badB ? " MISMATCH" : "", tC, tC - t0, badC ? " MISMATCH" : "");
        hipFree(part); hipFree(outA); hipFree(outB); hipFree(outC);
    }
    return 0;
}
