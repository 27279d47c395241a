// Fused GELU + dropout (the middle of the FFN block, reference
// lib/tgt/layers/layers.py:157-158: `x = gelu(lin_W1(x)); x = dropout(x)`) for gfx950.
//
// Pure streaming: one read + one write forward, two reads + one write backward, instead
// of four passes + a mask tensor each way.  The keep/drop decision of element i is a
// counter-based hash of (seed, i), recomputed in the backward -- no mask is stored.
//   y  = keep(i) ? gelu(x) / (1-p) : 0          gelu(x) = x * 0.5 * (1 + erf(x / sqrt2))
//   dx = keep(i) ? dy * gelu'(x) / (1-p) : 0    gelu'(x) = 0.5 (1 + erf(x/sqrt2)) + x exp(-x^2/2)/sqrt(2 pi)
#include <cstdlib>
#include "common.hpp"

namespace tgt {

// keep/drop of the V consecutive elements of vector `vec` (= first element index / V): one hash
// of (seed, vec), then one 32-bit word per TWO elements, 16 bits each;
// P(keep) = 1 - thresh16 / 65536.
template <int V>
__device__ __forceinline__ void keep_vector(uint64_t seed, int64_t vec, uint32_t thresh16, bool (&keep)[V]) {
    const uint32_t lo = (uint32_t)vec, hi = (uint32_t)((uint64_t)vec >> 32);
    const uint32_t base = mix32(lo ^ (uint32_t)seed) ^ mix32(hi + (uint32_t)(seed >> 32) + 0x9e3779b9u);
#pragma unroll
    for (int w = 0; w < V / 2; ++w) {
        const uint32_t r = mix32(base + (uint32_t)(w + 1) * 0x9e3779b9u);
        keep[2 * w] = (r & 0xffffu) >= thresh16;
        keep[2 * w + 1] = (r >> 16) >= thresh16;
    }
}
// Phi(v) = 0.5 (1 + erf(v / sqrt2)) and exp(-v^2/2): erf by Abramowitz & Stegun 7.1.26
// (|error| <= 1.5e-7, below fp32 parity tolerance); the kernel is otherwise ALU-bound on erff.
__device__ __forceinline__ float gelu_cdf(float v, float& e) {
    const float ax = fabsf(v) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * ax);
    e = __expf(-ax * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    return 0.5f + copysignf(0.5f - 0.5f * poly * e, v);
}

template <typename T, bool BWD>
__global__ void __launch_bounds__(256) gelu_dropout_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                          T* __restrict__ out, int64_t n, uint64_t seed,
                                                          uint32_t thresh, float inv_keep) {
    constexpr int V = 16 / (int)sizeof(T);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * V;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * V; i < n; i += stride) {
        T xv[V], gv[V], ov[V];
        if (i + V <= n) {
            uint4 raw = *reinterpret_cast<const uint4*>(x + i);
            __builtin_memcpy(xv, &raw, 16);
            if (BWD) {
                uint4 rg = *reinterpret_cast<const uint4*>(dy + i);
                __builtin_memcpy(gv, &rg, 16);
            }
        } else {
            for (int t = 0; t < V; ++t) {
                xv[t] = i + t < n ? x[i + t] : from_f32<T>(0.f);
                if (BWD) gv[t] = i + t < n ? dy[i + t] : from_f32<T>(0.f);
            }
        }
        bool keep[V];
        if (thresh != 0u) keep_vector<V>(seed, i / V, thresh, keep);
#pragma unroll
        for (int t = 0; t < V; ++t) {
            const float v = to_f32(xv[t]);
            float e;
            const float cdf = gelu_cdf(v, e);
            float r;
            if (!BWD) r = v * cdf;
            else r = to_f32(gv[t]) * (cdf + v * 0.3989422804014327f * e);
            ov[t] = from_f32<T>((thresh == 0u || keep[t]) ? r * inv_keep : 0.f);
        }
        if (i + V <= n) {
            uint4 raw;
            __builtin_memcpy(&raw, ov, 16);
            *reinterpret_cast<uint4*>(out + i) = raw;
        } else {
            for (int t = 0; t < V && i + t < n; ++t) out[i + t] = ov[t];
        }
    }
}

template <typename T>
static int gd_launch(const void* x, const void* dy, void* out, int64_t n, float p, uint64_t seed, bool bwd,
                     hipStream_t st) {
    const uint32_t thresh = p <= 0.f ? 0u : (uint32_t)fmin(65535.0, fmax(1.0, nearbyint((double)p * 65536.0)));   // 16-bit
    const float inv_keep = p <= 0.f ? 1.f : 1.f / (1.f - p);
    constexpr int V = 16 / (int)sizeof(T);
    int64_t blocks = (n / V + 255) / 256;
    // one 16-byte vector per thread, no revisits: +5 % over a 4096-workgroup grid-stride grid (a plain
    // copy shows the same: tools/probes/hbm_probe.hip)
    static const int64_t cap = getenv("TGT_EW_GRID_CAP") ? atoll(getenv("TGT_EW_GRID_CAP")) : (int64_t)1 << 30;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    if (!bwd)
        hipLaunchKernelGGL((gelu_dropout_kernel<T, false>), dim3((unsigned)blocks), dim3(256), 0, st,
                           reinterpret_cast<const T*>(x), nullptr, reinterpret_cast<T*>(out), n, seed, thresh, inv_keep);
    else
        hipLaunchKernelGGL((gelu_dropout_kernel<T, true>), dim3((unsigned)blocks), dim3(256), 0, st,
                           reinterpret_cast<const T*>(x), reinterpret_cast<const T*>(dy), reinterpret_cast<T*>(out), n,
                           seed, thresh, inv_keep);
    return check_launch(bwd ? "gelu_dropout_bwd_kernel" : "gelu_dropout_fwd_kernel");
}

int gelu_dropout_run(const void* x, const void* dy, void* out, int64_t n, int dtype, float p, uint64_t seed, bool bwd,
                     hipStream_t st) {
    if (!x || !out || n < 0 || (bwd && !dy)) return set_error(TGT_ERR_INVALID, "gelu_dropout: null tensor");
    if (p < 0.f || p >= 1.f) return set_error(TGT_ERR_INVALID, "gelu_dropout: p=%f outside [0,1)", p);
    if (((uintptr_t)x | (uintptr_t)out | (uintptr_t)dy) % 16) return set_error(TGT_ERR_INVALID, "gelu_dropout: tensors must be 16-byte aligned");
    if (n == 0) return TGT_OK;
    switch (dtype) {
        case TGT_F32: return gd_launch<float>(x, dy, out, n, p, seed, bwd, st);
        case TGT_BF16: return gd_launch<bf16_t>(x, dy, out, n, p, seed, bwd, st);
        case TGT_F16: return gd_launch<f16_t>(x, dy, out, n, p, seed, bwd, st);
        default: return set_error(TGT_ERR_INVALID, "gelu_dropout: bad dtype %d", dtype);
    }
}

}  // namespace tgt
