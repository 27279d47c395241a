// Gaussian basis of the 3-D distance embedding (reference lib/models/pcqm/layers.py:129-157: `gaussian()` and
// GaussianLayer.forward), forward and backward -- gfx950.
//
//   t[p]   = mul[p] * x[p] + bias[p]                      (p = one (b, i, j) node pair; mul, bias: summed type embeddings)
//   sigma_k = |std_k| + 0.01,  z = (t - mean_k) / sigma_k
//   y[p,k] = exp(-z^2 / 2) / (sqrt(2 * 3.14159) * sigma_k)          (sic: the reference's constant)
// ATen runs this as ~8 elementwise passes over the (B,N,N,K) tensor (134 MB in float32 at the BASELINE batch) forward and
// as many again backward.  Here: one pass each way, the result written directly in the dtype the consuming Linear
// computes in (under autocast the reference's float32 result is cast to it by that Linear: same rounding), the backward
// recomputes y from the 4-byte-per-pair inputs, reduces the mean / std gradients over all pairs in registers (one partial
// row per wave, fixed-order final sum by tgt_sum_rows) and emits d t per pair (dmul = dt * x, dbias = dt).
// Mapping: one 64-lane wave per pair, lane <-> kernels {2 lane, 2 lane + 1} (+128 v): a pair's K values are one
// contiguous store; HBM-bound streaming (K * sizeof(T) bytes per pair).
#include "common.hpp"

namespace tgt {

constexpr float kGaussNorm = 2.5066272159f;               // (2 * 3.14159) ** 0.5

template <typename T, int VPL>
__global__ void __launch_bounds__(256) gaussian_fwd_kernel(const float* x, const float* mul, const float* bias, const float* mean,
                                                            const float* std_p, int64_t pairs, int K, T* y) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    float mu[VPL][2], isg[VPL][2], coef[VPL][2];
#pragma unroll
    for (int v = 0; v < VPL; ++v)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int k = v * 128 + lane * 2 + c;
            const float sg = k < K ? fabsf(std_p[k]) + 1e-2f : 1.f;
            mu[v][c] = k < K ? mean[k] : 0.f;
            isg[v][c] = 1.f / sg;
            coef[v][c] = 1.f / (kGaussNorm * sg);
        }
    for (int64_t p = wave; p < pairs; p += nwaves) {
        const float t = mul[p] * x[p] + bias[p];
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            const int k = v * 128 + lane * 2;
            if (k < K) {
                float o[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const float z = (t - mu[v][c]) * isg[v][c];
                    o[c] = __expf(-0.5f * z * z) * coef[v][c];
                }
                if constexpr (sizeof(T) == 4) {
                    *reinterpret_cast<float2*>(y + p * K + k) = make_float2(o[0], o[1]);
                } else {
                    T tt[2] = {from_f32<T>(o[0]), from_f32<T>(o[1])};
                    uint32_t raw;
                    __builtin_memcpy(&raw, tt, 4);
                    *reinterpret_cast<uint32_t*>(y + p * K + k) = raw;
                }
            }
        }
    }
}

// g = dL/dy.  dt[p] = sum_k g y (-z / sigma);  partial[wave] = [ sum_p g y z / sigma | sum_p g y (z^2 - 1) / sigma * sign(std) ] (2K)
template <typename T, int VPL>
__global__ void __launch_bounds__(256) gaussian_bwd_kernel(const float* x, const float* mul, const float* bias, const float* mean,
                                                            const float* std_p, const T* g, int64_t pairs, int K, float* dt,
                                                            float* partial) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    float mu[VPL][2], isg[VPL][2], coef[VPL][2], dmu[VPL][2], dsg[VPL][2];
#pragma unroll
    for (int v = 0; v < VPL; ++v)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int k = v * 128 + lane * 2 + c;
            const float sg = k < K ? fabsf(std_p[k]) + 1e-2f : 1.f;
            mu[v][c] = k < K ? mean[k] : 0.f;
            isg[v][c] = 1.f / sg;
            coef[v][c] = 1.f / (kGaussNorm * sg);
            dmu[v][c] = dsg[v][c] = 0.f;
        }
    for (int64_t p = wave; p < pairs; p += nwaves) {
        const float t = mul[p] * x[p] + bias[p];
        float acc = 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            const int k = v * 128 + lane * 2;
            if (k < K) {
                float gv[2];
                if constexpr (sizeof(T) == 4) {
                    const float2 r = *reinterpret_cast<const float2*>(g + p * K + k);
                    gv[0] = r.x; gv[1] = r.y;
                } else {
                    const uint32_t raw = *reinterpret_cast<const uint32_t*>(g + p * K + k);
                    T tt[2];
                    __builtin_memcpy(tt, &raw, 4);
                    gv[0] = to_f32(tt[0]); gv[1] = to_f32(tt[1]);
                }
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const float z = (t - mu[v][c]) * isg[v][c];
                    const float gy = gv[c] * __expf(-0.5f * z * z) * coef[v][c] * isg[v][c];     // g y / sigma
                    acc -= gy * z;
                    dmu[v][c] += gy * z;
                    dsg[v][c] += gy * (z * z - 1.f);
                }
            }
        }
        acc = group_sum<64>(acc);
        if (lane == 0) dt[p] = acc;
    }
    float* row = partial + wave * 2 * K;
#pragma unroll
    for (int v = 0; v < VPL; ++v)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int k = v * 128 + lane * 2 + c;
            if (k < K) {
                row[k] = dmu[v][c];
                row[K + k] = std_p[k] < 0.f ? -dsg[v][c] : dsg[v][c];      // d|s|/ds (0 at s == 0, as torch.abs: sign(0) = 0 ...)
                if (std_p[k] == 0.f) row[K + k] = 0.f;
            }
        }
}

static int gauss_grid(int64_t pairs) {
    int64_t blocks = (pairs + 3) / 4;
    return (int)(blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks));
}
int gaussian_parts(int64_t pairs) { return gauss_grid(pairs) * 4; }

template <typename T>
static int gauss_launch(const float* x, const float* mul, const float* bias, const float* mean, const float* std_p, int64_t pairs,
                        int K, void* y, const void* g, float* dt, float* partial, hipStream_t st) {
    const int grid = gauss_grid(pairs);
    const int vpl = (K + 127) / 128;
#define TGT_GAUSS(V)                                                                                                                   \
    if (!g) hipLaunchKernelGGL((gaussian_fwd_kernel<T, V>), dim3(grid), dim3(256), 0, st, x, mul, bias, mean, std_p, pairs, K,         \
                               reinterpret_cast<T*>(y));                                                                               \
    else hipLaunchKernelGGL((gaussian_bwd_kernel<T, V>), dim3(grid), dim3(256), 0, st, x, mul, bias, mean, std_p,                      \
                            reinterpret_cast<const T*>(g), pairs, K, dt, partial)
    if (vpl == 1) { TGT_GAUSS(1); } else if (vpl == 2) { TGT_GAUSS(2); } else { TGT_GAUSS(4); }
#undef TGT_GAUSS
    return check_launch(g ? "gaussian_bwd_kernel" : "gaussian_fwd_kernel");
}

int gaussian_run(const float* x, const float* mul, const float* bias, const float* mean, const float* std_p, int64_t pairs, int K,
                 int dtype, void* y, const void* g, float* dt, float* partial, hipStream_t st) {
    if (!x || !mul || !bias || !mean || !std_p || pairs < 0) return set_error(TGT_ERR_INVALID, "gaussian basis: null argument");
    if (K <= 0 || K % 2 || K > 512) return set_error(TGT_ERR_UNSUPPORTED, "gaussian basis: K=%d must be even and <= 512", K);
    if (g ? (!dt || !partial) : !y) return set_error(TGT_ERR_INVALID, "gaussian basis: null output");
    if (pairs == 0) return TGT_OK;
    switch (dtype) {
        case TGT_F32: return gauss_launch<float>(x, mul, bias, mean, std_p, pairs, K, y, g, dt, partial, st);
        case TGT_BF16: return gauss_launch<bf16_t>(x, mul, bias, mean, std_p, pairs, K, y, g, dt, partial, st);
        case TGT_F16: return gauss_launch<f16_t>(x, mul, bias, mean, std_p, pairs, K, y, g, dt, partial, st);
        default: return set_error(TGT_ERR_INVALID, "gaussian basis: bad dtype %d", dtype);
    }
}

}  // namespace tgt
