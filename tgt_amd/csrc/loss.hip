// Row-wise cross entropy of the binned-distance head (reference
// lib/training_schemes/pcqm/commons.py:36-41: F.cross_entropy(logits.view(-1, num_bins), bins,
// reduction='none') under a pair mask), forward and backward, on the logits as the model stores them.
//
// (B,N,N,512) logits are 268 MB in bf16 at the BASELINE batch.  ATen's path under autocast casts
// them to fp32 (0.54 GB), runs log_softmax + nll_loss, and walks the same chain back: ~1.1 ms of
// HBM passes per step.  Here: one read for the forward (row log-sum-exp + the target's logit), one
// read + one write for the backward (softmax recomputed from the saved log-sum-exp), fp32 math on
// the stored values, no fp32 image of the logits.
//
// Mapping: one 64-lane wave per row chunk of 512 classes (8 contiguous classes = 16 B per lane and
// vector), VPL vectors per lane for wider heads; row max / sum by wave reductions.
#include "common.hpp"

namespace tgt {

template <typename T>
__device__ __forceinline__ void xe_load8(const T* p, float (&v)[8]) {
    if constexpr (sizeof(T) == 4) {
        const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
        const uint4 raw = *reinterpret_cast<const uint4*>(p);
        T t[8];
        __builtin_memcpy(t, &raw, 16);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = to_f32(t[i]);
    }
}
template <typename T>
__device__ __forceinline__ void xe_store8(T* p, const float (&v)[8]) {
    if constexpr (sizeof(T) == 4) {
        reinterpret_cast<float4*>(p)[0] = make_float4(v[0], v[1], v[2], v[3]);
        reinterpret_cast<float4*>(p)[1] = make_float4(v[4], v[5], v[6], v[7]);
    } else {
        T t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = from_f32<T>(v[i]);
        uint4 raw;
        __builtin_memcpy(&raw, t, 16);
        *reinterpret_cast<uint4*>(p) = raw;
    }
}
__device__ __forceinline__ float wave_max(float v) {
    return group_max<64>(v);
}
__device__ __forceinline__ float wave_sum(float v) {
    return group_sum<64>(v);
}

// forward: lse[row] = log sum_c exp(x[row][c]);  xent[row] = lse[row] - x[row][target[row]]
// (rows with a target outside [0, C) get xent = 0: F.cross_entropy's ignore_index behaviour is not
// needed by the reference, which clamps its bins)
template <typename T, int VPL>
__global__ void __launch_bounds__(256) xent_fwd_kernel(const T* x, const int64_t* target, int64_t rows, int C, float* lse,
                                                       float* xent) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t row = wave; row < rows; row += nwaves) {
        const T* xr = x + row * C;
        float v[VPL][8];
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            const int col = (k * 64 + lane) * 8;
            if (col < C) {
                xe_load8(xr + col, v[k]);
#pragma unroll
                for (int i = 0; i < 8; ++i) mx = fmaxf(mx, v[k][i]);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[k][i] = -INFINITY;
            }
        }
        mx = wave_max(mx);
        const int64_t t = target[row];
        float s = 0.f, xt = 0.f;
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            const int col = (k * 64 + lane) * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                s += expf(v[k][i] - mx);
                if (col + i == t) xt = v[k][i];
            }
        }
        s = wave_sum(s);
        xt = wave_sum(xt);               // exactly one lane holds it
        if (lane == 0) {
            const float l = mx + logf(s);
            lse[row] = l;
            xent[row] = (t >= 0 && t < C) ? l - xt : 0.f;
        }
    }
}

// backward: dx[row][c] = w[row] * (exp(x[row][c] - lse[row]) - [c == target[row]])
template <typename T, int VPL>
__global__ void __launch_bounds__(256) xent_bwd_kernel(const T* x, const int64_t* target, const float* lse, const float* w,
                                                       int64_t rows, int C, T* dx) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t row = wave; row < rows; row += nwaves) {
        const T* xr = x + row * C;
        T* dr = dx + row * C;
        const float l = lse[row], wr = w[row];
        const int64_t t = target[row];
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            const int col = (k * 64 + lane) * 8;
            if (col < C) {
                float v[8];
                if (wr != 0.f) {
                    xe_load8(xr + col, v);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = wr * (expf(v[i] - l) - (col + i == t ? 1.f : 0.f));
                } else {            // masked pair: exact zeros, and no read
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = 0.f;
                }
                xe_store8(dr + col, v);
            }
        }
    }
}

template <typename T, int VPL>
static int xent_launch(const void* x, const int64_t* target, const float* lse_in, const float* w, int64_t rows, int C,
                       float* lse, float* xent, void* dx, hipStream_t st) {
    int64_t blocks = (rows + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    if (!dx) {
        hipLaunchKernelGGL((xent_fwd_kernel<T, VPL>), dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<const T*>(x),
                           target, rows, C, lse, xent);
        return check_launch("xent_fwd_kernel");
    }
    hipLaunchKernelGGL((xent_bwd_kernel<T, VPL>), dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<const T*>(x), target,
                       lse_in, w, rows, C, reinterpret_cast<T*>(dx));
    return check_launch("xent_bwd_kernel");
}

template <typename T>
static int xent_vpl(const void* x, const int64_t* target, const float* lse_in, const float* w, int64_t rows, int C, float* lse,
                    float* xent, void* dx, hipStream_t st) {
    if (C <= 512) return xent_launch<T, 1>(x, target, lse_in, w, rows, C, lse, xent, dx, st);
    if (C <= 1024) return xent_launch<T, 2>(x, target, lse_in, w, rows, C, lse, xent, dx, st);
    return xent_launch<T, 4>(x, target, lse_in, w, rows, C, lse, xent, dx, st);
}

int xent_run(const void* x, int dtype, const int64_t* target, const float* lse_in, const float* w, int64_t rows, int C,
             float* lse, float* xent, void* dx, hipStream_t st) {
    const bool bwd = dx != nullptr;
    if (!x || !target || rows < 0 || (bwd ? (!lse_in || !w) : (!lse || !xent)))
        return set_error(TGT_ERR_INVALID, "cross entropy: null tensor");
    if (C <= 0 || C % 8 || C > 2048) return set_error(TGT_ERR_UNSUPPORTED, "cross entropy: C=%d must be a multiple of 8, <= 2048", C);
    if (((uintptr_t)x | (uintptr_t)dx) % 16) return set_error(TGT_ERR_INVALID, "cross entropy: logits must be 16-byte aligned");
    if (rows == 0) return TGT_OK;
    switch (dtype) {
        case TGT_F32: return xent_vpl<float>(x, target, lse_in, w, rows, C, lse, xent, dx, st);
        case TGT_BF16: return xent_vpl<bf16_t>(x, target, lse_in, w, rows, C, lse, xent, dx, st);
        case TGT_F16: return xent_vpl<f16_t>(x, target, lse_in, w, rows, C, lse, xent, dx, st);
        default: return set_error(TGT_ERR_INVALID, "cross entropy: bad dtype %d", dtype);
    }
}

}  // namespace tgt
