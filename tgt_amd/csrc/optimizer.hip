// Optimizer side of the training step on flat float32 buffers -- gfx950.
//
// Reference: lib/training/training.py:439-470 (training_step: GradScaler.unscale_, clip_grad_value_,
// clip_grad_norm_, optimizer.step, GradScaler.update), :159-171 (apex FusedAdam),
// lib/training_schemes/pcqm/tgt_training.py:141-171 (update_losses).  The reference does these with
// per-tensor launches and host round trips (GradScaler's found_inf `.item()`, two scalar all-reduces +
// `.item()` per step).  Here the whole decision chain stays on the device in a 16-float control block
// (`ctl`, layout in include/tgt_hip.h): one pass over the flat gradient computes its norm and a
// non-finite flag, a single thread applies the GradScaler / clipping rules, and the Adam kernel reads
// its gradient multiplier, step count and skip flag from the block.  No host sync anywhere.
#include "common.hpp"

namespace tgt {

enum { CTL_SCALE = 0, CTL_TRACKER = 1, CTL_FOUND_INF = 2, CTL_STEPS = 3, CTL_MULT = 4, CTL_COEF = 5, CTL_NORM = 6,
       CTL_SKIPPED = 7, CTL_LOSS = 8, CTL_SAMPLES = 9, CTL_NAN = 10, CTL_LOSS_LO = 11, CTL_PAIR = 12, CTL_SAMPLES_LO = 14, CTL_LR = 15 };

// running sums of update_losses as float32 PAIRS (value, low-order part): an error-free two-sum per step, so that the sample
// count stays exact far beyond 2^24 and the loss sum keeps ~48 bits (a plain float32 sum stops counting samples at 16.7 M)
__device__ __forceinline__ void two_sum_into(float& hi, float& lo, float x) {
    const float s = hi + x;
    const float bb = s - hi;
    const float err = (hi - (s - bb)) + (x - bb);
    hi = s;
    lo += err;
}

template <typename S>
__device__ __forceinline__ void shadow_store4(void* shadow, int64_t i, const float* P) {
    S t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = from_f32<S>(P[k]);
    uint2 raw;
    __builtin_memcpy(&raw, t, 8);
    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(shadow) + i) = raw;
}

__device__ __forceinline__ float clip_val(float g, float cv) { return cv > 0.f ? fminf(fmaxf(g, -cv), cv) : g; }

// Adam over flat float32 buffers: 4 reads + 3 writes per element (+ the 16-bit shadow), HBM-bound.
// ctl == NULL: bias corrections / gradient multiplier from the host arguments.
// ctl != NULL: skip when ctl[FOUND_INF] != 0; multiplier = ctl[MULT]*ctl[COEF]; t = ctl[STEPS].
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   float lr, float b1, float b2, float eps, float wd,
                                                   float bc1, float bc2_rsqrt, float gscale, float clip_value,
                                                   const float* __restrict__ ctl, void* shadow, int shadow_dtype) {
    float mult = gscale, coef = 1.f;
    if (ctl) {
        if (ctl[CTL_FOUND_INF] != 0.f) return;
        if (lr < 0.f) lr = ctl[CTL_LR];          // (a captured step: the schedule's value is written into the block before each replay)
        mult = ctl[CTL_MULT];
        coef = ctl[CTL_COEF];
        const double t = (double)ctl[CTL_STEPS];
        bc1 = (float)(1.0 - pow((double)b1, t));
        bc2_rsqrt = (float)(1.0 / sqrt(1.0 - pow((double)b2, t)));
    }
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 4 <= n) {
            float4 pp = *reinterpret_cast<float4*>(p + i), gg = *reinterpret_cast<const float4*>(g + i);
            float4 mm = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
            float* P = &pp.x; float* Gp = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float gr = clip_val(Gp[t] * mult, clip_value) * coef;
                M[t] = b1 * M[t] + (1.f - b1) * gr;
                V[t] = b2 * V[t] + (1.f - b2) * gr * gr;
                const float denom = sqrtf(V[t]) * bc2_rsqrt + eps;
                P[t] = P[t] * (1.f - lr * wd) - (lr / bc1) * (M[t] / denom);
            }
            *reinterpret_cast<float4*>(p + i) = pp;
            *reinterpret_cast<float4*>(m + i) = mm;
            *reinterpret_cast<float4*>(v + i) = vv;
            if (shadow) {
                if (shadow_dtype == TGT_BF16) shadow_store4<bf16_t>(shadow, i, P);
                else shadow_store4<f16_t>(shadow, i, P);
            }
        } else {
            for (int64_t k = i; k < n; ++k) {
                const float gr = clip_val(g[k] * mult, clip_value) * coef;
                m[k] = b1 * m[k] + (1.f - b1) * gr;
                v[k] = b2 * v[k] + (1.f - b2) * gr * gr;
                const float denom = sqrtf(v[k]) * bc2_rsqrt + eps;
                p[k] = p[k] * (1.f - lr * wd) - (lr / bc1) * (m[k] / denom);
                if (shadow) {
                    if (shadow_dtype == TGT_BF16) reinterpret_cast<bf16_t*>(shadow)[k] = from_f32<bf16_t>(p[k]);
                    else reinterpret_cast<f16_t*>(shadow)[k] = from_f32<f16_t>(p[k]);
                }
            }
        }
    }
}

// ---- gradient statistics: sum of squares of clip(g * mult) and a non-finite count, fixed order --------
constexpr int kStatBlocks = 1024;

__global__ void __launch_bounds__(256) grad_stats_kernel(const float* __restrict__ g, int64_t n, const float* __restrict__ ctl,
                                                         float world, float clip_value, float* __restrict__ partial) {
    __shared__ float red[2][4];
    const float mult = 1.f / (ctl[CTL_SCALE] * world);
    float ss = 0.f, bad = 0.f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        if (i + 4 <= n) {
            const float4 gg = *reinterpret_cast<const float4*>(g + i);
            t[0] = gg.x; t[1] = gg.y; t[2] = gg.z; t[3] = gg.w;
        } else {
            for (int64_t k = i; k < n; ++k) t[k - i] = g[k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float x = t[k] * mult;
            bad += (x - x != 0.f) ? 1.f : 0.f;               // inf or nan
            const float c = clip_val(x, clip_value);
            ss += c * c;
        }
    }
    ss = group_sum<64>(ss);
    bad = group_sum<64>(bad);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wave] = ss; red[1][wave] = bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        partial[kStatBlocks + blockIdx.x] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

// one workgroup: fold the partials, then thread 0 applies torch.cuda.amp.GradScaler's rules
// (unscale -> found_inf -> skip / step; update: backoff on inf, growth after `growth_interval` clean
// steps) and nn.utils.clip_grad_norm_'s coefficient  max_norm / (norm + 1e-6), clamped to 1.
__global__ void __launch_bounds__(256) scaler_update_kernel(const float* __restrict__ partial, int nparts, float* __restrict__ ctl,
                                                            float world, float clip_norm, int dynamic, float growth_factor,
                                                            float backoff_factor, int growth_interval) {
    __shared__ double red[2][256];
    double ss = 0.0, bad = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) {
        ss += partial[i];
        bad += partial[kStatBlocks + i];
    }
    red[0][threadIdx.x] = ss;
    red[1][threadIdx.x] = bad;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            red[0][threadIdx.x] += red[0][threadIdx.x + o];
            red[1][threadIdx.x] += red[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    const float norm = (float)sqrt(red[0][0]);
    const bool nonfinite = red[1][0] != 0.0;
    float S = ctl[CTL_SCALE];
    ctl[CTL_MULT] = 1.f / (S * world);
    ctl[CTL_NORM] = norm;
    // torch's clip_grad_norm_ (reference training.py:461): clamp(max_norm / (norm + 1e-6), max = 1) -- a NaN norm gives a NaN
    // coefficient that poisons every gradient, an infinite norm gives 0; fminf alone would DROP the NaN and apply the step
    ctl[CTL_COEF] = clip_norm > 0.f ? (norm != norm ? norm : fminf(1.f, clip_norm / (norm + 1e-6f))) : 1.f;
    const bool skip = dynamic && nonfinite;
    ctl[CTL_FOUND_INF] = skip ? 1.f : 0.f;
    if (skip) {
        ctl[CTL_SKIPPED] += 1.f;
        ctl[CTL_SCALE] = S * backoff_factor;
        ctl[CTL_TRACKER] = 0.f;
    } else {
        ctl[CTL_STEPS] += 1.f;
        if (dynamic) {
            const float tr = ctl[CTL_TRACKER] + 1.f;
            if (tr >= (float)growth_interval) {
                ctl[CTL_SCALE] = S * growth_factor;
                ctl[CTL_TRACKER] = 0.f;
            } else {
                ctl[CTL_TRACKER] = tr;
            }
        }
    }
}

// ---- update_losses (tgt_training.py:141-171) without `.item()` -----------------------------------
// mode 1: pair <- (loss*samples, samples); mode 2: accumulate pair into ctl; mode 3: both (one rank)
__global__ void loss_accumulate_kernel(const void* loss, int loss_f64, float samples, float* pair, float* ctl, int mixed,
                                       int mode) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (mode & 1) {
        const float l = loss_f64 ? (float)*reinterpret_cast<const double*>(loss) : *reinterpret_cast<const float*>(loss);
        pair[0] = l * samples;
        pair[1] = samples;
    }
    if (mode & 2) {
        const float sl = pair[0], ss = pair[1];
        if (mixed) {                                   // a NaN loss (overflowed fp16 step) is skipped, unless 10 in a row
            if (sl == sl || ctl[CTL_NAN] >= 10.f) {
                ctl[CTL_NAN] = 0.f;
                two_sum_into(ctl[CTL_LOSS], ctl[CTL_LOSS_LO], sl);
                two_sum_into(ctl[CTL_SAMPLES], ctl[CTL_SAMPLES_LO], ss);
            } else {
                ctl[CTL_NAN] += 1.f;
            }
        } else {
            two_sum_into(ctl[CTL_LOSS], ctl[CTL_LOSS_LO], sl);
            two_sum_into(ctl[CTL_SAMPLES], ctl[CTL_SAMPLES_LO], ss);
        }
    }
}

}  // namespace tgt

using namespace tgt;

extern "C" {

int tgt_grad_stats_parts(void) { return 2 * kStatBlocks; }

int tgt_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                  float beta1, float beta2, float eps, float weight_decay, int32_t step, float grad_scale,
                  float clip_value, const float* ctl, void* shadow, int32_t shadow_dtype, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n < 0 || (!ctl && step < 1))
        return set_error(TGT_ERR_INVALID, "adam: null buffer or bad n/step");
    if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16)
        return set_error(TGT_ERR_INVALID, "adam: buffers must be 16-byte aligned");
    if (shadow && shadow_dtype != TGT_BF16 && shadow_dtype != TGT_F16)
        return set_error(TGT_ERR_INVALID, "adam: shadow dtype must be bf16 or f16");
    if (shadow && ((uintptr_t)shadow % 8)) return set_error(TGT_ERR_INVALID, "adam: shadow must be 8-byte aligned");
    if (n == 0) return TGT_OK;
    const int t = step < 1 ? 1 : step;
    const double bc1 = 1.0 - pow((double)beta1, t), bc2 = 1.0 - pow((double)beta2, t);
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), param,
                       grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, (float)bc1,
                       (float)(1.0 / sqrt(bc2)), grad_scale, clip_value, ctl, shadow, shadow_dtype);
    return check_launch("adam_kernel");
}

int tgt_grad_scaler_step(const float* grad, int64_t n, float* ctl, float* partial, int32_t world, float clip_value,
                         float clip_norm, int32_t dynamic, float growth_factor, float backoff_factor,
                         int32_t growth_interval, void* stream) {
    if (!grad || !ctl || !partial || n < 0 || world < 1)
        return set_error(TGT_ERR_INVALID, "grad scaler: null buffer or bad n/world");
    if ((uintptr_t)grad % 16) return set_error(TGT_ERR_INVALID, "grad scaler: grad must be 16-byte aligned");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > kStatBlocks) blocks = kStatBlocks;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(grad_stats_kernel, dim3((unsigned)blocks), dim3(256), 0, st, grad, n, ctl, (float)world, clip_value,
                       partial);
    hipLaunchKernelGGL(scaler_update_kernel, dim3(1), dim3(256), 0, st, partial, (int)blocks, ctl, (float)world, clip_norm,
                       dynamic, growth_factor, backoff_factor, growth_interval);
    return check_launch("grad_scaler kernels");
}

int tgt_loss_accumulate(const void* loss, int32_t loss_is_f64, float samples, float* ctl, int32_t mixed, int32_t mode,
                        void* stream) {
    if (!ctl || ((mode & 1) && !loss) || mode < 1 || mode > 3) return set_error(TGT_ERR_INVALID, "loss accumulate: bad arguments");
    hipLaunchKernelGGL(loss_accumulate_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), loss, loss_is_f64,
                       samples, ctl + CTL_PAIR, ctl, mixed, mode);
    return check_launch("loss_accumulate_kernel");
}

}  // extern "C"
