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

// CS (backward only): also the column sums of `out` seen as (n / cols, cols) rows -- the bias gradient of the Linear in front of
// the activation -- as one partial row per workgroup in cs_partial (gridDim.x, cols).  A thread's vectors all start at the same
// column (the grid stride is a multiple of cols: host-checked), so it keeps 16/sizeof(T) float accumulators; the 256 / (cols/V)
// threads of a workgroup that share a column group are folded through LDS.
template <typename T, bool BWD, bool CS = false>
__global__ void __launch_bounds__(256) gelu_dropout_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                          T* __restrict__ out, int64_t n, uint64_t seed,
                                                          uint32_t thresh, float inv_keep, const float* __restrict__ row_scale,
                                                          int64_t elems_per_sample, int cols = 0, float* __restrict__ cs_partial = nullptr,
                                                          const uint64_t* __restrict__ seed_ctr = nullptr) {
    constexpr int V = 16 / (int)sizeof(T);
    seed = step_seed(seed, seed_ctr);
    float csum[CS ? V : 1];
    if constexpr (CS)
        for (int t = 0; t < V; ++t) csum[t] = 0.f;
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
        // per-sample factor (DropPath of the residual branch this activation feeds, folded in here so that the
        // branch's Linear + residual add need no scaled copy of their gradient); a vector never straddles samples
        // (32-bit division in vector units: a 64-bit one per vector costs as much as the rest of the loop body)
        const float ik = row_scale ? inv_keep * row_scale[(uint32_t)(i / V) / (uint32_t)(elems_per_sample / V)] : inv_keep;
#pragma unroll
        for (int t = 0; t < V; ++t) {
            const float v = to_f32(xv[t]);
            float e;
            const float cdf = gelu_cdf(v, e);
            float r;
            if (!BWD) r = v * cdf;
            else r = to_f32(gv[t]) * (cdf + v * 0.3989422804014327f * e);
            ov[t] = from_f32<T>((thresh == 0u || keep[t]) ? r * ik : 0.f);
            if constexpr (CS) csum[t] += (i + t < n) ? to_f32(ov[t]) : 0.f;        // of the value as stored
        }
        if (i + V <= n) {
            uint4 raw;
            __builtin_memcpy(&raw, ov, 16);
            *reinterpret_cast<uint4*>(out + i) = raw;
        } else {
            for (int t = 0; t < V && i + t < n; ++t) out[i + t] = ov[t];
        }
    }
    if constexpr (CS) {
        __shared__ float fold[256 * V];
#pragma unroll
        for (int t = 0; t < V; ++t) fold[threadIdx.x * V + t] = csum[t];
        __syncthreads();
        const int groups = cols / V;                         // threads per row of vectors; 256 % groups == 0 (host-checked)
        // thread tid starts at column ((blockIdx*256 + tid) * V) % cols = ((tid % groups) * V): (256 * V) % cols == 0
        for (int c = threadIdx.x; c < cols; c += 256) {
            const int gidx = c / V, e = c % V;
            float v = 0.f;
            for (int t = gidx; t < 256; t += groups) v += fold[t * V + e];
            cs_partial[(int64_t)blockIdx.x * cols + c] = v;
        }
    }
}

template <typename T>
static int gd_launch(const void* x, const void* dy, void* out, int64_t n, float p, uint64_t seed, bool bwd,
                     const float* row_scale, int64_t eps_, hipStream_t st) {
    const uint32_t thresh = p <= 0.f ? 0u : (uint32_t)fmin(65535.0, fmax(1.0, nearbyint((double)p * 65536.0)));   // 16-bit
    const float inv_keep = p <= 0.f ? 1.f : 1.f / (1.f - p);
    constexpr int V = 16 / (int)sizeof(T);
    int64_t blocks = (n / V + 255) / 256;
    // one 16-byte vector per thread, no revisits: +5 % over a 4096-workgroup grid-stride grid (a plain
    // copy shows the same: tools/probes/hbm_probe.hip)
    if (blocks < 1) blocks = 1;
    if (!bwd)
        hipLaunchKernelGGL((gelu_dropout_kernel<T, false>), dim3((unsigned)blocks), dim3(256), 0, st,
                           reinterpret_cast<const T*>(x), nullptr, reinterpret_cast<T*>(out), n, seed, thresh, inv_keep, row_scale, eps_, 0,
                           (float*)nullptr, seed_counter());
    else
        hipLaunchKernelGGL((gelu_dropout_kernel<T, true>), dim3((unsigned)blocks), dim3(256), 0, st,
                           reinterpret_cast<const T*>(x), reinterpret_cast<const T*>(dy), reinterpret_cast<T*>(out), n,
                           seed, thresh, inv_keep, row_scale, eps_, 0, (float*)nullptr, seed_counter());
    return check_launch(bwd ? "gelu_dropout_bwd_kernel" : "gelu_dropout_fwd_kernel");
}

int sum_rows_run(const float* x, int rows, int C, float* out, hipStream_t st);      // layernorm.hip: fixed-order sum over rows

static constexpr int kGdColsumParts = 4096;        // workgroups (= partial rows) of the column-sum variant: a grid-stride grid
int gelu_colsum_parts() { return kGdColsumParts; }

template <typename T>
static int gd_colsum_launch(const void* x, const void* dy, void* out, int64_t n, float p, uint64_t seed, const float* row_scale,
                            int64_t eps_, int cols, float* partial, float* colsum, hipStream_t st) {
    const uint32_t thresh = p <= 0.f ? 0u : (uint32_t)fmin(65535.0, fmax(1.0, nearbyint((double)p * 65536.0)));
    const float inv_keep = p <= 0.f ? 1.f : 1.f / (1.f - p);
    constexpr int V = 16 / (int)sizeof(T);
    if (cols % V || cols < V || (256 * V) % cols || n % cols)
        return set_error(TGT_ERR_UNSUPPORTED, "gelu_dropout colsum: cols=%d must divide %d and n", cols, 256 * V);
    int64_t blocks = (n / V + 255) / 256;
    if (blocks > kGdColsumParts) blocks = kGdColsumParts;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((gelu_dropout_kernel<T, true, true>), dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<const T*>(x),
                       reinterpret_cast<const T*>(dy), reinterpret_cast<T*>(out), n, seed, thresh, inv_keep, row_scale, eps_, cols, partial,
                       seed_counter());
    if (int e = check_launch("gelu_dropout_bwd_colsum_kernel")) return e;
    return sum_rows_run(partial, (int)blocks, cols, colsum, st);
}

int gelu_dropout_bwd_colsum_run(const void* x, const void* dy, void* out, int64_t n, int dtype, float p, uint64_t seed,
                                const float* row_scale, int64_t elems_per_sample, int cols, float* partial, float* colsum,
                                hipStream_t st) {
    if (!x || !out || !dy || !partial || !colsum || n <= 0 || cols <= 0) return set_error(TGT_ERR_INVALID, "gelu_dropout colsum: bad argument");
    if (p < 0.f || p >= 1.f) return set_error(TGT_ERR_INVALID, "gelu_dropout: p=%f outside [0,1)", p);
    if (((uintptr_t)x | (uintptr_t)out | (uintptr_t)dy) % 16) return set_error(TGT_ERR_INVALID, "gelu_dropout: tensors must be 16-byte aligned");
    if (row_scale && (elems_per_sample <= 0 || elems_per_sample % 8 || n / 4 > 0xffffffffLL))
        return set_error(TGT_ERR_INVALID, "gelu_dropout: row_scale needs elems_per_sample, a multiple of 8 (and n < 2^34)");
    switch (dtype) {
        case TGT_F32: return gd_colsum_launch<float>(x, dy, out, n, p, seed, row_scale, elems_per_sample, cols, partial, colsum, st);
        case TGT_BF16: return gd_colsum_launch<bf16_t>(x, dy, out, n, p, seed, row_scale, elems_per_sample, cols, partial, colsum, st);
        case TGT_F16: return gd_colsum_launch<f16_t>(x, dy, out, n, p, seed, row_scale, elems_per_sample, cols, partial, colsum, st);
        default: return set_error(TGT_ERR_INVALID, "gelu_dropout: bad dtype %d", dtype);
    }
}

int gelu_dropout_run(const void* x, const void* dy, void* out, int64_t n, int dtype, float p, uint64_t seed, bool bwd,
                     const float* row_scale, int64_t elems_per_sample, hipStream_t st) {
    if (!x || !out || n < 0 || (bwd && !dy)) return set_error(TGT_ERR_INVALID, "gelu_dropout: null tensor");
    if (p < 0.f || p >= 1.f) return set_error(TGT_ERR_INVALID, "gelu_dropout: p=%f outside [0,1)", p);
    if (((uintptr_t)x | (uintptr_t)out | (uintptr_t)dy) % 16) return set_error(TGT_ERR_INVALID, "gelu_dropout: tensors must be 16-byte aligned");
    if (n == 0) return TGT_OK;
    if (row_scale && (elems_per_sample <= 0 || elems_per_sample % 8 || n / 4 > 0xffffffffLL))
        return set_error(TGT_ERR_INVALID, "gelu_dropout: row_scale needs elems_per_sample, a multiple of 8 (and n < 2^34)");
    switch (dtype) {
        case TGT_F32: return gd_launch<float>(x, dy, out, n, p, seed, bwd, row_scale, elems_per_sample, st);
        case TGT_BF16: return gd_launch<bf16_t>(x, dy, out, n, p, seed, bwd, row_scale, elems_per_sample, st);
        case TGT_F16: return gd_launch<f16_t>(x, dy, out, n, p, seed, bwd, row_scale, elems_per_sample, st);
        default: return set_error(TGT_ERR_INVALID, "gelu_dropout: bad dtype %d", dtype);
    }
}

}  // namespace tgt
