// Assembling the kernel-order projection parameters of a triplet module for gfx950.
//
// The reference keeps lin_QKV_in / lin_EG_in / lin_QKV_out / lin_EG_out as four nn.Linear
// (lib/tgt/layers/triplet.py:198-202) with head-MINOR output channels; the kernels consume ONE
// fused projection whose rows are head-major (include/tgt_hip.h).  Instead of a dozen
// gather / cat / cast launches per layer and step, one launch gathers the rows of up to 8 source
// matrices (+ their bias vectors) into the fused matrix, casting on the way, and one launch
// scatters the fused gradient back to per-parameter gradients.  Pure data movement (a few
// hundred KB); the point is the launch count.
#include "common.hpp"

namespace tgt {

template <typename S, typename D>
__global__ void __launch_bounds__(128) fuse_rows_kernel(const tgt_fuse_rows_args a, bool scatter) {
    const int r = blockIdx.x;
    const int sid = a.row_src[r], sr = a.row_idx[r];
    // gather: fused (type D) <- source (type S);  scatter: source (type S) <- fused (type D)
    D* frow = reinterpret_cast<D*>(a.fused) + (int64_t)r * a.n_cols;
    if (sid < 0) {
        if (!scatter) {
            for (int c = threadIdx.x; c < a.n_cols; c += blockDim.x) frow[c] = from_f32<D>(0.f);
            if (a.fused_bias && threadIdx.x == 0) reinterpret_cast<D*>(a.fused_bias)[r] = from_f32<D>(0.f);
        }
        return;
    }
    S* srow = reinterpret_cast<S*>(a.src[sid]) + (int64_t)sr * a.n_cols;
    if (!scatter) {
        for (int c = threadIdx.x; c < a.n_cols; c += blockDim.x) frow[c] = from_f32<D>(to_f32(srow[c]));
        if (a.fused_bias && threadIdx.x == 0)
            reinterpret_cast<D*>(a.fused_bias)[r] = from_f32<D>(to_f32(reinterpret_cast<const S*>(a.src_bias[sid])[sr]));
    } else {
        for (int c = threadIdx.x; c < a.n_cols; c += blockDim.x) srow[c] = from_f32<S>(to_f32(frow[c]));
        if (a.fused_bias && threadIdx.x == 0)
            reinterpret_cast<S*>(a.src_bias[sid])[sr] = from_f32<S>(to_f32(reinterpret_cast<const D*>(a.fused_bias)[r]));
    }
}

// dst[r][c] = src[r][idx[c]]  (lin_O's input columns in the kernel's [dir][h][d] order; the
// gradient goes back through the same kernel with the inverse index)
template <typename S, typename D>
__global__ void __launch_bounds__(256) permute_cols_kernel(const S* src, const int32_t* idx, D* dst, int rows, int cols) {
    const int64_t n = (int64_t)rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols), c = (int)(i % cols);
        dst[i] = from_f32<D>(to_f32(src[(int64_t)r * cols + idx[c]]));
    }
}

template <typename S>
static int fuse_rows_d(const tgt_fuse_rows_args& a, bool scatter, hipStream_t st) {
    switch (a.fused_dtype) {
        case TGT_F32: hipLaunchKernelGGL((fuse_rows_kernel<S, float>), dim3(a.n_rows), dim3(128), 0, st, a, scatter); break;
        case TGT_BF16: hipLaunchKernelGGL((fuse_rows_kernel<S, bf16_t>), dim3(a.n_rows), dim3(128), 0, st, a, scatter); break;
        case TGT_F16: hipLaunchKernelGGL((fuse_rows_kernel<S, f16_t>), dim3(a.n_rows), dim3(128), 0, st, a, scatter); break;
        default: return set_error(TGT_ERR_INVALID, "fuse_rows: bad fused dtype %d", a.fused_dtype);
    }
    return check_launch("fuse_rows_kernel");
}

int fuse_rows_run(const tgt_fuse_rows_args* a, bool scatter, hipStream_t st) {
    if (!a) return set_error(TGT_ERR_INVALID, "fuse_rows: null args");
    if (a->n_rows < 0 || a->n_cols <= 0 || a->n_src <= 0 || a->n_src > 8)
        return set_error(TGT_ERR_INVALID, "fuse_rows: bad sizes rows=%d cols=%d sources=%d", a->n_rows, a->n_cols, a->n_src);
    if (!a->row_src || !a->row_idx || !a->fused) return set_error(TGT_ERR_INVALID, "fuse_rows: null tensor");
    for (int i = 0; i < a->n_src; ++i)
        if (!a->src[i] || (a->fused_bias && !a->src_bias[i])) return set_error(TGT_ERR_INVALID, "fuse_rows: null source %d", i);
    if (a->n_rows == 0) return TGT_OK;
    switch (a->src_dtype) {
        case TGT_F32: return fuse_rows_d<float>(*a, scatter, st);
        case TGT_BF16: return fuse_rows_d<bf16_t>(*a, scatter, st);
        case TGT_F16: return fuse_rows_d<f16_t>(*a, scatter, st);
        default: return set_error(TGT_ERR_INVALID, "fuse_rows: bad source dtype %d", a->src_dtype);
    }
}

template <typename S>
static int permute_cols_d(const void* src, const int32_t* idx, void* dst, int dd, int rows, int cols, hipStream_t st) {
    const int64_t n = (int64_t)rows * cols;
    const unsigned grid = (unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
    const S* s = reinterpret_cast<const S*>(src);
    switch (dd) {
        case TGT_F32: hipLaunchKernelGGL((permute_cols_kernel<S, float>), dim3(grid), dim3(256), 0, st, s, idx, reinterpret_cast<float*>(dst), rows, cols); break;
        case TGT_BF16: hipLaunchKernelGGL((permute_cols_kernel<S, bf16_t>), dim3(grid), dim3(256), 0, st, s, idx, reinterpret_cast<bf16_t*>(dst), rows, cols); break;
        case TGT_F16: hipLaunchKernelGGL((permute_cols_kernel<S, f16_t>), dim3(grid), dim3(256), 0, st, s, idx, reinterpret_cast<f16_t*>(dst), rows, cols); break;
        default: return set_error(TGT_ERR_INVALID, "permute_cols: bad dst dtype %d", dd);
    }
    return check_launch("permute_cols_kernel");
}

int permute_cols_run(const void* src, int sd, const int32_t* idx, void* dst, int dd, int rows, int cols, hipStream_t st) {
    if (!src || !idx || !dst || rows < 0 || cols <= 0) return set_error(TGT_ERR_INVALID, "permute_cols: bad arguments");
    if (rows == 0) return TGT_OK;
    switch (sd) {
        case TGT_F32: return permute_cols_d<float>(src, idx, dst, dd, rows, cols, st);
        case TGT_BF16: return permute_cols_d<bf16_t>(src, idx, dst, dd, rows, cols, st);
        case TGT_F16: return permute_cols_d<f16_t>(src, idx, dst, dd, rows, cols, st);
        default: return set_error(TGT_ERR_INVALID, "permute_cols: bad src dtype %d", sd);
    }
}

// ---------------------------------------------------------------------------
// out[i] = sum_p x[p*n + i]: the last stage of the weight gradients (dW = sum over row chunks of
// dY_c^T X_c, ops._wgrad_into).  Block = 32 float4 columns x 8 plane slices; a thread sums its slice's
// planes (4 loads in flight), the 8 slices meet in LDS -- fixed order, so bit-reproducible.  ATen's
// generic reduction takes 10-40 us on these (8..128 planes of 16K..590K elements); this is HBM/L2-bound.
// ---------------------------------------------------------------------------
template <int V>
struct PlaneVec;
template <>
struct PlaneVec<4> { using type = float4; };
template <>
struct PlaneVec<1> { using type = float; };
__device__ __forceinline__ void vadd(float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
__device__ __forceinline__ void vadd(float& a, const float& b) { a += b; }

template <int V>
__global__ void __launch_bounds__(256) sum_planes_kernel(const float* x, int planes, int64_t nv, float* out) {
    using VT = typename PlaneVec<V>::type;
    __shared__ VT red[8][32];
    const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int64_t col = (int64_t)blockIdx.x * 32 + cl;
    const VT* xv = reinterpret_cast<const VT*>(x);
    VT s0 = {}, s1 = {}, s2 = {}, s3 = {};
    if (col < nv) {
        int p = sl;
        for (; p + 24 < planes; p += 32) {
            const VT a = xv[(int64_t)p * nv + col], b = xv[(int64_t)(p + 8) * nv + col];
            const VT c = xv[(int64_t)(p + 16) * nv + col], d = xv[(int64_t)(p + 24) * nv + col];
            vadd(s0, a); vadd(s1, b); vadd(s2, c); vadd(s3, d);
        }
        for (; p < planes; p += 8) vadd(s0, xv[(int64_t)p * nv + col]);
        vadd(s0, s1); vadd(s2, s3); vadd(s0, s2);
    }
    red[sl][cl] = s0;
    __syncthreads();
    if (sl == 0 && col < nv) {
        VT t = red[0][cl];
#pragma unroll
        for (int k = 1; k < 8; ++k) vadd(t, red[k][cl]);
        reinterpret_cast<VT*>(out)[col] = t;
    }
}

int sum_planes_run(const float* x, int planes, int64_t n, float* out, hipStream_t st) {
    if (!x || !out || planes <= 0 || n < 0) return set_error(TGT_ERR_INVALID, "sum_planes: bad arguments");
    if (n == 0) return TGT_OK;
    if (n % 4 == 0 && ((uintptr_t)x | (uintptr_t)out) % 16 == 0) {
        const int64_t nv = n / 4;
        hipLaunchKernelGGL((sum_planes_kernel<4>), dim3((unsigned)((nv + 31) / 32)), dim3(256), 0, st, x, planes, nv, out);
    } else {
        hipLaunchKernelGGL((sum_planes_kernel<1>), dim3((unsigned)((n + 31) / 32)), dim3(256), 0, st, x, planes, n, out);
    }
    return check_launch("sum_planes_kernel");
}

// ---------------------------------------------------------------------------
// Many plane sums in ONE launch (round 4, the backward's launch diet): the closing fixed-order sums behind the split-M weight
// gradients and the column-sum partials of a layer (13 + 8 launches of 4-9 us per backward layer, each with its dispatch gap)
// are collected by the host (ops.DeferredSums) and run as one grid.  The descriptors travel as KERNEL ARGUMENTS (the pointers
// change every step; no staging copy), each workgroup finds its item in the block prefix and then does exactly what
// sum_planes_kernel does for it -- same slices, same order, bit-identical results.
// ---------------------------------------------------------------------------
constexpr int kSumManyMax = 64;
struct SumManyArgs {
    const float* src[kSumManyMax];
    float* dst[kSumManyMax];
    int64_t nv[kSumManyMax];             // elements (vec = 1) or float4 groups (vec = 4) per plane
    int32_t planes[kSumManyMax];
    int32_t first_block[kSumManyMax + 1];
    uint64_t vec4;                       // bit i: item i is summed as float4
    int32_t n;
};
template <int V>
__device__ __forceinline__ void sum_planes_block(const float* x, int planes, int64_t nv, float* out, int block, void* lds) {
    using VT = typename PlaneVec<V>::type;
    VT (*red)[32] = reinterpret_cast<VT (*)[32]>(lds);
    const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int64_t col = (int64_t)block * 32 + cl;
    const VT* xv = reinterpret_cast<const VT*>(x);
    VT s0 = {}, s1 = {}, s2 = {}, s3 = {};
    if (col < nv) {
        int p = sl;
        for (; p + 24 < planes; p += 32) {
            const VT a = xv[(int64_t)p * nv + col], b = xv[(int64_t)(p + 8) * nv + col];
            const VT c = xv[(int64_t)(p + 16) * nv + col], d = xv[(int64_t)(p + 24) * nv + col];
            vadd(s0, a); vadd(s1, b); vadd(s2, c); vadd(s3, d);
        }
        for (; p < planes; p += 8) vadd(s0, xv[(int64_t)p * nv + col]);
        vadd(s0, s1); vadd(s2, s3); vadd(s0, s2);
    }
    red[sl][cl] = s0;
    __syncthreads();
    if (sl == 0 && col < nv) {
        VT t = red[0][cl];
#pragma unroll
        for (int k = 1; k < 8; ++k) vadd(t, red[k][cl]);
        reinterpret_cast<VT*>(out)[col] = t;
    }
}
__global__ void __launch_bounds__(256) sum_many_kernel(const SumManyArgs a) {
    __shared__ float4 red[8][32];
    int it = 0;
    while (it + 1 < a.n && (int)blockIdx.x >= a.first_block[it + 1]) ++it;          // (scalar: <= 64 steps)
    const int block = (int)blockIdx.x - a.first_block[it];
    if ((a.vec4 >> it) & 1) sum_planes_block<4>(a.src[it], a.planes[it], a.nv[it], a.dst[it], block, red);
    else sum_planes_block<1>(a.src[it], a.planes[it], a.nv[it], a.dst[it], block, red);
}
struct SumItem { const float* src; float* dst; int32_t planes; int32_t _pad; int64_t n; };
int sum_many_run(const void* items_host, int n, hipStream_t st) {
    static_assert(sizeof(SumItem) == 32, "tgt_sum_item layout");
    if (n < 0 || (n > 0 && !items_host)) return set_error(TGT_ERR_INVALID, "sum_many: bad arguments");
    if (n > kSumManyMax) return set_error(TGT_ERR_UNSUPPORTED, "sum_many: at most %d sums per launch", kSumManyMax);
    const SumItem* items = reinterpret_cast<const SumItem*>(items_host);
    SumManyArgs a = {};
    int blocks = 0, m = 0;
    for (int i = 0; i < n; ++i) {
        const SumItem& s = items[i];
        if (!s.src || !s.dst || s.planes <= 0 || s.n < 0) return set_error(TGT_ERR_INVALID, "sum_many: bad item %d", i);
        if (s.n == 0) continue;
        const bool v4 = s.n % 4 == 0 && ((uintptr_t)s.src | (uintptr_t)s.dst) % 16 == 0;          // (the rule of sum_planes_run)
        a.src[m] = s.src; a.dst[m] = s.dst; a.planes[m] = s.planes;
        a.nv[m] = v4 ? s.n / 4 : s.n;
        if (v4) a.vec4 |= (uint64_t)1 << m;
        a.first_block[m] = blocks;
        const int64_t nb = (a.nv[m] + 31) / 32;
        if (nb > 0x3fffffff - blocks) return set_error(TGT_ERR_UNSUPPORTED, "sum_many: too many elements");
        blocks += (int)nb;
        ++m;
    }
    if (m == 0) return TGT_OK;
    a.first_block[m] = blocks;
    a.n = m;
    hipLaunchKernelGGL(sum_many_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
    return check_launch("sum_many_kernel");
}

// ---------------------------------------------------------------------------
// Transposes of many small 16-bit matrices in ONE launch: the data-gradient kernels of csrc/edge_gemm.hip take the weight as
// (N, K) = W^T of the nn.Linear they differentiate.  Per step that was one 4 us copy kernel per Linear (plus its dispatch gap)
// in front of 96 backward launches; the weights only change in the optimizer step, so the trainer refreshes all of them at once.
// items: device array of {src, dst, rows, cols}; src (rows, cols) row-major contiguous, dst (cols, rows).
// ---------------------------------------------------------------------------
struct TransposeItem { const uint16_t* src; uint16_t* dst; int32_t rows, cols; };
__global__ void __launch_bounds__(256) transpose_many_kernel(const TransposeItem* items) {
    __shared__ uint16_t tile[32][33];
    const TransposeItem it = items[blockIdx.y];
    const int tc = (it.cols + 31) / 32, tr = (it.rows + 31) / 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int t = blockIdx.x; t < tc * tr; t += gridDim.x) {
        const int r0 = (t / tc) * 32, c0 = (t % tc) * 32;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r0 + ty + 8 * k, c = c0 + tx;
            if (r < it.rows && c < it.cols) tile[ty + 8 * k][tx] = it.src[(int64_t)r * it.cols + c];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c0 + ty + 8 * k, r = r0 + tx;
            if (r < it.rows && c < it.cols) it.dst[(int64_t)c * it.rows + r] = tile[tx][ty + 8 * k];
        }
        __syncthreads();
    }
}
int transpose_many_run(const void* items, int n, int blocks_per_item, hipStream_t st) {
    if (n < 0 || (n > 0 && !items) || blocks_per_item <= 0) return set_error(TGT_ERR_INVALID, "transpose_many: bad arguments");
    if (n == 0) return TGT_OK;
    if (n > 65535) return set_error(TGT_ERR_UNSUPPORTED, "transpose_many: more than 65535 matrices");
    static_assert(sizeof(TransposeItem) == 24, "tgt_transpose_item layout");
    hipLaunchKernelGGL(transpose_many_kernel, dim3((unsigned)blocks_per_item, (unsigned)n), dim3(256), 0, st,
                       reinterpret_cast<const TransposeItem*>(items));
    return check_launch("transpose_many_kernel");
}

}  // namespace tgt
