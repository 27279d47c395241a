// Device side of the reference's MC-sampled prediction steps -- gfx950.
//
//   dist_bins_argmax   lib/training_schemes/pcqm/dist_pred/scheme.py:186-196  softmax -> p + p^T over the pair axes -> argmax
//   softmax_accumulate lib/training_schemes/pcqm/dist_pred/scheme.py:143-155  probs += softmax(logits)
//   probs_finish       lib/training_schemes/pcqm/dist_pred/scheme.py:164-166, :173  (p + p^T) / (2 valid) [, log(. + 1e-9)]
//   pack_triu          lib/data/pcqm/bin_ops.py:5-37 + dist_pred/scheme.py:221-226  strict upper triangle of the real nodes
//   bins_to_dist       lib/training_schemes/pcqm/commons.py:72-82 (+ unpack_bins_multi, bin_ops.py:39-46, as a mask)
//   gap_commit         lib/training_schemes/pcqm/gap_pred/scheme.py:88-96
//
// The reference asks the HOST after every stochastic forward whether the sample had a NaN/Inf (`.any()`: one device
// sync per sample, 50 samples per batch in the shipped configs) and only then decides where the sample goes.  Here the
// decision lives on the device: a 4-int state {valid, nonfinite, tries, _} travels with the loop; a sample is written
// into slot `valid` (or checked first, where it accumulates), and a one-thread commit kernel advances `valid` iff the
// sample was finite -- the same accept/skip sequence, no sync until the results are consumed.
//
// All of this is integer / byte / streaming work on (B,N,N,bins) logits (268 MB at B=512, 256 bins, fp16): HBM-bound,
// one read of the logits per sample.  Mapping of the row kernels: one 64-lane wave per logit row (or per unordered
// node pair: the symmetrised probabilities of (i,j) and (j,i) are the same sum), 8 contiguous bins = 16 B per lane.
#include "common.hpp"

namespace tgt {

enum { ST_VALID = 0, ST_NONFINITE = 1, ST_TRIES = 2 };

template <typename T>
__device__ __forceinline__ void pr_load8(const T* p, float (&v)[8]) {
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
__device__ __forceinline__ float pr_wave_max(float v) {
    return group_max<64>(v);
}
__device__ __forceinline__ float pr_wave_sum(float v) {
    return group_sum<64>(v);
}
__device__ __forceinline__ bool pr_finite(float x) { return fabsf(x) <= 3.402823466e+38f; }       // false for NaN and +-Inf

// one row of logits -> exp(x - max) per element (in v), returns 1 / sum; bad |= a non-finite logit
template <typename T, int VPL>
__device__ __forceinline__ float pr_softmax_row(const T* xr, int NB, int lane, float (&v)[VPL][8], bool& bad) {
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
        const int col = (k * 64 + lane) * 8;
        if (col < NB) {
            pr_load8(xr + col, v[k]);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                bad |= !pr_finite(v[k][i]);
                mx = fmaxf(mx, v[k][i]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[k][i] = -INFINITY;
        }
    }
    mx = pr_wave_max(mx);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < VPL; ++k)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v[k][i] = expf(v[k][i] - mx);
            s += v[k][i];
        }
    return 1.f / pr_wave_sum(s);
}

// bins[b, slot, i, j] = bins[b, slot, j, i] = argmax_c softmax(x[b,i,j,:])[c] + softmax(x[b,j,i,:])[c]   (first maximum)
template <typename T, typename OUT, int VPL>
__global__ void __launch_bounds__(256) dist_bins_kernel(const T* x, int64_t B, int N, int NB, OUT* bins, int64_t ld_b, int S,
                                                         int* state) {
#pragma clang fp contract(off)          // p1 + p2 as two rounded products and one add, on every path
    const int valid = state[ST_VALID];
    if (valid >= S) return;                                   // the loop already has its S samples: this try is not used
    OUT* out = bins + (int64_t)valid * N * N;
    const int lane = threadIdx.x & 63;
    const int64_t pairs = (int64_t)N * (N + 1) / 2, total = B * pairs;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    bool bad = false;
    for (int64_t w = wave; w < total; w += nwaves) {
        const int64_t b = w / pairs;
        int p = (int)(w - b * pairs), i = 0;
        while (p >= N - i) { p -= N - i; ++i; }               // row i of the upper triangle (with diagonal), j = i + p
        const int j = i + p;
        const T* r1 = x + ((b * N + i) * N + j) * (int64_t)NB;
        const T* r2 = x + ((b * N + j) * N + i) * (int64_t)NB;
        float v1[VPL][8], v2[VPL][8];
        const float inv1 = pr_softmax_row<T, VPL>(r1, NB, lane, v1, bad);
        const float inv2 = pr_softmax_row<T, VPL>(r2, NB, lane, v2, bad);
        float best = -1.f;
        int arg = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            const int col = (k * 64 + lane) * 8;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const float q1 = v1[k][t] * inv1, q2 = v2[k][t] * inv2;
                const float pr = q1 + q2;
                if (col + t < NB && pr > best) { best = pr; arg = col + t; }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oa = __shfl_xor(arg, o, 64);
            if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
        }
        if (lane == 0) {
            if (arg == 0x7fffffff) arg = 0;                    // an all-NaN row (the sample is discarded anyway)
            out[b * ld_b + (int64_t)i * N + j] = (OUT)arg;
            out[b * ld_b + (int64_t)j * N + i] = (OUT)arg;
        }
    }
    if (__any(bad) && lane == 0) atomicOr(&state[ST_NONFINITE], 1);
}

// state[NONFINITE] |= any non-finite element among n (n % 8 == 0)
template <typename T>
__global__ void __launch_bounds__(256) finite_check_kernel(const T* x, int64_t n, int* state) {
    bool bad = false;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += (int64_t)gridDim.x * 256 * 8) {
        float v[8];
        pr_load8(x + i, v);
#pragma unroll
        for (int t = 0; t < 8; ++t) bad |= !pr_finite(v[t]);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(&state[ST_NONFINITE], 1);
}

// acc[row][:] += softmax(x[row][:])  unless this sample was flagged or the loop is complete
template <typename T, int VPL>
__global__ void __launch_bounds__(256) softmax_accumulate_kernel(const T* x, int64_t rows, int NB, float* acc, const int* state, int S) {
    if (state[ST_NONFINITE] || state[ST_VALID] >= S) return;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t row = wave; row < rows; row += nwaves) {
        float v[VPL][8];
        bool bad = false;
        const float inv = pr_softmax_row<T, VPL>(x + row * NB, NB, lane, v, bad);
        float* ar = acc + row * NB;
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            const int col = (k * 64 + lane) * 8;
            if (col < NB) {
                float4 a0 = reinterpret_cast<float4*>(ar + col)[0], a1 = reinterpret_cast<float4*>(ar + col)[1];
                // softmax value rounded as torch stores it (one division), then the fp32 accumulation of the reference
                a0.x += v[k][0] * inv; a0.y += v[k][1] * inv; a0.z += v[k][2] * inv; a0.w += v[k][3] * inv;
                a1.x += v[k][4] * inv; a1.y += v[k][5] * inv; a1.z += v[k][6] * inv; a1.w += v[k][7] * inv;
                reinterpret_cast<float4*>(ar + col)[0] = a0;
                reinterpret_cast<float4*>(ar + col)[1] = a1;
            }
        }
    }
}

// out[b,i,j,c] = (acc[b,i,j,c] + acc[b,j,i,c]) / (2 valid)   [as_log: log(. + eps)]
__global__ void __launch_bounds__(256) probs_finish_kernel(const float* acc, int64_t B, int N, int NB, const int* state, int as_log,
                                                            float eps, float* out) {
    const float div = 2.f * (float)state[ST_VALID];
    const int64_t total = B * N * N * (int64_t)NB;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % NB);
        int64_t r = e / NB;
        const int j = (int)(r % N);
        r /= N;
        const int i = (int)(r % N);
        const int64_t b = r / N;
        float p = (acc[e] + acc[((b * N + j) * N + i) * (int64_t)NB + c]) / div;
        if (as_log) p = logf(p + eps);
        out[e] = p;
    }
}

// if the sample in flight was finite (and is still wanted) it becomes sample number `valid`
__global__ void sample_commit_kernel(int* state, int S) {
    if (!state[ST_NONFINITE] && state[ST_VALID] < S) state[ST_VALID] += 1;
    state[ST_NONFINITE] = 0;
    state[ST_TRIES] += 1;
}

// out[b, valid] = gap[b] and commit, or skip the sample when one of the B values is NaN/Inf
template <typename T>
__global__ void __launch_bounds__(1024) gap_commit_kernel(const T* gap, int B, float* out, int S, int* state) {
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    const int valid = state[ST_VALID];
    bool b_ = false;
    for (int b = threadIdx.x; b < B; b += blockDim.x) b_ |= !pr_finite(to_f32(gap[b]));
    if (b_) atomicOr(&bad, 1);
    __syncthreads();
    if (!bad && valid < S)
        for (int b = threadIdx.x; b < B; b += blockDim.x) out[(int64_t)b * S + valid] = to_f32(gap[b]);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (!bad && valid < S) state[ST_VALID] = valid + 1;
        state[ST_TRIES] += 1;
        state[ST_NONFINITE] = 0;
    }
}

// flat[offsets[b] + s*T_b + k] = bins[b, s, i, j], (i,j) the k-th pair of the strict upper triangle of the n_b real nodes
template <typename E>
__global__ void __launch_bounds__(256) pack_triu_kernel(const E* bins, int B, int S, int N, const int64_t* num_nodes,
                                                         const int64_t* offsets, E* flat) {
    const int64_t total = offsets[B];
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
        int lo = 0, hi = B - 1;                              // largest b with offsets[b] <= g
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (offsets[mid] <= g) lo = mid; else hi = mid - 1;
        }
        const int b = lo, n = (int)num_nodes[b];
        const int64_t T = (int64_t)n * (n - 1) / 2;
        const int64_t local = g - offsets[b];
        const int s = (int)(local / T);
        int k = (int)(local - (int64_t)s * T), i = 0;
        while (k >= n - 1 - i) { k -= n - 1 - i; ++i; }
        const int j = i + 1 + k;
        flat[g] = bins[(((int64_t)b * S + s) * N + i) * N + j];
    }
}

// dist[r,i,j] = ((u(i,j) + h) * bin_size) + ((u(j,i) + h) * bin_size), 0 on the diagonal when zero_diag,
// u(i,j) = bins[r,i,j], or -- num_nodes given -- bins[r,i,j] only for i < j < n_b, else 0 (what packing the bins and
// unpacking them into the zero-padded batch leaves, bin_ops.py:39-46 + stack_with_pad): r = b*S + s
template <typename E>
__global__ void __launch_bounds__(256) bins_to_dist_kernel(const E* bins, int64_t R, int N, const int64_t* num_nodes, int S,
                                                            float bin_size, float half, int zero_diag, float* out) {
#pragma clang fp contract(off)          // the reference's separate float32 multiply and add (HIP's __fmul_rn is a plain `*`)
    const int64_t total = R * N * N;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int j = (int)(e % N);
        int64_t r = e / N;
        const int i = (int)(r % N);
        r /= N;
        float u1 = (float)bins[e], u2 = (float)bins[(r * N + j) * N + i];
        if (num_nodes) {
            const int n = (int)num_nodes[r / S];
            if (!(i < j && j < n)) u1 = 0.f;
            if (!(j < i && i < n)) u2 = 0.f;
        }
        const float t1 = (u1 + half) * bin_size, t2 = (u2 + half) * bin_size;
        const float d = t1 + t2;
        out[e] = (zero_diag && i == j) ? 0.f : d;
    }
}

static int pr_grid(int64_t work, int per_block) {
    int64_t blocks = (work + per_block - 1) / per_block;
    return (int)(blocks < 1 ? 1 : (blocks > 16384 ? 16384 : blocks));
}

template <typename T, typename OUT>
static int dist_bins_vpl(const void* x, int64_t B, int N, int NB, void* bins, int64_t ld_b, int S, int* state, hipStream_t st) {
    const int grid = pr_grid(B * N * (N + 1) / 2, 4);
    const T* xp = reinterpret_cast<const T*>(x);
    OUT* bp = reinterpret_cast<OUT*>(bins);
    if (NB <= 512) hipLaunchKernelGGL((dist_bins_kernel<T, OUT, 1>), dim3(grid), dim3(256), 0, st, xp, B, N, NB, bp, ld_b, S, state);
    else if (NB <= 1024) hipLaunchKernelGGL((dist_bins_kernel<T, OUT, 2>), dim3(grid), dim3(256), 0, st, xp, B, N, NB, bp, ld_b, S, state);
    else hipLaunchKernelGGL((dist_bins_kernel<T, OUT, 4>), dim3(grid), dim3(256), 0, st, xp, B, N, NB, bp, ld_b, S, state);
    return check_launch("dist_bins_kernel");
}

template <typename T>
static int dist_bins_out(const void* x, int64_t B, int N, int NB, void* bins, int esz, int64_t ld_b, int S, int* state, hipStream_t st) {
    switch (esz) {
        case 1: return dist_bins_vpl<T, uint8_t>(x, B, N, NB, bins, ld_b, S, state, st);
        case 2: return dist_bins_vpl<T, uint16_t>(x, B, N, NB, bins, ld_b, S, state, st);
        case 4: return dist_bins_vpl<T, int32_t>(x, B, N, NB, bins, ld_b, S, state, st);
        default: return set_error(TGT_ERR_INVALID, "dist bins: element size %d (1, 2 or 4)", esz);
    }
}

static int pr_check_logits(const void* x, int NB, const char* what) {
    if (NB <= 0 || NB % 8 || NB > 2048) return set_error(TGT_ERR_UNSUPPORTED, "%s: %d bins must be a multiple of 8, <= 2048", what, NB);
    if ((uintptr_t)x % 16) return set_error(TGT_ERR_INVALID, "%s: logits must be 16-byte aligned", what);
    return TGT_OK;
}

int dist_bins_run(const void* x, int dtype, int64_t B, int N, int NB, void* bins, int esz, int64_t ld_b, int S, int* state,
                  hipStream_t st) {
    if (!x || !bins || !state || B < 0 || N <= 0 || S <= 0) return set_error(TGT_ERR_INVALID, "dist bins: bad argument");
    if (int rc = pr_check_logits(x, NB, "dist bins")) return rc;
    if ((esz == 1 && NB > 256) || (esz == 2 && NB > 65536)) return set_error(TGT_ERR_INVALID, "dist bins: %d bins do not fit %d-byte bins", NB, esz);
    if (B == 0) return TGT_OK;
    switch (dtype) {
        case TGT_F32: return dist_bins_out<float>(x, B, N, NB, bins, esz, ld_b, S, state, st);
        case TGT_BF16: return dist_bins_out<bf16_t>(x, B, N, NB, bins, esz, ld_b, S, state, st);
        case TGT_F16: return dist_bins_out<f16_t>(x, B, N, NB, bins, esz, ld_b, S, state, st);
        default: return set_error(TGT_ERR_INVALID, "dist bins: bad dtype %d", dtype);
    }
}

int sample_commit_run(int* state, int S, hipStream_t st) {
    if (!state || S <= 0) return set_error(TGT_ERR_INVALID, "sample commit: bad argument");
    hipLaunchKernelGGL(sample_commit_kernel, dim3(1), dim3(1), 0, st, state, S);
    return check_launch("sample_commit_kernel");
}

template <typename T>
static int softmax_acc_t(const void* x, int64_t rows, int NB, float* acc, int* state, int S, hipStream_t st) {
    const T* xp = reinterpret_cast<const T*>(x);
    hipLaunchKernelGGL((finite_check_kernel<T>), dim3(pr_grid(rows * NB, 2048)), dim3(256), 0, st, xp, rows * NB, state);
    const int grid = pr_grid(rows, 4);
    if (NB <= 512) hipLaunchKernelGGL((softmax_accumulate_kernel<T, 1>), dim3(grid), dim3(256), 0, st, xp, rows, NB, acc, state, S);
    else if (NB <= 1024) hipLaunchKernelGGL((softmax_accumulate_kernel<T, 2>), dim3(grid), dim3(256), 0, st, xp, rows, NB, acc, state, S);
    else hipLaunchKernelGGL((softmax_accumulate_kernel<T, 4>), dim3(grid), dim3(256), 0, st, xp, rows, NB, acc, state, S);
    return check_launch("softmax_accumulate_kernel");
}

int softmax_accumulate_run(const void* x, int dtype, int64_t rows, int NB, float* acc, int* state, int S, hipStream_t st) {
    if (!x || !acc || !state || rows < 0 || S <= 0) return set_error(TGT_ERR_INVALID, "softmax accumulate: bad argument");
    if (int rc = pr_check_logits(x, NB, "softmax accumulate")) return rc;
    if ((uintptr_t)acc % 16) return set_error(TGT_ERR_INVALID, "softmax accumulate: accumulator must be 16-byte aligned");
    if (rows == 0) return TGT_OK;
    switch (dtype) {
        case TGT_F32: return softmax_acc_t<float>(x, rows, NB, acc, state, S, st);
        case TGT_BF16: return softmax_acc_t<bf16_t>(x, rows, NB, acc, state, S, st);
        case TGT_F16: return softmax_acc_t<f16_t>(x, rows, NB, acc, state, S, st);
        default: return set_error(TGT_ERR_INVALID, "softmax accumulate: bad dtype %d", dtype);
    }
}

int probs_finish_run(const float* acc, int64_t B, int N, int NB, const int* state, int as_log, float eps, float* out, hipStream_t st) {
    if (!acc || !out || !state || B < 0 || N <= 0 || NB <= 0) return set_error(TGT_ERR_INVALID, "probs finish: bad argument");
    if (acc == out) return set_error(TGT_ERR_INVALID, "probs finish: in place is not possible (reads the transposed pair)");
    if (B == 0) return TGT_OK;
    hipLaunchKernelGGL(probs_finish_kernel, dim3(pr_grid(B * N * N * NB, 1024)), dim3(256), 0, st, acc, B, N, NB, state, as_log, eps, out);
    return check_launch("probs_finish_kernel");
}

int gap_commit_run(const void* gap, int dtype, int B, float* out, int S, int* state, hipStream_t st) {
    if (!gap || !out || !state || B <= 0 || S <= 0) return set_error(TGT_ERR_INVALID, "gap commit: bad argument");
    switch (dtype) {
        case TGT_F32: hipLaunchKernelGGL((gap_commit_kernel<float>), dim3(1), dim3(1024), 0, st, reinterpret_cast<const float*>(gap), B, out, S, state); break;
        case TGT_BF16: hipLaunchKernelGGL((gap_commit_kernel<bf16_t>), dim3(1), dim3(1024), 0, st, reinterpret_cast<const bf16_t*>(gap), B, out, S, state); break;
        case TGT_F16: hipLaunchKernelGGL((gap_commit_kernel<f16_t>), dim3(1), dim3(1024), 0, st, reinterpret_cast<const f16_t*>(gap), B, out, S, state); break;
        default: return set_error(TGT_ERR_INVALID, "gap commit: bad dtype %d", dtype);
    }
    return check_launch("gap_commit_kernel");
}

int pack_triu_run(const void* bins, int esz, int B, int S, int N, const int64_t* num_nodes, const int64_t* offsets, void* flat,
                  int64_t total, hipStream_t st) {
    if (!bins || !num_nodes || !offsets || B < 0 || S <= 0 || N <= 0 || total < 0) return set_error(TGT_ERR_INVALID, "pack triu: bad argument");
    if (B == 0 || total == 0) return TGT_OK;
    if (!flat) return set_error(TGT_ERR_INVALID, "pack triu: null output");
    const int grid = pr_grid(total, 256);
    switch (esz) {
        case 1: hipLaunchKernelGGL((pack_triu_kernel<uint8_t>), dim3(grid), dim3(256), 0, st, reinterpret_cast<const uint8_t*>(bins), B, S, N, num_nodes, offsets, reinterpret_cast<uint8_t*>(flat)); break;
        case 2: hipLaunchKernelGGL((pack_triu_kernel<uint16_t>), dim3(grid), dim3(256), 0, st, reinterpret_cast<const uint16_t*>(bins), B, S, N, num_nodes, offsets, reinterpret_cast<uint16_t*>(flat)); break;
        case 4: hipLaunchKernelGGL((pack_triu_kernel<int32_t>), dim3(grid), dim3(256), 0, st, reinterpret_cast<const int32_t*>(bins), B, S, N, num_nodes, offsets, reinterpret_cast<int32_t*>(flat)); break;
        default: return set_error(TGT_ERR_INVALID, "pack triu: element size %d (1, 2 or 4)", esz);
    }
    return check_launch("pack_triu_kernel");
}

int bins_to_dist_run(const void* bins, int kind, int64_t R, int N, const int64_t* num_nodes, int S, float bin_size, int shift_half,
                     int zero_diag, float* out, hipStream_t st) {
    if (!bins || !out || R < 0 || N <= 0 || (num_nodes && S <= 0)) return set_error(TGT_ERR_INVALID, "bins to dist: bad argument");
    if (R == 0) return TGT_OK;
    const int grid = pr_grid(R * N * N, 256);
    const float half = shift_half ? 0.5f : 0.f;
#define TGT_B2D(E) hipLaunchKernelGGL((bins_to_dist_kernel<E>), dim3(grid), dim3(256), 0, st, reinterpret_cast<const E*>(bins), R, N, \
                                      num_nodes, S > 0 ? S : 1, bin_size, half, zero_diag, out)
    switch (kind) {
        case TGT_BINS_U8: TGT_B2D(uint8_t); break;
        case TGT_BINS_U16: TGT_B2D(uint16_t); break;
        case TGT_BINS_I32: TGT_B2D(int32_t); break;
        case TGT_BINS_I64: TGT_B2D(int64_t); break;
        case TGT_BINS_F32: TGT_B2D(float); break;
        default: return set_error(TGT_ERR_INVALID, "bins to dist: bad element kind %d", kind);
    }
#undef TGT_B2D
    return check_launch("bins_to_dist_kernel");
}

}  // namespace tgt
