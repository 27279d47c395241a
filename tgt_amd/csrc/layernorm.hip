// LayerNorm over the channel axis, forward and backward, for gfx950.
//
// The TGT layer opens every sub-block with a LayerNorm over the (B,N,N,C=256)
// edge tensor or the (B,N,W=768) node tensor (reference lib/tgt/layers/layers.py:37-38,
// :150, triplet.py:195): five per layer.  It is pure HBM streaming, so the kernel
// reads the row in its storage type (bf16/fp16/fp32), keeps it in registers for
// the two statistics passes, and writes the result directly in the type the
// consuming GEMM wants -- no fp32 round trip through HBM.
//   forward : y = (x - mean) * rstd * gamma + beta          (stats in fp32, saved)
//   backward: g = dy*gamma; dx = rstd*(g - mean(g) - xhat*mean(g*xhat))
//             dgamma = sum_rows dy*xhat, dbeta = sum_rows dy
// Mapping: LPR lanes share one row (8 contiguous channels = 16 B per lane and
// vector), 64/LPR rows per wave, shuffle reductions inside the LPR-lane group.
// dgamma/dbeta: each lane owns fixed channels and accumulates over the rows it
// visits (grid-stride); a workgroup folds its partials through LDS and writes
// ONE row of a (parts, 2C) fp32 buffer; a second tiny kernel sums the parts in
// a fixed order (deterministic, no atomics).
#include <cstdlib>
#include "common.hpp"

namespace tgt {

constexpr int kLnParts = 1536;

__device__ __forceinline__ void ln_load8(const void* p, int dtype, int64_t idx, float (&v)[8]) {
    if (dtype == TGT_F32) {
        const float4* q = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + idx);
        float4 a = q[0], b = q[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
        uint4 raw = ld16_stream<TGT_NT_LNLOAD != 0>(reinterpret_cast<const uint16_t*>(p) + idx);
        if (dtype == TGT_BF16) {
            bf16_t t[8];
            __builtin_memcpy(t, &raw, 16);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = to_f32(t[i]);
        } else {
            f16_t t[8];
            __builtin_memcpy(t, &raw, 16);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = to_f32(t[i]);
        }
    }
}

__device__ __forceinline__ void ln_store8(void* p, int dtype, int64_t idx, const float (&v)[8]) {
    if (dtype == TGT_F32) {
        float4* q = reinterpret_cast<float4*>(reinterpret_cast<float*>(p) + idx);
        q[0] = make_float4(v[0], v[1], v[2], v[3]);
        q[1] = make_float4(v[4], v[5], v[6], v[7]);
    } else {
        uint4 raw;
        if (dtype == TGT_BF16) {
            bf16_t t[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] = from_f32<bf16_t>(v[i]);
            __builtin_memcpy(&raw, t, 16);
        } else {
            f16_t t[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] = from_f32<f16_t>(v[i]);
            __builtin_memcpy(&raw, t, 16);
        }
        st16_stream<TGT_NT_LN != 0>(reinterpret_cast<uint16_t*>(p) + idx, raw);
    }
}

// (group_sum<LPR>: common.hpp -- DPP adds inside a 16-lane row, v_permlane16_swap / v_permlane32_swap across; no ds_bpermute)

struct LnArgs {
    const void* x; const void* dy; void* y; void* dx;
    const float* gamma; const float* beta;
    float* mean; float* rstd; float* partial;
    int64_t rows; int C; float eps;
    int x_dtype, y_dtype, dy_dtype, dx_dtype;
    // fused residual entry (all optional):
    //   forward : s = res + x * scale[row / rows_per_sample]  -> s_out (x_dtype); y = LN(s)
    //   backward: d_total = ds_in + LN_bwd(dy)  -> dx;  dx2 = d_total * scale[...]
    const void* res; int res_dtype; void* s_out;
    const float* scale; int64_t rows_per_sample;
    const void* ds_in; int ds_dtype; void* dx2;
    float* x_colsum;     // backward, optional: column sums of the x-branch gradient
};

template <int LPR, int VPL>
__global__ void __launch_bounds__(256) ln_fwd_kernel(const LnArgs a) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, sub = lane / LPR, gl = lane % LPR;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    const float invC = 1.f / a.C;
    float gam[VPL][8], bet[VPL][8];
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        const int col = (v * LPR + gl) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            gam[v][i] = col < a.C ? a.gamma[col + i] : 0.f;
            bet[v][i] = col < a.C ? a.beta[col + i] : 0.f;
        }
    }
    for (int64_t row = wave * RPW + sub; row < a.rows; row += nwaves * RPW) {
        float x[VPL][8];
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            const int col = (v * LPR + gl) * 8;
            if (col < a.C) {
                ln_load8(a.x, a.x_dtype, row * a.C + col, x[v]);
                if (a.res) {
                    float rr[8];
                    ln_load8(a.res, a.res_dtype, row * a.C + col, rr);
                    const float sc = a.scale ? a.scale[(uint32_t)row / (uint32_t)a.rows_per_sample] : 1.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) x[v][i] = rr[i] + x[v][i] * sc;
                    ln_store8(a.s_out, a.x_dtype, row * a.C + col, x[v]);
                    // LN sees the stream value as stored (rounded to its storage dtype)
                    if (a.x_dtype == TGT_BF16) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) x[v][i] = to_f32(from_f32<bf16_t>(x[v][i]));
                    } else if (a.x_dtype == TGT_F16) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) x[v][i] = to_f32(from_f32<f16_t>(x[v][i]));
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) x[v][i] = 0.f;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) s += x[v][i];
        }
        const float mean = group_sum<LPR>(s) * invC;
        float q = 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            const int col = (v * LPR + gl) * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float d = col < a.C ? x[v][i] - mean : 0.f;
                q += d * d;
            }
        }
        const float rstd = rsqrtf(group_sum<LPR>(q) * invC + a.eps);
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            const int col = (v * LPR + gl) * 8;
            if (col < a.C) {
                float y[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) y[i] = (x[v][i] - mean) * rstd * gam[v][i] + bet[v][i];
                ln_store8(a.y, a.y_dtype, row * a.C + col, y);
            }
        }
        if (gl == 0) {
            a.mean[row] = mean;
            a.rstd[row] = rstd;
        }
    }
}

// CS: also accumulate the column sums of the x-branch gradient (d_total * scale) -- the bias
// gradient of the Linear that produced x -- as a third column plane of the partial buffer.
template <int LPR, int VPL, bool CS>
__global__ void __launch_bounds__(256, VPL == 1 ? 5 : 1) ln_bwd_kernel(const LnArgs a) {
    constexpr int RPW = 64 / LPR, NP = CS ? 3 : 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);        // [4 waves][RPW][NP][VPL*LPR*8]
    const int lane = threadIdx.x & 63, sub = lane / LPR, gl = lane % LPR, w = threadIdx.x >> 6;
    const int64_t wave = (int64_t)blockIdx.x * 4 + w, nwaves = (int64_t)gridDim.x * 4;
    const float invC = 1.f / a.C;
    float gam[VPL][8], dgam[VPL][8], dbet[VPL][8], dxs[CS ? VPL : 1][8];
    // CS, one vector per lane: the column sums of the x-branch gradient accumulate in LDS (this thread's own 32 bytes, plain
    // read-modify-write) instead of 8 more registers -- the kernel then fits 96 registers = 5 waves per SIMD, and occupancy
    // is what this load-latency-bound pass runs on (plain backward 105 -> 81 us at 5 waves)
    constexpr bool kLdsAcc = CS && VPL == 1;
    float4* accx = reinterpret_cast<float4*>(smem) + 4 * threadIdx.x;      // [column sums of dx | dbeta]; ALIASES `red` (used after the walk only)
    if constexpr (kLdsAcc) accx[0] = accx[1] = accx[2] = accx[3] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        const int col = (v * LPR + gl) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            gam[v][i] = col < a.C ? a.gamma[col + i] : 0.f;
            dgam[v][i] = dbet[v][i] = 0.f;
            if constexpr (CS) dxs[v][i] = 0.f;
        }
    }
    for (int64_t row = wave * RPW + sub; row < a.rows; row += nwaves * RPW) {
        const float mean = a.mean[row], rstd = a.rstd[row];
        float xh[VPL][8], g[VPL][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            const int col = (v * LPR + gl) * 8;
            if (col < a.C) {
                float dy[8];
                ln_load8(a.x, a.x_dtype, row * a.C + col, xh[v]);
                ln_load8(a.dy, a.dy_dtype, row * a.C + col, dy);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    xh[v][i] = (xh[v][i] - mean) * rstd;
                    g[v][i] = dy[i] * gam[v][i];
                    dgam[v][i] += dy[i] * xh[v][i];
                    if constexpr (!kLdsAcc) dbet[v][i] += dy[i];
                    s1 += g[v][i];
                    s2 += g[v][i] * xh[v][i];
                }
                if constexpr (kLdsAcc) {
                    float4 b0 = accx[2], b1 = accx[3];
                    b0.x += dy[0]; b0.y += dy[1]; b0.z += dy[2]; b0.w += dy[3];
                    b1.x += dy[4]; b1.y += dy[5]; b1.z += dy[6]; b1.w += dy[7];
                    accx[2] = b0; accx[3] = b1;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) xh[v][i] = g[v][i] = 0.f;
            }
        }
        const float c1 = group_sum<LPR>(s1) * invC, c2 = group_sum<LPR>(s2) * invC;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            const int col = (v * LPR + gl) * 8;
            if (col < a.C) {
                float dx[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) dx[i] = rstd * (g[v][i] - c1 - xh[v][i] * c2);
                if (a.ds_in) {
                    float dd[8];
                    ln_load8(a.ds_in, a.ds_dtype, row * a.C + col, dd);
#pragma unroll
                    for (int i = 0; i < 8; ++i) dx[i] += dd[i];
                }
                ln_store8(a.dx, a.dx_dtype, row * a.C + col, dx);
                if (a.dx2 || a.scale) {
                    // (scale without dx2: the x branch was pre-scaled by its producer -- its gradient IS d_total, only the
                    //  bias gradient of the Linear in between carries the factor: column sums of d_total * scale)
                    const float sc = a.scale ? a.scale[(uint32_t)row / (uint32_t)a.rows_per_sample] : 1.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) dx[i] *= sc;
                    if (a.dx2) ln_store8(a.dx2, a.dx_dtype, row * a.C + col, dx);
                }
                if constexpr (kLdsAcc) {
                    float4 a0 = accx[0], a1 = accx[1];
                    a0.x += dx[0]; a0.y += dx[1]; a0.z += dx[2]; a0.w += dx[3];
                    a1.x += dx[4]; a1.y += dx[5]; a1.z += dx[6]; a1.w += dx[7];
                    accx[0] = a0; accx[1] = a1;
                } else if constexpr (CS) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) dxs[v][i] += dx[i];
                }
            }
        }
    }
    if constexpr (kLdsAcc) {
        const float4 a0 = accx[0], a1 = accx[1];
        dxs[0][0] = a0.x; dxs[0][1] = a0.y; dxs[0][2] = a0.z; dxs[0][3] = a0.w;
        dxs[0][4] = a1.x; dxs[0][5] = a1.y; dxs[0][6] = a1.z; dxs[0][7] = a1.w;
        const float4 b0 = accx[2], b1 = accx[3];
        dbet[0][0] = b0.x; dbet[0][1] = b0.y; dbet[0][2] = b0.z; dbet[0][3] = b0.w;
        dbet[0][4] = b1.x; dbet[0][5] = b1.y; dbet[0][6] = b1.z; dbet[0][7] = b1.w;
        __syncthreads();          // every thread has its sums in registers before `red` overwrites the accumulators
    }
    // fold the 4*RPW row-groups of this workgroup, fixed order
    constexpr int CW = VPL * LPR * 8;
#pragma unroll
    for (int v = 0; v < VPL; ++v)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int col = (v * LPR + gl) * 8 + i;
            red[((w * RPW + sub) * NP + 0) * CW + col] = dgam[v][i];
            red[((w * RPW + sub) * NP + 1) * CW + col] = dbet[v][i];
            if constexpr (CS) red[((w * RPW + sub) * NP + 2) * CW + col] = dxs[v][i];
        }
    __syncthreads();
    for (int idx = threadIdx.x; idx < NP * a.C; idx += 256) {
        const int which = idx / a.C, col = idx % a.C;
        float s = 0.f;
        for (int gidx = 0; gidx < 4 * RPW; ++gidx) s += red[(gidx * NP + which) * CW + col];
        a.partial[(int64_t)blockIdx.x * NP * a.C + idx] = s;
    }
}

// sum a (parts, W) fp32 partial buffer over parts: block = 8 columns x 32 part-slices, fixed order.
// out0 gets columns [0,split), out1 columns [split,W).
__global__ void __launch_bounds__(256) partial_sum_kernel(const float* partial, int parts, int W, int split, float* out0,
                                                          float* out1) {
    __shared__ float red[32][9];
    const int cl = threadIdx.x & 7, sl = threadIdx.x >> 3;
    const int idx = blockIdx.x * 8 + cl;
    float s = 0.f;
    if (idx < W) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int p = sl;
        for (; p + 96 < parts; p += 128) {
            s0 += partial[(int64_t)p * W + idx];
            s1 += partial[(int64_t)(p + 32) * W + idx];
            s2 += partial[(int64_t)(p + 64) * W + idx];
            s3 += partial[(int64_t)(p + 96) * W + idx];
        }
        for (; p < parts; p += 32) s0 += partial[(int64_t)p * W + idx];
        s = (s0 + s1) + (s2 + s3);
    }
    red[sl][cl] = s;
    __syncthreads();
    if (sl == 0 && idx < W) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) t += red[k][cl];
        if (idx < split) out0[idx] = t;
        else out1[idx - split] = t;
    }
}

// column sums of a (rows, C) tensor (bias gradients of the Linear layers): same lane<->column
// ownership and two-stage deterministic reduction as the LayerNorm backward.
template <int LPR, int VPL>
__global__ void __launch_bounds__(256) colsum_kernel(const void* x, int x_dtype, int64_t rows, int C, float* partial) {
    constexpr int RPW = 64 / LPR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);        // [4 waves][RPW][VPL*LPR*8]
    const int lane = threadIdx.x & 63, sub = lane / LPR, gl = lane % LPR, w = threadIdx.x >> 6;
    const int64_t wave = (int64_t)blockIdx.x * 4 + w, nwaves = (int64_t)gridDim.x * 4;
    float acc[VPL][8];
#pragma unroll
    for (int v = 0; v < VPL; ++v)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[v][i] = 0.f;
    for (int64_t row = wave * RPW + sub; row < rows; row += nwaves * RPW) {
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            const int col = (v * LPR + gl) * 8;
            if (col < C) {
                float t[8];
                ln_load8(x, x_dtype, row * C + col, t);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[v][i] += t[i];
            }
        }
    }
    constexpr int CW = VPL * LPR * 8;
#pragma unroll
    for (int v = 0; v < VPL; ++v)
#pragma unroll
        for (int i = 0; i < 8; ++i) red[(w * RPW + sub) * CW + (v * LPR + gl) * 8 + i] = acc[v][i];
    __syncthreads();
    for (int col = threadIdx.x; col < C; col += 256) {
        float s = 0.f;
        for (int g = 0; g < 4 * RPW; ++g) s += red[g * CW + col];
        partial[(int64_t)blockIdx.x * C + col] = s;
    }
}

template <int LPR, int VPL>
static int ln_launch(const LnArgs& a, bool bwd, float* dgamma, float* dbeta, hipStream_t st) {
    constexpr int RPW = 64 / LPR;
    int64_t blocks = (a.rows + 4 * RPW - 1) / (4 * RPW);
    if (!bwd) {
        // grid-stride over at most 4096 workgroups: swept 1024..32768 on MI355X (tools/kernel_bench.py --only ln);
        // smaller grids lose parallelism, larger ones pay the per-workgroup gamma/beta loads (uncapped: 4x slower)
        constexpr int64_t cap = 4096;
        if (blocks > cap) blocks = cap;
        hipLaunchKernelGGL((ln_fwd_kernel<LPR, VPL>), dim3((unsigned)blocks), dim3(256), 0, st, a);
        return check_launch("ln_fwd_kernel");
    }
    // grid = rows of the partial buffer that get written.  One vector per lane: 96 registers = 5 waves per SIMD, 1280 workgroups of
    // 4 waves are exactly one round on 256 CUs (measured: 1024 and 1536 are both slower)
    const int parts = VPL == 1 ? 1280 : 1024;
    const bool cs = a.x_colsum != nullptr;     // then dbeta and x_colsum are ONE buffer [dbeta | x_colsum] (checked by the caller)
    const int np = cs ? 3 : 2;
    size_t lds = (size_t)4 * RPW * np * VPL * LPR * 8 * sizeof(float);
    if (cs && VPL == 1 && lds < 256 * 64) lds = 256 * 64;          // the in-LDS accumulators alias the fold buffer
    if (cs) hipLaunchKernelGGL((ln_bwd_kernel<LPR, VPL, true>), dim3(parts), dim3(256), lds, st, a);
    else hipLaunchKernelGGL((ln_bwd_kernel<LPR, VPL, false>), dim3(parts), dim3(256), lds, st, a);
    if (int e = check_launch("ln_bwd_kernel")) return e;
    hipLaunchKernelGGL(partial_sum_kernel, dim3((np * a.C + 7) / 8), dim3(256), 0, st, a.partial, parts, np * a.C, a.C,
                       dgamma, dbeta);
    return check_launch("partial_sum_kernel");
}

static int ln_dispatch(const LnArgs& a, bool bwd, float* dgamma, float* dbeta, hipStream_t st) {
    if (a.C % 8 || a.C <= 0 || a.C > 2048) return set_error(TGT_ERR_UNSUPPORTED, "layer norm: C=%d must be a multiple of 8, <= 2048", a.C);
    const int v8 = a.C / 8;
    if (v8 <= 4) return ln_launch<4, 1>(a, bwd, dgamma, dbeta, st);
    if (v8 <= 8) return ln_launch<8, 1>(a, bwd, dgamma, dbeta, st);
    if (v8 <= 16) return ln_launch<16, 1>(a, bwd, dgamma, dbeta, st);
    if (v8 <= 32) return ln_launch<32, 1>(a, bwd, dgamma, dbeta, st);
    if (v8 <= 64) return ln_launch<64, 1>(a, bwd, dgamma, dbeta, st);
    if (v8 <= 128) return ln_launch<64, 2>(a, bwd, dgamma, dbeta, st);
    return ln_launch<64, 4>(a, bwd, dgamma, dbeta, st);
}

static bool bad_dtype(int d) { return d != TGT_F32 && d != TGT_BF16 && d != TGT_F16; }

int layer_norm_fwd_run(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, int y_dtype,
                       float* mean, float* rstd, int64_t rows, int C, float eps, hipStream_t st) {
    if (!x || !gamma || !beta || !y || !mean || !rstd || rows < 0) return set_error(TGT_ERR_INVALID, "layer norm fwd: null tensor");
    if (bad_dtype(x_dtype) || bad_dtype(y_dtype)) return set_error(TGT_ERR_INVALID, "layer norm fwd: bad dtype");
    if (((uintptr_t)x | (uintptr_t)y) % 16) return set_error(TGT_ERR_INVALID, "layer norm fwd: x/y must be 16-byte aligned");
    if (rows == 0) return TGT_OK;
    LnArgs a = {};
    a.x = x; a.y = y; a.gamma = gamma; a.beta = beta; a.mean = mean; a.rstd = rstd;
    a.rows = rows; a.C = C; a.eps = eps; a.x_dtype = x_dtype; a.y_dtype = y_dtype;
    return ln_dispatch(a, false, nullptr, nullptr, st);
}

int layer_norm_bwd_run(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* gamma, const float* mean,
                       const float* rstd, void* dx, int dx_dtype, float* dgamma, float* dbeta, float* partial,
                       int64_t rows, int C, hipStream_t st) {
    if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || !partial || rows < 0)
        return set_error(TGT_ERR_INVALID, "layer norm bwd: null tensor");
    if (bad_dtype(x_dtype) || bad_dtype(dy_dtype) || bad_dtype(dx_dtype)) return set_error(TGT_ERR_INVALID, "layer norm bwd: bad dtype");
    if (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) % 16) return set_error(TGT_ERR_INVALID, "layer norm bwd: tensors must be 16-byte aligned");
    LnArgs a = {};
    a.x = x; a.dy = dy; a.dx = dx; a.gamma = gamma; a.mean = const_cast<float*>(mean); a.rstd = const_cast<float*>(rstd);
    a.partial = partial; a.rows = rows; a.C = C; a.x_dtype = x_dtype; a.dy_dtype = dy_dtype; a.dx_dtype = dx_dtype;
    return ln_dispatch(a, true, dgamma, dbeta, st);
}

int add_layer_norm_fwd_run(const void* x, int x_dtype, const void* res, int res_dtype, const float* scale,
                           int64_t rows_per_sample, void* s_out, const float* gamma, const float* beta, void* y,
                           int y_dtype, float* mean, float* rstd, int64_t rows, int C, float eps, hipStream_t st) {
    if (!x || !res || !s_out || !gamma || !beta || !y || !mean || !rstd || rows < 0)
        return set_error(TGT_ERR_INVALID, "add_layer_norm fwd: null tensor");
    if (bad_dtype(x_dtype) || bad_dtype(y_dtype) || bad_dtype(res_dtype)) return set_error(TGT_ERR_INVALID, "add_layer_norm fwd: bad dtype");
    if (scale && (rows_per_sample <= 0 || rows > 0xffffffffLL)) return set_error(TGT_ERR_INVALID, "add_layer_norm fwd: rows_per_sample (and rows < 2^32)");
    if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)res | (uintptr_t)s_out) % 16) return set_error(TGT_ERR_INVALID, "add_layer_norm fwd: tensors must be 16-byte aligned");
    if (rows == 0) return TGT_OK;
    LnArgs a = {};
    a.x = x; a.y = y; a.gamma = gamma; a.beta = beta; a.mean = mean; a.rstd = rstd;
    a.rows = rows; a.C = C; a.eps = eps; a.x_dtype = x_dtype; a.y_dtype = y_dtype;
    a.res = res; a.res_dtype = res_dtype; a.s_out = s_out; a.scale = scale; a.rows_per_sample = rows_per_sample;
    return ln_dispatch(a, false, nullptr, nullptr, st);
}

int add_layer_norm_bwd_run(const void* dy, int dy_dtype, const void* s, int s_dtype, const void* ds_in, int ds_dtype,
                           const float* scale, int64_t rows_per_sample, const float* gamma, const float* mean,
                           const float* rstd, void* d_res, void* d_x, int d_dtype, float* dgamma, float* dbeta,
                           float* d_x_colsum, float* partial, int64_t rows, int C, hipStream_t st) {
    if (!dy || !s || !gamma || !mean || !rstd || !d_res || !dgamma || !dbeta || !partial || rows < 0)
        return set_error(TGT_ERR_INVALID, "add_layer_norm bwd: null tensor");
    if (bad_dtype(s_dtype) || bad_dtype(dy_dtype) || bad_dtype(d_dtype) || (ds_in && bad_dtype(ds_dtype)))
        return set_error(TGT_ERR_INVALID, "add_layer_norm bwd: bad dtype");
    if (scale && (rows_per_sample <= 0 || rows > 0xffffffffLL)) return set_error(TGT_ERR_INVALID, "add_layer_norm bwd: scale needs rows_per_sample (and rows < 2^32)");
    if (scale && !d_x && !d_x_colsum) return set_error(TGT_ERR_INVALID, "add_layer_norm bwd: scale without d_x only shapes d_x_colsum");
    if (d_x_colsum && d_x_colsum != dbeta + C)
        return set_error(TGT_ERR_INVALID, "add_layer_norm bwd: d_x_colsum must directly follow dbeta (one 2C buffer)");
    if (((uintptr_t)s | (uintptr_t)dy | (uintptr_t)d_res | (uintptr_t)d_x | (uintptr_t)ds_in) % 16)
        return set_error(TGT_ERR_INVALID, "add_layer_norm bwd: tensors must be 16-byte aligned");
    LnArgs a = {};
    a.x = s; a.dy = dy; a.dx = d_res; a.gamma = gamma; a.mean = const_cast<float*>(mean); a.rstd = const_cast<float*>(rstd);
    a.partial = partial; a.rows = rows; a.C = C; a.x_dtype = s_dtype; a.dy_dtype = dy_dtype; a.dx_dtype = d_dtype;
    a.ds_in = ds_in; a.ds_dtype = ds_dtype; a.dx2 = d_x; a.scale = scale; a.rows_per_sample = rows_per_sample;
    a.x_colsum = d_x_colsum;
    return ln_dispatch(a, true, dgamma, dbeta, st);
}

int layer_norm_parts() { return kLnParts; }

template <int LPR, int VPL>
static int colsum_launch(const void* x, int dt, int64_t rows, int C, float* out, float* partial, hipStream_t st) {
    constexpr int RPW = 64 / LPR;
    const size_t lds = (size_t)4 * RPW * VPL * LPR * 8 * sizeof(float);
    int64_t parts = (rows + 4 * RPW - 1) / (4 * RPW);        // node-sized inputs: fewer, still full, blocks
    parts = parts < 1 ? 1 : (parts > kLnParts ? kLnParts : parts);
    hipLaunchKernelGGL((colsum_kernel<LPR, VPL>), dim3((unsigned)parts), dim3(256), lds, st, x, dt, rows, C, partial);
    if (int e = check_launch("colsum_kernel")) return e;
    hipLaunchKernelGGL(partial_sum_kernel, dim3((C + 7) / 8), dim3(256), 0, st, partial, (int)parts, C, C, out, out);
    return check_launch("partial_sum_kernel");
}

int sum_rows_run(const float* x, int rows, int C, float* out, hipStream_t st) {
    if (!x || !out || rows < 0 || C <= 0) return set_error(TGT_ERR_INVALID, "sum_rows: bad arguments");
    hipLaunchKernelGGL(partial_sum_kernel, dim3((C + 7) / 8), dim3(256), 0, st, x, rows, C, C, out, out);
    return check_launch("partial_sum_kernel");
}

int colsum_run(const void* x, int x_dtype, int64_t rows, int C, float* out, float* partial, hipStream_t st) {
    if (!x || !out || !partial || rows < 0) return set_error(TGT_ERR_INVALID, "colsum: null tensor");
    if (bad_dtype(x_dtype)) return set_error(TGT_ERR_INVALID, "colsum: bad dtype");
    if (C % 8 || C <= 0 || C > 4096) return set_error(TGT_ERR_UNSUPPORTED, "colsum: C=%d must be a multiple of 8, <= 4096", C);
    if ((uintptr_t)x % 16) return set_error(TGT_ERR_INVALID, "colsum: x must be 16-byte aligned");
    const int v8 = C / 8;
    if (v8 <= 4) return colsum_launch<4, 1>(x, x_dtype, rows, C, out, partial, st);
    if (v8 <= 8) return colsum_launch<8, 1>(x, x_dtype, rows, C, out, partial, st);
    if (v8 <= 16) return colsum_launch<16, 1>(x, x_dtype, rows, C, out, partial, st);
    if (v8 <= 32) return colsum_launch<32, 1>(x, x_dtype, rows, C, out, partial, st);
    if (v8 <= 64) return colsum_launch<64, 1>(x, x_dtype, rows, C, out, partial, st);
    if (v8 <= 128) return colsum_launch<64, 2>(x, x_dtype, rows, C, out, partial, st);
    if (v8 <= 256) return colsum_launch<64, 4>(x, x_dtype, rows, C, out, partial, st);
    return colsum_launch<64, 8>(x, x_dtype, rows, C, out, partial, st);
}

}  // namespace tgt
