// Node attention with edge bias and gate (EGT_Attention) and the logits-only
// EdgeUpdate, forward and backward, for gfx950.
//
// Replaces reference lib/tgt/layers/layers.py:62-77 (einsum -> +E -> softmax *
// sigmoid gate -> einsum -> degree scaler) and :120-124, plus their autograd
// backward.  Math: SURVEY.md App. A.1 / A.4.
//
// This path is HBM-bound on the three (B,N,N,H) tensors E, G, H_hat
// (0.59 MB/graph vs 3 MFLOP/graph), so it is laid out for coalescing, not for
// the matrix core: the reference's channel order is HEAD-MINOR (c = d*H + h),
// hence   lane <-> head.   A 64-lane wave reads one (l,m) row of E/G and writes
// one row of H_hat as a single contiguous segment, and Q/K/V[.,d,:] rows are
// contiguous over heads too.  Each lane owns one (query l, head h) pair, keeps
// q[D] and the output accumulator in registers and walks the keys m with an
// online softmax: no cross-lane traffic, no LDS, no atomics.
//   backward = two passes with the same mapping:
//     row pass    (lane = (l,h)):  dE, dG (written once), dQ
//     column pass (lane = (m,h)):  dK, dV  (reads the dE the row pass wrote)
#include <cstdlib>
#include "common.hpp"

namespace tgt {

// One lane owns head h of RL consecutive nodes x0..x0+RL-1 (queries in the forward
// and row pass, keys in the column pass): every K/V (or Q/dV_att) value a lane
// loads is reused RL times from registers, which divides the vector-memory
// instruction count -- the real limiter of this kernel -- by ~RL.
struct NodeLane {
    bool active;
    int b, x0, h;
};

// lanes per node block: smallest power of two >= min(H,64); blocks per wave = 64/that
__host__ __device__ inline int node_lpr(int H) {
    int l = 1;
    while (l < H && l < 64) l <<= 1;
    return l;
}

template <int RL>
__device__ __forceinline__ NodeLane node_lane(const tgt_node_attention_args& a) {
    const int lpr = node_lpr(a.H), rpw = 64 / lpr, hb_count = (a.H + 63) / 64;
    const int nblk = (a.N + RL - 1) / RL;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t unit = wave * rpw + lane / lpr;          // (b, node block, hb)
    const int64_t total = (int64_t)a.B * nblk * hb_count;
    NodeLane n;
    const int hb = (int)(unit % hb_count);
    const int64_t bx = unit / hb_count;
    n.h = hb * 64 + lane % lpr;
    n.x0 = (int)(bx % nblk) * RL;
    n.b = (int)(bx / nblk);
    n.active = unit < total && n.h < a.H;
    return n;
}

template <typename T>
__device__ __forceinline__ float ld(const T* p, int64_t i) { return to_f32(p[i]); }

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
template <typename T, int D, int RL>
__global__ void __launch_bounds__(256) node_att_fwd_kernel(const tgt_node_attention_args a) {
    const NodeLane n = node_lane<RL>(a);
    if (!n.active) return;
    const int N = a.N, H = a.H, h = n.h;
    const T* qkv = reinterpret_cast<const T*>(a.qkv);
    const T* eg = reinterpret_cast<const T*>(a.eg);
    T* hhat = reinterpret_cast<T*>(a.hhat);
    const int64_t row0 = (int64_t)n.b * N;

    float q[RL][D], acc[RL][D], mx[RL], sum[RL], gsum[RL];
    bool live[RL];
#pragma unroll
    for (int t = 0; t < RL; ++t) {
        live[t] = n.x0 + t < N;
        const int64_t rl = row0 + (live[t] ? n.x0 + t : n.x0);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            q[t][d] = ld(qkv, rl * a.ld_qkv + a.q_off + d * H + h) * a.scale;
            acc[t][d] = 0.f;
        }
        mx[t] = -INFINITY;
        sum[t] = gsum[t] = 0.f;
    }
    for (int m = 0; m < N; ++m) {
        const int64_t row_m = row0 + m;
        float kk[D], vv[D];
#pragma unroll
        for (int d = 0; d < D; ++d) kk[d] = ld(qkv, row_m * a.ld_qkv + a.k_off + d * H + h);
        if (!a.logits_only) {
#pragma unroll
            for (int d = 0; d < D; ++d) vv[d] = ld(qkv, row_m * a.ld_qkv + a.v_off + d * H + h);
        }
#pragma unroll
        for (int t = 0; t < RL; ++t) {
            if (!live[t]) continue;
            const int64_t lm = (row0 + n.x0 + t) * N + m;
            float dot = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) dot += q[t][d] * kk[d];
            const float s = dot + ld(eg, lm * a.ld_eg + a.e_off + h);
            if (hhat) hhat[lm * H + h] = from_f32<T>(s);
            if (a.logits_only) continue;
            const float mk = a.mask[lm];
            const float x = s + mk;
            const float g = fast_sigmoid(ld(eg, lm * a.ld_eg + a.g_off + h) + mk);
            // online softmax; mref = 0 while everything seen so far is -inf
            const float mnew = fmaxf(mx[t], x);
            const float mref = mnew == -INFINITY ? 0.f : mnew;
            const float corr = fast_exp(mx[t] - mref), p = fast_exp(x - mref);
            sum[t] = sum[t] * corr + p;
            const float w = p * g;
#pragma unroll
            for (int d = 0; d < D; ++d) acc[t][d] = acc[t][d] * corr + w * vv[d];
            gsum[t] += g;
            mx[t] = mnew;
        }
    }
    if (a.logits_only) return;
    T* vatt = reinterpret_cast<T*>(a.vatt);
#pragma unroll
    for (int t = 0; t < RL; ++t) {
        if (!live[t]) continue;
        const int64_t rl = row0 + n.x0 + t;
        const float f = __frcp_rn(sum[t]) * (a.scale_degree ? __logf(1.f + gsum[t]) : 1.f);
#pragma unroll
        for (int d = 0; d < D; ++d) vatt[rl * (int64_t)(D * H) + d * H + h] = from_f32<T>(acc[t][d] * f);
        a.lse[rl * H + h] = mx[t] + __logf(sum[t]);
        a.gsum[rl * H + h] = gsum[t];
    }
}

// ---------------------------------------------------------------------------
// backward, row pass: lane = (RL queries, head h).  Writes dE, dG and dQ.
// ---------------------------------------------------------------------------
template <typename T, int D, int RL>
__global__ void __launch_bounds__(256) node_att_bwd_row_kernel(const tgt_node_attention_args a) {
    const NodeLane n = node_lane<RL>(a);
    if (!n.active) return;
    const int N = a.N, H = a.H, h = n.h;
    const T* qkv = reinterpret_cast<const T*>(a.qkv);
    const T* eg = reinterpret_cast<const T*>(a.eg);
    const T* dhh = reinterpret_cast<const T*>(a.d_hhat);
    T* dqkv = reinterpret_cast<T*>(a.d_qkv);
    T* deg = reinterpret_cast<T*>(a.d_eg);
    const int64_t row0 = (int64_t)n.b * N;

    float q[RL][D], dq[RL][D];
    bool live[RL];
#pragma unroll
    for (int t = 0; t < RL; ++t) {
        live[t] = n.x0 + t < N;
        const int64_t rl = row0 + (live[t] ? n.x0 + t : n.x0);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            q[t][d] = ld(qkv, rl * a.ld_qkv + a.q_off + d * H + h) * a.scale;
            dq[t][d] = 0.f;
        }
    }

    if (a.logits_only) {
        for (int m = 0; m < N; ++m) {
            float kk[D];
#pragma unroll
            for (int d = 0; d < D; ++d) kk[d] = ld(qkv, (row0 + m) * a.ld_qkv + a.k_off + d * H + h);
#pragma unroll
            for (int t = 0; t < RL; ++t) {
                if (!live[t]) continue;
                const int64_t lm = (row0 + n.x0 + t) * N + m;
                const float dH = dhh ? ld(dhh, lm * H + h) : 0.f;
                deg[lm * a.ld_eg + a.e_off + h] = from_f32<T>(dH);
#pragma unroll
                for (int d = 0; d < D; ++d) dq[t][d] += dH * kk[d];
            }
        }
#pragma unroll
        for (int t = 0; t < RL; ++t)
            if (live[t])
#pragma unroll
                for (int d = 0; d < D; ++d)
                    dqkv[(row0 + n.x0 + t) * a.ld_qkv + a.q_off + d * H + h] = from_f32<T>(dq[t][d] * a.scale);
        return;
    }

    const T* dva = reinterpret_cast<const T*>(a.d_vatt);
    float lse[RL], gsum[RL], dv_att[RL][D], vu[RL][D];
#pragma unroll
    for (int t = 0; t < RL; ++t) {
        const int64_t rl = row0 + (live[t] ? n.x0 + t : n.x0);
        lse[t] = a.lse[rl * H + h];
        gsum[t] = a.gsum[rl * H + h];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            dv_att[t][d] = ld(dva, rl * (int64_t)(D * H) + d * H + h);
            vu[t][d] = 0.f;
        }
    }
    // unscaled V_att (needed for delta and for d(log(1+gsum))) from the saved forward output:
    // V_att = vu * log(1+gsum); a zero scaler means every gate of the row was 0, i.e. vu = 0.
    {
        const T* va = reinterpret_cast<const T*>(a.vatt);
#pragma unroll
        for (int t = 0; t < RL; ++t) {
            const int64_t rl = row0 + (live[t] ? n.x0 + t : n.x0);
            const float dsc = a.scale_degree ? __logf(1.f + gsum[t]) : 1.f;
            const float inv = dsc != 0.f ? __frcp_rn(dsc) : 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) vu[t][d] = ld(va, rl * (int64_t)(D * H) + d * H + h) * inv;
        }
    }
    float delta[RL], dgsum[RL];
#pragma unroll
    for (int t = 0; t < RL; ++t) {
        const float dsc = a.scale_degree ? __logf(1.f + gsum[t]) : 1.f;
        float d_dsc = 0.f;
        delta[t] = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            d_dsc += dv_att[t][d] * vu[t][d];
            dv_att[t][d] *= dsc;                 // gradient wrt the unscaled V_att
            delta[t] += dv_att[t][d] * vu[t][d];
        }
        dgsum[t] = a.scale_degree ? d_dsc * __frcp_rn(1.f + gsum[t]) : 0.f;
    }
    // pass 2
    for (int m = 0; m < N; ++m) {
        const int64_t row_m = row0 + m;
        float kk[D], vv[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            kk[d] = ld(qkv, row_m * a.ld_qkv + a.k_off + d * H + h);
            vv[d] = ld(qkv, row_m * a.ld_qkv + a.v_off + d * H + h);
        }
#pragma unroll
        for (int t = 0; t < RL; ++t) {
            if (!live[t]) continue;
            const int64_t lm = (row0 + n.x0 + t) * N + m;
            float dot = 0.f, dA = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                dot += q[t][d] * kk[d];
                dA += dv_att[t][d] * vv[d];
            }
            const float mk = a.mask[lm];
            const float p = fast_exp(dot + ld(eg, lm * a.ld_eg + a.e_off + h) + mk - lse[t]);
            const float g = fast_sigmoid(ld(eg, lm * a.ld_eg + a.g_off + h) + mk);
            const float dS = p * (dA * g - delta[t]);
            const float dGl = (dA * p + dgsum[t]) * g * (1.f - g);
            const float dH = dS + (dhh ? ld(dhh, lm * H + h) : 0.f);
            deg[lm * a.ld_eg + a.e_off + h] = from_f32<T>(dH);
            deg[lm * a.ld_eg + a.g_off + h] = from_f32<T>(dGl);
#pragma unroll
            for (int d = 0; d < D; ++d) dq[t][d] += dH * kk[d];
        }
    }
#pragma unroll
    for (int t = 0; t < RL; ++t)
        if (live[t])
#pragma unroll
            for (int d = 0; d < D; ++d)
                dqkv[(row0 + n.x0 + t) * a.ld_qkv + a.q_off + d * H + h] = from_f32<T>(dq[t][d] * a.scale);
}

// ---------------------------------------------------------------------------
// backward, column pass: lane = (RL keys m, head h).  dK and dV, reading the
// dH = dE the row pass stored (same stream, so ordered).
// ---------------------------------------------------------------------------
template <typename T, int D, int RL>
__global__ void __launch_bounds__(256) node_att_bwd_col_kernel(const tgt_node_attention_args a) {
    const NodeLane n = node_lane<RL>(a);
    if (!n.active) return;
    const int N = a.N, H = a.H, h = n.h;
    const T* qkv = reinterpret_cast<const T*>(a.qkv);
    const T* eg = reinterpret_cast<const T*>(a.eg);
    const T* deg = reinterpret_cast<const T*>(a.d_eg);
    const T* dva = reinterpret_cast<const T*>(a.d_vatt);
    T* dqkv = reinterpret_cast<T*>(a.d_qkv);
    const int64_t row0 = (int64_t)n.b * N;

    float k[RL][D], dk[RL][D], dv[RL][D];
    bool live[RL];
#pragma unroll
    for (int t = 0; t < RL; ++t) {
        live[t] = n.x0 + t < N;
        const int64_t rm = row0 + (live[t] ? n.x0 + t : n.x0);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            k[t][d] = ld(qkv, rm * a.ld_qkv + a.k_off + d * H + h);
            dk[t][d] = dv[t][d] = 0.f;
        }
    }
    for (int l = 0; l < N; ++l) {
        const int64_t row_l = row0 + l;
        float ql[D], dvl[D];
#pragma unroll
        for (int d = 0; d < D; ++d) ql[d] = ld(qkv, row_l * a.ld_qkv + a.q_off + d * H + h);
        float lse_l = 0.f, dsc = 1.f;
        if (!a.logits_only) {
            lse_l = a.lse[row_l * H + h];
            dsc = a.scale_degree ? __logf(1.f + a.gsum[row_l * H + h]) : 1.f;
#pragma unroll
            for (int d = 0; d < D; ++d) dvl[d] = ld(dva, row_l * (int64_t)(D * H) + d * H + h) * dsc;
        }
#pragma unroll
        for (int t = 0; t < RL; ++t) {
            if (!live[t]) continue;
            const int64_t lm = row_l * N + n.x0 + t;
            const float dH = ld(deg, lm * a.ld_eg + a.e_off + h);
            float dot = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                dot += ql[d] * k[t][d];
                dk[t][d] += dH * ql[d];
            }
            if (a.logits_only) continue;
            const float mk = a.mask[lm];
            const float p = fast_exp(dot * a.scale + ld(eg, lm * a.ld_eg + a.e_off + h) + mk - lse_l);
            const float g = fast_sigmoid(ld(eg, lm * a.ld_eg + a.g_off + h) + mk);
            const float w = p * g;
#pragma unroll
            for (int d = 0; d < D; ++d) dv[t][d] += w * dvl[d];
        }
    }
#pragma unroll
    for (int t = 0; t < RL; ++t) {
        if (!live[t]) continue;
        const int64_t rm = row0 + n.x0 + t;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            dqkv[rm * a.ld_qkv + a.k_off + d * H + h] = from_f32<T>(dk[t][d] * a.scale);
            if (!a.logits_only) dqkv[rm * a.ld_qkv + a.v_off + d * H + h] = from_f32<T>(dv[t][d]);
        }
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static int node_grid(const tgt_node_attention_args& a, int rl) {
    const int lpr = node_lpr(a.H), rpw = 64 / lpr, hb = (a.H + 63) / 64;
    const int64_t units = (int64_t)a.B * ((a.N + rl - 1) / rl) * hb;
    const int64_t waves = (units + rpw - 1) / rpw;
    return (int)((waves + 3) / 4);
}

template <typename T, int D, int RF, int RR, int RC>
static int launch_node_r(const tgt_node_attention_args& a, bool bwd, hipStream_t st) {
    if (!bwd) {
        hipLaunchKernelGGL((node_att_fwd_kernel<T, D, RF>), dim3(node_grid(a, RF)), dim3(256), 0, st, a);
        return check_launch("node_att_fwd_kernel");
    }
    hipLaunchKernelGGL((node_att_bwd_row_kernel<T, D, RR>), dim3(node_grid(a, RR)), dim3(256), 0, st, a);
    if (int e = check_launch("node_att_bwd_row_kernel")) return e;
    hipLaunchKernelGGL((node_att_bwd_col_kernel<T, D, RC>), dim3(node_grid(a, RC)), dim3(256), 0, st, a);
    return check_launch("node_att_bwd_col_kernel");
}

template <typename T, int D>
static int launch_node(const tgt_node_attention_args& a, bool bwd, hipStream_t st) {
    // nodes per lane: registers hold RL x (q, acc) [fwd], RL x (q, dq, dV_att, V_att) [row], RL x (k, dk, dv) [col]
    static const int knob = getenv("TGT_NODE_RL") ? atoi(getenv("TGT_NODE_RL")) : 0;     // experiment knob
    if constexpr (D <= 16) {
        if (knob == 1) return launch_node_r<T, D, 1, 1, 1>(a, bwd, st);
        if (knob == 2) return launch_node_r<T, D, 2, 2, 2>(a, bwd, st);
        if (knob == 4) return launch_node_r<T, D, 4, 4, 4>(a, bwd, st);
        return launch_node_r<T, D, 4, 2, 2>(a, bwd, st);
    } else {
        return launch_node_r<T, D, 2, 1, 1>(a, bwd, st);
    }
}

template <typename T>
static int dispatch_node_d(const tgt_node_attention_args& a, bool bwd, hipStream_t st) {
    switch (a.D) {
        case 4: return launch_node<T, 4>(a, bwd, st);
        case 8: return launch_node<T, 8>(a, bwd, st);
        case 12: return launch_node<T, 12>(a, bwd, st);
        case 16: return launch_node<T, 16>(a, bwd, st);
        case 24: return launch_node<T, 24>(a, bwd, st);
        case 32: return launch_node<T, 32>(a, bwd, st);
        default: return set_error(TGT_ERR_UNSUPPORTED, "node attention: D=%d not in {4,8,12,16,24,32}", a.D);
    }
}

int node_attention_run(const tgt_node_attention_args* a, bool bwd, hipStream_t st) {
    if (!a) return set_error(TGT_ERR_INVALID, "node attention: null args");
    if (a->B < 0 || a->N < 0 || a->H <= 0) return set_error(TGT_ERR_INVALID, "node attention: bad sizes B=%d N=%d H=%d", a->B, a->N, a->H);
    if (a->B == 0 || a->N == 0) return TGT_OK;
    if (!a->qkv || !a->eg) return set_error(TGT_ERR_INVALID, "node attention: null qkv/eg");
    if (a->logits_only) {
        if (!bwd && !a->hhat) return set_error(TGT_ERR_INVALID, "node attention: logits_only needs hhat");
    } else if (!a->mask || !a->lse || !a->gsum || !a->vatt) {
        return set_error(TGT_ERR_INVALID, "node attention: null mask/vatt/lse/gsum");
    }
    if (bwd) {
        if (!a->d_qkv || !a->d_eg) return set_error(TGT_ERR_INVALID, "node attention bwd: null d_qkv/d_eg");
        if (!a->logits_only && !a->d_vatt) return set_error(TGT_ERR_INVALID, "node attention bwd: null d_vatt");
    }
    switch (a->dtype) {
        case TGT_F32: return dispatch_node_d<float>(*a, bwd, st);
        case TGT_BF16: return dispatch_node_d<bf16_t>(*a, bwd, st);
        case TGT_F16: return dispatch_node_d<f16_t>(*a, bwd, st);
        default: return set_error(TGT_ERR_INVALID, "node attention: bad dtype %d", a->dtype);
    }
}

}  // namespace tgt
