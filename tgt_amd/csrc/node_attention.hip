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
#include "common.hpp"

namespace tgt {

struct NodeLane {
    bool active;
    int b, x, h;          // graph, node (query l or key m), head
};

// lanes per node row: smallest power of two >= min(H,64); nodes per wave = 64/that
__host__ __device__ inline int node_lpr(int H) {
    int l = 1;
    while (l < H && l < 64) l <<= 1;
    return l;
}

__device__ __forceinline__ NodeLane node_lane(const tgt_node_attention_args& a) {
    const int lpr = node_lpr(a.H), rpw = 64 / lpr, hb_count = (a.H + 63) / 64;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t unit = wave * rpw + lane / lpr;          // (b, x, hb)
    const int64_t total = (int64_t)a.B * a.N * hb_count;
    NodeLane n;
    const int hb = (int)(unit % hb_count);
    const int64_t bx = unit / hb_count;
    n.h = hb * 64 + lane % lpr;
    n.x = (int)(bx % a.N);
    n.b = (int)(bx / a.N);
    n.active = unit < total && n.h < a.H;
    return n;
}

template <typename T>
__device__ __forceinline__ float ld(const T* p, int64_t i) { return to_f32(p[i]); }

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
template <typename T, int D>
__global__ void __launch_bounds__(256) node_att_fwd_kernel(const tgt_node_attention_args a) {
    const NodeLane n = node_lane(a);
    if (!n.active) return;
    const int N = a.N, H = a.H, l = n.x, h = n.h;
    const T* qkv = reinterpret_cast<const T*>(a.qkv);
    const T* eg = reinterpret_cast<const T*>(a.eg);
    T* hhat = reinterpret_cast<T*>(a.hhat);
    const int64_t row_l = ((int64_t)n.b * N + l);

    float q[D], acc[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        q[d] = ld(qkv, row_l * a.ld_qkv + a.q_off + d * H + h) * a.scale;
        acc[d] = 0.f;
    }
    float mx = -INFINITY, sum = 0.f, gsum = 0.f;
    for (int m = 0; m < N; ++m) {
        const int64_t row_m = (int64_t)n.b * N + m, lm = row_l * N + m;
        float dot = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) dot += q[d] * ld(qkv, row_m * a.ld_qkv + a.k_off + d * H + h);
        const float s = dot + ld(eg, lm * a.ld_eg + a.e_off + h);
        if (hhat) hhat[lm * H + h] = from_f32<T>(s);
        if (a.logits_only) continue;
        const float mk = a.mask[lm];
        const float x = s + mk;
        const float g = fast_sigmoid(ld(eg, lm * a.ld_eg + a.g_off + h) + mk);
        // online softmax; mref = 0 while everything seen so far is -inf
        const float mnew = fmaxf(mx, x);
        const float mref = mnew == -INFINITY ? 0.f : mnew;
        const float corr = fast_exp(mx - mref), p = fast_exp(x - mref);
        sum = sum * corr + p;
        const float w = p * g;
#pragma unroll
        for (int d = 0; d < D; ++d)
            acc[d] = acc[d] * corr + w * ld(qkv, row_m * a.ld_qkv + a.v_off + d * H + h);
        gsum += g;
        mx = mnew;
    }
    if (a.logits_only) return;
    const float f = __frcp_rn(sum) * (a.scale_degree ? __logf(1.f + gsum) : 1.f);
    T* vatt = reinterpret_cast<T*>(a.vatt);
#pragma unroll
    for (int d = 0; d < D; ++d) vatt[row_l * (int64_t)(D * H) + d * H + h] = from_f32<T>(acc[d] * f);
    a.lse[row_l * H + h] = mx + __logf(sum);
    a.gsum[row_l * H + h] = gsum;
}

// ---------------------------------------------------------------------------
// backward, row pass: lane = (query l, head h).  Writes dE, dG and dQ.
// ---------------------------------------------------------------------------
template <typename T, int D>
__global__ void __launch_bounds__(256) node_att_bwd_row_kernel(const tgt_node_attention_args a) {
    const NodeLane n = node_lane(a);
    if (!n.active) return;
    const int N = a.N, H = a.H, l = n.x, h = n.h;
    const T* qkv = reinterpret_cast<const T*>(a.qkv);
    const T* eg = reinterpret_cast<const T*>(a.eg);
    const T* dhh = reinterpret_cast<const T*>(a.d_hhat);
    T* dqkv = reinterpret_cast<T*>(a.d_qkv);
    T* deg = reinterpret_cast<T*>(a.d_eg);
    const int64_t row_l = ((int64_t)n.b * N + l);

    float q[D], dq[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        q[d] = ld(qkv, row_l * a.ld_qkv + a.q_off + d * H + h) * a.scale;
        dq[d] = 0.f;
    }

    if (a.logits_only) {
        for (int m = 0; m < N; ++m) {
            const int64_t row_m = (int64_t)n.b * N + m, lm = row_l * N + m;
            const float dH = dhh ? ld(dhh, lm * H + h) : 0.f;
            deg[lm * a.ld_eg + a.e_off + h] = from_f32<T>(dH);
#pragma unroll
            for (int d = 0; d < D; ++d) dq[d] += dH * ld(qkv, row_m * a.ld_qkv + a.k_off + d * H + h);
        }
#pragma unroll
        for (int d = 0; d < D; ++d) dqkv[row_l * a.ld_qkv + a.q_off + d * H + h] = from_f32<T>(dq[d] * a.scale);
        return;
    }

    const T* dva = reinterpret_cast<const T*>(a.d_vatt);
    const float lse = a.lse[row_l * H + h], gsum = a.gsum[row_l * H + h];
    const float dsc = a.scale_degree ? __logf(1.f + gsum) : 1.f;
    float dv_att[D], vu[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        dv_att[d] = ld(dva, row_l * (int64_t)(D * H) + d * H + h);
        vu[d] = 0.f;
    }
    // pass 1: unscaled V_att (needed for delta and for d(log(1+gsum)))
    for (int m = 0; m < N; ++m) {
        const int64_t row_m = (int64_t)n.b * N + m, lm = row_l * N + m;
        float dot = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) dot += q[d] * ld(qkv, row_m * a.ld_qkv + a.k_off + d * H + h);
        const float mk = a.mask[lm];
        const float p = fast_exp(dot + ld(eg, lm * a.ld_eg + a.e_off + h) + mk - lse);
        const float g = fast_sigmoid(ld(eg, lm * a.ld_eg + a.g_off + h) + mk);
        const float w = p * g;
#pragma unroll
        for (int d = 0; d < D; ++d) vu[d] += w * ld(qkv, row_m * a.ld_qkv + a.v_off + d * H + h);
    }
    float d_dsc = 0.f, delta = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        d_dsc += dv_att[d] * vu[d];
        dv_att[d] *= dsc;                 // gradient wrt the unscaled V_att
        delta += dv_att[d] * vu[d];
    }
    const float dgsum = a.scale_degree ? d_dsc * __frcp_rn(1.f + gsum) : 0.f;
    // pass 2
    for (int m = 0; m < N; ++m) {
        const int64_t row_m = (int64_t)n.b * N + m, lm = row_l * N + m;
        float dot = 0.f, dA = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            dot += q[d] * ld(qkv, row_m * a.ld_qkv + a.k_off + d * H + h);
            dA += dv_att[d] * ld(qkv, row_m * a.ld_qkv + a.v_off + d * H + h);
        }
        const float mk = a.mask[lm];
        const float p = fast_exp(dot + ld(eg, lm * a.ld_eg + a.e_off + h) + mk - lse);
        const float g = fast_sigmoid(ld(eg, lm * a.ld_eg + a.g_off + h) + mk);
        const float dS = p * (dA * g - delta);
        const float dGl = (dA * p + dgsum) * g * (1.f - g);
        const float dH = dS + (dhh ? ld(dhh, lm * H + h) : 0.f);
        deg[lm * a.ld_eg + a.e_off + h] = from_f32<T>(dH);
        deg[lm * a.ld_eg + a.g_off + h] = from_f32<T>(dGl);
#pragma unroll
        for (int d = 0; d < D; ++d) dq[d] += dH * ld(qkv, row_m * a.ld_qkv + a.k_off + d * H + h);
    }
#pragma unroll
    for (int d = 0; d < D; ++d) dqkv[row_l * a.ld_qkv + a.q_off + d * H + h] = from_f32<T>(dq[d] * a.scale);
}

// ---------------------------------------------------------------------------
// backward, column pass: lane = (key m, head h).  dK and dV, reading the
// dH = dE the row pass stored (same stream, so ordered).
// ---------------------------------------------------------------------------
template <typename T, int D>
__global__ void __launch_bounds__(256) node_att_bwd_col_kernel(const tgt_node_attention_args a) {
    const NodeLane n = node_lane(a);
    if (!n.active) return;
    const int N = a.N, H = a.H, m = n.x, h = n.h;
    const T* qkv = reinterpret_cast<const T*>(a.qkv);
    const T* eg = reinterpret_cast<const T*>(a.eg);
    const T* deg = reinterpret_cast<const T*>(a.d_eg);
    const T* dva = reinterpret_cast<const T*>(a.d_vatt);
    T* dqkv = reinterpret_cast<T*>(a.d_qkv);
    const int64_t row_m = ((int64_t)n.b * N + m);

    float k[D], dk[D], dv[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        k[d] = ld(qkv, row_m * a.ld_qkv + a.k_off + d * H + h);
        dk[d] = dv[d] = 0.f;
    }
    for (int l = 0; l < N; ++l) {
        const int64_t row_l = (int64_t)n.b * N + l, lm = row_l * N + m;
        const float dH = ld(deg, lm * a.ld_eg + a.e_off + h);
        float dot = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float ql = ld(qkv, row_l * a.ld_qkv + a.q_off + d * H + h);
            dot += ql * k[d];
            dk[d] += dH * ql;
        }
        if (a.logits_only) continue;
        const float mk = a.mask[lm];
        const float p = fast_exp(dot * a.scale + ld(eg, lm * a.ld_eg + a.e_off + h) + mk - a.lse[row_l * H + h]);
        const float g = fast_sigmoid(ld(eg, lm * a.ld_eg + a.g_off + h) + mk);
        const float dsc = a.scale_degree ? __logf(1.f + a.gsum[row_l * H + h]) : 1.f;
        const float w = p * g * dsc;
#pragma unroll
        for (int d = 0; d < D; ++d) dv[d] += w * ld(dva, row_l * (int64_t)(D * H) + d * H + h);
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
        dqkv[row_m * a.ld_qkv + a.k_off + d * H + h] = from_f32<T>(dk[d] * a.scale);
        if (!a.logits_only) dqkv[row_m * a.ld_qkv + a.v_off + d * H + h] = from_f32<T>(dv[d]);
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
template <typename T, int D>
static int launch_node(const tgt_node_attention_args& a, bool bwd, hipStream_t st) {
    const int lpr = node_lpr(a.H), rpw = 64 / lpr, hb = (a.H + 63) / 64;
    const int64_t units = (int64_t)a.B * a.N * hb;
    const int64_t waves = (units + rpw - 1) / rpw;
    const int grid = (int)((waves + 3) / 4);
    if (!bwd) {
        hipLaunchKernelGGL((node_att_fwd_kernel<T, D>), dim3(grid), dim3(256), 0, st, a);
        return check_launch("node_att_fwd_kernel");
    }
    hipLaunchKernelGGL((node_att_bwd_row_kernel<T, D>), dim3(grid), dim3(256), 0, st, a);
    if (int e = check_launch("node_att_bwd_row_kernel")) return e;
    hipLaunchKernelGGL((node_att_bwd_col_kernel<T, D>), dim3(grid), dim3(256), 0, st, a);
    return check_launch("node_att_bwd_col_kernel");
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
    if (a->B <= 0 || a->N <= 0 || a->H <= 0) return set_error(TGT_ERR_INVALID, "node attention: bad sizes B=%d N=%d H=%d", a->B, a->N, a->H);
    if (!a->qkv || !a->eg) return set_error(TGT_ERR_INVALID, "node attention: null qkv/eg");
    if (a->logits_only) {
        if (!bwd && !a->hhat) return set_error(TGT_ERR_INVALID, "node attention: logits_only needs hhat");
    } else if (!a->mask || !a->lse || !a->gsum || (!bwd && !a->vatt)) {
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
