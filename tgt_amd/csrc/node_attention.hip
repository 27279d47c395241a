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

// One lane owns HV adjacent heads of one node (a query in the forward and row pass, a key in
// the column pass).  HV = 4 turns every access into an 8-byte (bf16) / 16-byte (fp32) vector:
// this kernel is bound by vector-memory INSTRUCTIONS (2-byte accesses fill 128 B per wave
// instruction), not by bytes, so wider lanes are the lever.  64 / (H/HV) nodes share a wave;
// they read the same K/V rows, which the memory pipeline serves once per instruction.
struct NodeLane {
    bool active;
    int b, x, h;          // graph, node, first head of this lane
};

// lanes per node = smallest power of two >= min(H/HV, 64)
__host__ __device__ inline int node_lpr(int H, int HV) {
    int l = 1;
    while (l < (H + HV - 1) / HV && l < 64) l <<= 1;
    return l;
}

template <int HV>
__device__ __forceinline__ NodeLane node_lane(const tgt_node_attention_args& a) {
    const int lpr = node_lpr(a.H, HV), rpw = 64 / lpr, hb_count = (a.H + 64 * HV - 1) / (64 * HV);
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t unit = wave * rpw + lane / lpr;          // (b, node, head block)
    const int64_t total = (int64_t)a.B * a.N * hb_count;
    NodeLane n;
    const int hb = (int)(unit % hb_count);
    const int64_t bx = unit / hb_count;
    n.h = (hb * 64 + lane % lpr) * HV;
    n.x = (int)(bx % a.N);
    n.b = (int)(bx / a.N);
    n.active = unit < total && n.h < a.H;
    return n;
}

// HV contiguous elements (HV*sizeof(T)-byte aligned) <-> floats
template <typename T, int HV>
__device__ __forceinline__ void ldv(const T* p, int64_t i, float (&o)[HV]) {
    T t[HV];
    if constexpr (HV * sizeof(T) == 16) { uint4 r = *reinterpret_cast<const uint4*>(p + i); __builtin_memcpy(t, &r, 16); }
    else if constexpr (HV * sizeof(T) == 8) { uint2 r = *reinterpret_cast<const uint2*>(p + i); __builtin_memcpy(t, &r, 8); }
    else if constexpr (HV * sizeof(T) == 4) { uint32_t r = *reinterpret_cast<const uint32_t*>(p + i); __builtin_memcpy(t, &r, 4); }
    else { t[0] = p[i]; }
#pragma unroll
    for (int k = 0; k < HV; ++k) o[k] = to_f32(t[k]);
}
template <typename T, int HV>
__device__ __forceinline__ void stv(T* p, int64_t i, const float (&v)[HV]) {
    T t[HV];
#pragma unroll
    for (int k = 0; k < HV; ++k) t[k] = from_f32<T>(v[k]);
    if constexpr (HV * sizeof(T) == 16) { uint4 r; __builtin_memcpy(&r, t, 16); *reinterpret_cast<uint4*>(p + i) = r; }
    else if constexpr (HV * sizeof(T) == 8) { uint2 r; __builtin_memcpy(&r, t, 8); *reinterpret_cast<uint2*>(p + i) = r; }
    else if constexpr (HV * sizeof(T) == 4) { uint32_t r; __builtin_memcpy(&r, t, 4); *reinterpret_cast<uint32_t*>(p + i) = r; }
    else { p[i] = t[0]; }
}
template <int HV>
__device__ __forceinline__ void ldf(const float* p, int64_t i, float (&o)[HV]) {
#pragma unroll
    for (int k = 0; k < HV; ++k) o[k] = p[i + k];
}

// All D values of HV adjacent heads of one Q/K/V (or V_att) row, kept packed in T: the reference's head-MINOR channel
// order c = d*H + h -> D accesses of HV elements.  (A head-major ABI option, c = h*D + d with one contiguous block per lane,
// was measured slower end to end in round 1 and removed in round 3.)
template <typename T, int D, int HV>
struct DH {
    T v[D * HV];
    __device__ __forceinline__ float at(int d, int k) const { return to_f32(v[d * HV + k]); }
    __device__ __forceinline__ void load(const T* p, int64_t base, int H, int h) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            constexpr int NB = HV * (int)sizeof(T);
            const T* src = p + base + (int64_t)d * H + h;
            if constexpr (NB == 16) { uint4 w = *reinterpret_cast<const uint4*>(src); __builtin_memcpy(v + d * HV, &w, 16); }
            else if constexpr (NB == 8) { uint2 w = *reinterpret_cast<const uint2*>(src); __builtin_memcpy(v + d * HV, &w, 8); }
            else if constexpr (NB == 4) { uint32_t w = *reinterpret_cast<const uint32_t*>(src); __builtin_memcpy(v + d * HV, &w, 4); }
            else { v[d * HV] = src[0]; }
        }
    }
    __device__ static __forceinline__ void store(T* p, int64_t base, int H, int h, const float (&x)[D][HV]) {
#pragma unroll
        for (int d = 0; d < D; ++d) stv<T, HV>(p, base + (int64_t)d * H + h, x[d]);
    }
};

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
template <typename T, int D, int HV>
__global__ void __launch_bounds__(256) node_att_fwd_kernel(const tgt_node_attention_args a) {
    const NodeLane n = node_lane<HV>(a);
    if (!n.active) return;
    const int N = a.N, H = a.H, h = n.h;
    const T* qkv = reinterpret_cast<const T*>(a.qkv);
    const T* eg = reinterpret_cast<const T*>(a.eg);
    T* hhat = reinterpret_cast<T*>(a.hhat);
    const float hs = a.hhat_scale ? a.hhat_scale[n.b] : 1.f;       // H_hat is written times the branch's DropPath factor
    const int64_t row0 = (int64_t)n.b * N, row_l = row0 + n.x;

    float q[D][HV], acc[D][HV], mx[HV], sum[HV], gsum[HV];
    {
        DH<T, D, HV> qb;
        qb.load(qkv, row_l * a.ld_qkv + a.q_off, H, h);
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int k = 0; k < HV; ++k) {
                q[d][k] = qb.at(d, k) * a.scale;
                acc[d][k] = 0.f;
            }
    }
#pragma unroll
    for (int k = 0; k < HV; ++k) {
        mx[k] = -INFINITY;
        sum[k] = gsum[k] = 0.f;
    }
    for (int m = 0; m < N; ++m) {
        const int64_t row_m = row0 + m, lm = row_l * N + m;
        float e[HV], g[HV], s[HV];
        ldv<T, HV>(eg, lm * a.ld_eg + a.e_off + h, e);
        if (!a.logits_only) ldv<T, HV>(eg, lm * a.ld_eg + a.g_off + h, g);
#pragma unroll
        for (int k = 0; k < HV; ++k) s[k] = e[k];
        {
            DH<T, D, HV> kb;
            kb.load(qkv, row_m * a.ld_qkv + a.k_off, H, h);
#pragma unroll
            for (int d = 0; d < D; ++d)
#pragma unroll
                for (int k = 0; k < HV; ++k) s[k] += q[d][k] * kb.at(d, k);
        }
        if (hhat) {
                        float so[HV];
#pragma unroll
                        for (int k = 0; k < HV; ++k) so[k] = s[k] * hs;
                        stv<T, HV>(hhat, lm * H + h, so);
                    }
        if (a.logits_only) continue;
        const float mk = a.mask[lm];
        float corr[HV], w[HV];
#pragma unroll
        for (int k = 0; k < HV; ++k) {
            const float x = s[k] + mk;
            const float gt = fast_sigmoid(g[k] + mk);
            // online softmax; mref = 0 while everything seen so far is -inf
            const float mnew = fmaxf(mx[k], x);
            const float mref = mnew == -INFINITY ? 0.f : mnew;
            corr[k] = fast_exp(mx[k] - mref);
            const float p = fast_exp(x - mref);
            sum[k] = sum[k] * corr[k] + p;
            w[k] = p * gt;
            gsum[k] += gt;
            mx[k] = mnew;
        }
        {
            DH<T, D, HV> vb;
            vb.load(qkv, row_m * a.ld_qkv + a.v_off, H, h);
#pragma unroll
            for (int d = 0; d < D; ++d)
#pragma unroll
                for (int k = 0; k < HV; ++k) acc[d][k] = acc[d][k] * corr[k] + w[k] * vb.at(d, k);
        }
    }
    if (a.logits_only) return;
    T* vatt = reinterpret_cast<T*>(a.vatt);
    float f[HV], lse[HV];
#pragma unroll
    for (int k = 0; k < HV; ++k) {
        f[k] = fast_rcp(sum[k]) * (a.scale_degree ? __logf(1.f + gsum[k]) : 1.f);
        lse[k] = mx[k] + __logf(sum[k]);
    }
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int k = 0; k < HV; ++k) acc[d][k] *= f[k];
    DH<T, D, HV>::store(vatt, row_l * (int64_t)(D * H), H, h, acc);
#pragma unroll
    for (int k = 0; k < HV; ++k) {
        a.lse[row_l * H + h + k] = lse[k];
        a.gsum[row_l * H + h + k] = gsum[k];
    }
}

// ---------------------------------------------------------------------------
// forward, LDS variant (the fast path for 16-bit types): one workgroup owns QB query nodes of
// ONE graph and stages that graph's K and V rows in LDS once (tiles of MT keys), so the walk
// over keys issues only the streaming accesses to global memory -- E, G (read) and H_hat
// (written), 8 bytes per lane -- and those run PD keys ahead in a register ring.
//   lane = (query l, HV heads), lanes of the same wave that share heads read the same LDS words.
// ---------------------------------------------------------------------------
template <typename T, int D, int HV, int MT>
__global__ void __launch_bounds__(512) node_att_fwd_lds_kernel(const tgt_node_attention_args a, int lpr, int qb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int N = a.N, H = a.H, W = D * H;
    T* sK = reinterpret_cast<T*>(smem);
    T* sV = sK + (size_t)MT * W;
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int nblk = (N + qb - 1) / qb;
    const int b = blockIdx.x / nblk, l = (blockIdx.x % nblk) * qb + tid / lpr, h = (tid % lpr) * HV;
    const bool active = l < N && h < H;
    const T* qkv = reinterpret_cast<const T*>(a.qkv);
    const T* eg = reinterpret_cast<const T*>(a.eg);
    T* hhat = reinterpret_cast<T*>(a.hhat);
    const float hs = a.hhat_scale ? a.hhat_scale[b] : 1.f;         // H_hat is written times the branch's DropPath factor
    const int64_t row0 = (int64_t)b * N, row_l = row0 + (active ? l : 0);

    float q[D][HV], acc[D][HV], mx[HV], sum[HV], gsum[HV];
    {
        DH<T, D, HV> qb;
        if (active) qb.load(qkv, row_l * a.ld_qkv + a.q_off, H, h);
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int k = 0; k < HV; ++k) {
                q[d][k] = active ? qb.at(d, k) * a.scale : 0.f;
                acc[d][k] = 0.f;
            }
    }
#pragma unroll
    for (int k = 0; k < HV; ++k) {
        mx[k] = -INFINITY;
        sum[k] = gsum[k] = 0.f;
    }

    constexpr int PD = 8;
    constexpr int VE = 16 / (int)sizeof(T);              // elements per 16-byte staging chunk
    for (int mt0 = 0; mt0 < N; mt0 += MT) {
        const int mt = min(MT, N - mt0);
        __syncthreads();                                   // previous tile fully consumed
        for (int c = tid; c < mt * (W / VE); c += nthreads) {
            const int m = c / (W / VE), off = (c % (W / VE)) * VE;
            const int64_t g = (row0 + mt0 + m) * a.ld_qkv;
            *reinterpret_cast<uint4*>(sK + (size_t)m * W + off) = *reinterpret_cast<const uint4*>(qkv + g + a.k_off + off);
            if (!a.logits_only)
                *reinterpret_cast<uint4*>(sV + (size_t)m * W + off) = *reinterpret_cast<const uint4*>(qkv + g + a.v_off + off);
        }
        __syncthreads();
        if (!active) continue;

        float er[PD][HV], gr[PD][HV], mr[PD];
        auto fetch = [&](int slot, int m) {
            const int64_t lm = row_l * N + mt0 + m;
            ldv<T, HV>(eg, lm * a.ld_eg + a.e_off + h, er[slot]);
            if (!a.logits_only) {
                ldv<T, HV>(eg, lm * a.ld_eg + a.g_off + h, gr[slot]);
                mr[slot] = a.mask[lm];
            }
        };
#pragma unroll
        for (int k = 0; k < PD; ++k)
            if (k < mt) fetch(k, k);
        for (int m0 = 0; m0 < mt; m0 += PD) {
#pragma unroll
            for (int kk = 0; kk < PD; ++kk) {
                const int m = m0 + kk;
                if (m < mt) {
                    const int64_t lm = row_l * N + mt0 + m;
                    float s[HV];
#pragma unroll
                    for (int k = 0; k < HV; ++k) s[k] = er[kk][k];
                    {
                        DH<T, D, HV> kb;
                        kb.load(sK, (int64_t)m * W, H, h);
#pragma unroll
                        for (int d = 0; d < D; ++d)
#pragma unroll
                            for (int k = 0; k < HV; ++k) s[k] += q[d][k] * kb.at(d, k);
                    }
                    if (hhat) {
                        float so[HV];
#pragma unroll
                        for (int k = 0; k < HV; ++k) so[k] = s[k] * hs;
                        stv<T, HV>(hhat, lm * H + h, so);
                    }
                    if (!a.logits_only) {
                        const float mk = mr[kk];
                        float corr[HV], w[HV];
#pragma unroll
                        for (int k = 0; k < HV; ++k) {
                            const float x = s[k] + mk;
                            const float gt = fast_sigmoid(gr[kk][k] + mk);
                            const float mnew = fmaxf(mx[k], x);
                            const float mref = mnew == -INFINITY ? 0.f : mnew;
                            corr[k] = fast_exp(mx[k] - mref);
                            const float p = fast_exp(x - mref);
                            sum[k] = sum[k] * corr[k] + p;
                            w[k] = p * gt;
                            gsum[k] += gt;
                            mx[k] = mnew;
                        }
                        {
                            DH<T, D, HV> vb;
                            vb.load(sV, (int64_t)m * W, H, h);
#pragma unroll
                            for (int d = 0; d < D; ++d)
#pragma unroll
                                for (int k = 0; k < HV; ++k) acc[d][k] = acc[d][k] * corr[k] + w[k] * vb.at(d, k);
                        }
                    }
                    if (m + PD < mt) fetch(kk, m + PD);
                }
            }
        }
    }
    if (a.logits_only || !active) return;
    T* vatt = reinterpret_cast<T*>(a.vatt);
    float f[HV];
#pragma unroll
    for (int k = 0; k < HV; ++k) {
        f[k] = fast_rcp(sum[k]) * (a.scale_degree ? __logf(1.f + gsum[k]) : 1.f);
        a.lse[row_l * H + h + k] = mx[k] + __logf(sum[k]);
        a.gsum[row_l * H + h + k] = gsum[k];
    }
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int k = 0; k < HV; ++k) acc[d][k] *= f[k];
    DH<T, D, HV>::store(vatt, row_l * (int64_t)(D * H), H, h, acc);
}

// ---------------------------------------------------------------------------
// backward, row pass: lane = (query l, HV heads).  Writes dE, dG and dQ.
// ---------------------------------------------------------------------------
template <typename T, int D, int HV>
__global__ void __launch_bounds__(256) node_att_bwd_row_kernel(const tgt_node_attention_args a) {
    const NodeLane n = node_lane<HV>(a);
    if (!n.active) return;
    const int N = a.N, H = a.H, h = n.h;
    const T* qkv = reinterpret_cast<const T*>(a.qkv);
    const T* eg = reinterpret_cast<const T*>(a.eg);
    const T* dhh = reinterpret_cast<const T*>(a.d_hhat);
    T* dqkv = reinterpret_cast<T*>(a.d_qkv);
    T* deg = reinterpret_cast<T*>(a.d_eg);
    const float hs = a.hhat_scale ? a.hhat_scale[n.b] : 1.f;       // d_hhat is the gradient of hhat_scale * H_hat
    const int64_t row0 = (int64_t)n.b * N, row_l = row0 + n.x;

    float q[D][HV], dq[D][HV];
    {
        DH<T, D, HV> qb;
        qb.load(qkv, row_l * a.ld_qkv + a.q_off, H, h);
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int k = 0; k < HV; ++k) {
                q[d][k] = qb.at(d, k) * a.scale;
                dq[d][k] = 0.f;
            }
    }

    if (a.logits_only) {
        for (int m = 0; m < N; ++m) {
            const int64_t lm = row_l * N + m;
            float dH[HV];
#pragma unroll
            for (int k = 0; k < HV; ++k) dH[k] = 0.f;
            if (dhh) {
                ldv<T, HV>(dhh, lm * H + h, dH);
#pragma unroll
                for (int k = 0; k < HV; ++k) dH[k] *= hs;
            }
            stv<T, HV>(deg, lm * a.ld_eg + a.e_off + h, dH);
            DH<T, D, HV> kb;
            kb.load(qkv, (row0 + m) * a.ld_qkv + a.k_off, H, h);
#pragma unroll
            for (int d = 0; d < D; ++d)
#pragma unroll
                for (int k = 0; k < HV; ++k) dq[d][k] += dH[k] * kb.at(d, k);
        }
    } else {
        const T* dva = reinterpret_cast<const T*>(a.d_vatt);
        const T* va = reinterpret_cast<const T*>(a.vatt);
        float lse[HV], gsum[HV], dsc[HV], inv[HV], d_dsc[HV], delta[HV], dgsum[HV], dvu[D][HV];
        ldf<HV>(a.lse, row_l * H + h, lse);
        ldf<HV>(a.gsum, row_l * H + h, gsum);
#pragma unroll
        for (int k = 0; k < HV; ++k) {
            dsc[k] = a.scale_degree ? __logf(1.f + gsum[k]) : 1.f;
            inv[k] = dsc[k] != 0.f ? fast_rcp(dsc[k]) : 0.f;     // zero scaler <=> every gate 0 <=> V_att 0
            d_dsc[k] = delta[k] = 0.f;
        }
        // unscaled V_att from the saved forward output: V_att = vu * log(1+gsum)
        {
            DH<T, D, HV> db, ob;
            db.load(dva, row_l * (int64_t)(D * H), H, h);
            ob.load(va, row_l * (int64_t)(D * H), H, h);
#pragma unroll
            for (int d = 0; d < D; ++d)
#pragma unroll
                for (int k = 0; k < HV; ++k) {
                    const float vu = ob.at(d, k) * inv[k];
                    dvu[d][k] = db.at(d, k);
                    d_dsc[k] += dvu[d][k] * vu;
                    dvu[d][k] *= dsc[k];                       // gradient wrt the unscaled V_att
                    delta[k] += dvu[d][k] * vu;
                }
        }
#pragma unroll
        for (int k = 0; k < HV; ++k) dgsum[k] = a.scale_degree ? d_dsc[k] * fast_rcp(1.f + gsum[k]) : 0.f;
        for (int m = 0; m < N; ++m) {
            const int64_t row_m = row0 + m, lm = row_l * N + m;
            float e[HV], g[HV], dot[HV], dA[HV], dH[HV], dGl[HV];
            ldv<T, HV>(eg, lm * a.ld_eg + a.e_off + h, e);
            ldv<T, HV>(eg, lm * a.ld_eg + a.g_off + h, g);
#pragma unroll
            for (int k = 0; k < HV; ++k) dH[k] = 0.f;
            if (dhh) {
                ldv<T, HV>(dhh, lm * H + h, dH);
#pragma unroll
                for (int k = 0; k < HV; ++k) dH[k] *= hs;
            }
#pragma unroll
            for (int k = 0; k < HV; ++k) dot[k] = dA[k] = 0.f;
            DH<T, D, HV> kb, vb;
            kb.load(qkv, row_m * a.ld_qkv + a.k_off, H, h);
            vb.load(qkv, row_m * a.ld_qkv + a.v_off, H, h);
#pragma unroll
            for (int d = 0; d < D; ++d)
#pragma unroll
                for (int k = 0; k < HV; ++k) {
                    dot[k] += q[d][k] * kb.at(d, k);
                    dA[k] += dvu[d][k] * vb.at(d, k);
                }
            const float mk = a.mask[lm];
#pragma unroll
            for (int k = 0; k < HV; ++k) {
                const float p = fast_exp(dot[k] + e[k] + mk - lse[k]);
                const float gt = fast_sigmoid(g[k] + mk);
                const float dS = p * (dA[k] * gt - delta[k]);
                dGl[k] = (dA[k] * p + dgsum[k]) * gt * (1.f - gt);
                dH[k] += dS;
            }
            stv<T, HV>(deg, lm * a.ld_eg + a.e_off + h, dH);
            stv<T, HV>(deg, lm * a.ld_eg + a.g_off + h, dGl);
#pragma unroll
            for (int d = 0; d < D; ++d)
#pragma unroll
                for (int k = 0; k < HV; ++k) dq[d][k] += dH[k] * kb.at(d, k);
        }
    }
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int k = 0; k < HV; ++k) dq[d][k] *= a.scale;
    DH<T, D, HV>::store(dqkv, row_l * a.ld_qkv + a.q_off, H, h, dq);
}

// ---------------------------------------------------------------------------
// backward, column pass: lane = (key m, HV heads).  dK and dV, reading the
// dH = dE the row pass stored (same stream, so ordered).
// ---------------------------------------------------------------------------
template <typename T, int D, int HV>
__global__ void __launch_bounds__(256) node_att_bwd_col_kernel(const tgt_node_attention_args a) {
    const NodeLane n = node_lane<HV>(a);
    if (!n.active) return;
    const int N = a.N, H = a.H, h = n.h, m = n.x;
    const T* qkv = reinterpret_cast<const T*>(a.qkv);
    const T* eg = reinterpret_cast<const T*>(a.eg);
    const T* deg = reinterpret_cast<const T*>(a.d_eg);
    const T* dva = reinterpret_cast<const T*>(a.d_vatt);
    T* dqkv = reinterpret_cast<T*>(a.d_qkv);
    const int64_t row0 = (int64_t)n.b * N, row_m = row0 + m;

    float kv[D][HV], dk[D][HV], dv[D][HV];
    {
        DH<T, D, HV> kb;
        kb.load(qkv, row_m * a.ld_qkv + a.k_off, H, h);
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int k = 0; k < HV; ++k) {
                kv[d][k] = kb.at(d, k);
                dk[d][k] = dv[d][k] = 0.f;
            }
    }
    for (int l = 0; l < N; ++l) {
        const int64_t row_l = row0 + l, lm = row_l * N + m;
        float dH[HV], dot[HV];
        ldv<T, HV>(deg, lm * a.ld_eg + a.e_off + h, dH);
#pragma unroll
        for (int k = 0; k < HV; ++k) dot[k] = 0.f;
        {
            DH<T, D, HV> qb;
            qb.load(qkv, row_l * a.ld_qkv + a.q_off, H, h);
#pragma unroll
            for (int d = 0; d < D; ++d)
#pragma unroll
                for (int k = 0; k < HV; ++k) {
                    const float ql = qb.at(d, k);
                    dot[k] += ql * kv[d][k];
                    dk[d][k] += dH[k] * ql;
                }
        }
        if (a.logits_only) continue;
        float e[HV], g[HV], lse[HV], gsum[HV], w[HV];
        ldv<T, HV>(eg, lm * a.ld_eg + a.e_off + h, e);
        ldv<T, HV>(eg, lm * a.ld_eg + a.g_off + h, g);
        ldf<HV>(a.lse, row_l * H + h, lse);
        ldf<HV>(a.gsum, row_l * H + h, gsum);
        const float mk = a.mask[lm];
#pragma unroll
        for (int k = 0; k < HV; ++k) {
            const float p = fast_exp(dot[k] * a.scale + e[k] + mk - lse[k]);
            const float gt = fast_sigmoid(g[k] + mk);
            w[k] = p * gt * (a.scale_degree ? __logf(1.f + gsum[k]) : 1.f);
        }
        {
            DH<T, D, HV> db;
            db.load(dva, row_l * (int64_t)(D * H), H, h);
#pragma unroll
            for (int d = 0; d < D; ++d)
#pragma unroll
                for (int k = 0; k < HV; ++k) dv[d][k] += w[k] * db.at(d, k);
        }
    }
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int k = 0; k < HV; ++k) dk[d][k] *= a.scale;
    DH<T, D, HV>::store(dqkv, row_m * a.ld_qkv + a.k_off, H, h, dk);
    if (!a.logits_only) DH<T, D, HV>::store(dqkv, row_m * a.ld_qkv + a.v_off, H, h, dv);
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static int node_grid(const tgt_node_attention_args& a, int hv) {
    const int lpr = node_lpr(a.H, hv), rpw = 64 / lpr, hb = (a.H + 64 * hv - 1) / (64 * hv);
    const int64_t units = (int64_t)a.B * a.N * hb;
    const int64_t waves = (units + rpw - 1) / rpw;
    return (int)((waves + 3) / 4);
}

// heads per lane: the widest vector the layout allows (every offset / row length a multiple of it)
static int node_vec(const tgt_node_attention_args& a, int esz, int want) {
    for (int hv = want; hv > 1; hv >>= 1) {
        const bool ok = a.H % hv == 0 && a.ld_qkv % hv == 0 && a.ld_eg % hv == 0 && a.q_off % hv == 0 &&
                        a.k_off % hv == 0 && a.v_off % hv == 0 && a.e_off % hv == 0 && a.g_off % hv == 0 &&
                        ((uintptr_t)a.qkv % (hv * esz)) == 0 && ((uintptr_t)a.eg % (hv * esz)) == 0;
        if (ok) return hv;
    }
    return 1;
}

#define TGT_NODE_LAUNCH(KERNEL, NAME, WANT)                                                                        \
    do {                                                                                                          \
        const int hv = node_vec(a, (int)sizeof(T), (D <= 16) ? (WANT) : ((WANT) > 2 ? 2 : (WANT)));               \
        if (hv == 4) { if constexpr (D <= 16) hipLaunchKernelGGL((KERNEL<T, D, 4>), dim3(node_grid(a, 4)), dim3(256), 0, st, a); } \
        else if (hv == 2) hipLaunchKernelGGL((KERNEL<T, D, 2>), dim3(node_grid(a, 2)), dim3(256), 0, st, a);       \
        else hipLaunchKernelGGL((KERNEL<T, D, 1>), dim3(node_grid(a, 1)), dim3(256), 0, st, a);                    \
        if (int e = check_launch(NAME)) return e;                                                                 \
    } while (0)

template <typename T, int D>
static int launch_node(const tgt_node_attention_args& a, bool bwd, hipStream_t st) {
    // measured on MI355X (B=256 N=32 H=64 D=12 bf16): forward best with 4 heads per lane, both
    // backward passes with 2 (4 pushes them to one wave per SIMD)
    constexpr int hv_f = 4, hv_r = 2, hv_c = 2;
    if (!bwd) {
        // LDS variant: 16-bit types, 4 heads per lane, K/V rows 16-byte aligned, one key tile of <= 32 keys
        // fits the 160 KB LDS (2 * 32 * W * 2 bytes = 96 KB at W = 768)
        constexpr bool use_lds = true;
        if constexpr (sizeof(T) == 2 && D <= 16) {
            constexpr int MT = 32;
            const int W = D * a.H;
            const size_t lds = (size_t)2 * MT * W * sizeof(T);
            const int lpr = (a.H + 3) / 4;
            if (use_lds && node_vec(a, 2, 4) == 4 && W % 8 == 0 && a.ld_qkv % 8 == 0 && a.k_off % 8 == 0 &&
                a.v_off % 8 == 0 && lds <= 160 * 1024 && lpr <= 512 && a.H % 4 == 0) {
                int qb = 512 / lpr;                        // query nodes per workgroup
                if (qb > a.N) qb = a.N;
                if (qb > 32) qb = 32;
                const int threads = ((qb * lpr + 63) / 64) * 64;
                const int grid = a.B * ((a.N + qb - 1) / qb);
                hipLaunchKernelGGL((node_att_fwd_lds_kernel<T, D, 4, MT>), dim3(grid), dim3(threads), lds, st, a, lpr, qb);
                return check_launch("node_att_fwd_lds_kernel");
            }
        }
        TGT_NODE_LAUNCH(node_att_fwd_kernel, "node_att_fwd_kernel", hv_f);
        return TGT_OK;
    }
    TGT_NODE_LAUNCH(node_att_bwd_row_kernel, "node_att_bwd_row_kernel", hv_r);
    TGT_NODE_LAUNCH(node_att_bwd_col_kernel, "node_att_bwd_col_kernel", hv_c);
    return TGT_OK;
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

// matrix-core kernels for the 16-bit hot shapes (node_attention_mfma.hip)
bool node_attention_mfma_eligible(const tgt_node_attention_args& a, bool bwd);
int node_attention_mfma_run(const tgt_node_attention_args& a, bool bwd, hipStream_t st);
// 16-wide matrix-core tiles for 33 <= N <= 64 (node_attention16.hip): BASELINE config 4
bool node_attention16_eligible(const tgt_node_attention_args& a, bool bwd);
int node_attention16_run(const tgt_node_attention_args& a, bool bwd, hipStream_t st);
// key-blocked, whole 128-byte E | G rows per workgroup (node_attention_kb.hip): H a multiple of 32
bool node_attention_kb_eligible(const tgt_node_attention_args& a, bool bwd);
int node_attention_kb_run(const tgt_node_attention_args& a, bool bwd, hipStream_t st);

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
    if (node_attention_kb_eligible(*a, bwd)) return node_attention_kb_run(*a, bwd, st);    // forward, H % 32 == 0
    if (node_attention16_eligible(*a, bwd)) return node_attention16_run(*a, bwd, st);      // N > 32 (every N <= 64 under TGT_NODE_MFMA16=2)
    if (node_attention_mfma_eligible(*a, bwd)) return node_attention_mfma_run(*a, bwd, st);
    switch (a->dtype) {
        case TGT_F32: return dispatch_node_d<float>(*a, bwd, st);
        case TGT_BF16: return dispatch_node_d<bf16_t>(*a, bwd, st);
        case TGT_F16: return dispatch_node_d<f16_t>(*a, bwd, st);
        default: return set_error(TGT_ERR_INVALID, "node attention: bad dtype %d", a->dtype);
    }
}

}  // namespace tgt
