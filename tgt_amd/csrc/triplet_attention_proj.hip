// Triplet attention forward with the Q/K/V projection fused in -- gfx950, wave roles (round 3).
//
// Reference lib/tgt/layers/triplet.py:210-246: `lin_QKV_in/out(e_ln)` followed by the two
// einsum -> softmax -> gate -> einsum chains.  The unfused path writes the 1536 projected channels of every edge
// (0.8 GB at B=256) with a library GEMM (0.26 ms) and reads them straight back in the attention kernel (0.225 ms).
// Here the workgroup that walks node j projects the edge rows it needs itself and feeds the attention core through
// LDS; Q/K/V still go to HBM once (the backward kernel reads them), but nothing is read back in the forward and the
// tall-skinny GEMM disappears.
//
//   workgroup = (graph b, direction, 8 heads), 16 waves in two ROLES (128 registers each, 4 waves per SIMD):
//     waves 8-15  PROJECTION of head h = wave - 8: the 48 weight rows [W_Q(h); W_K(h); W_V(h)] x 256 k resident in 96
//                 registers for the whole walk; per step the LayerNorm'd edge rows of step j+1 (an X tile in LDS) give
//                 [K|V]^T = [W_K;W_V] X_kv^T (16 v_mfma_32x32x16) and Q^T = W_Q X_q^T (16 v_mfma_16x16x32), written as
//                 rows into the slab set of step j+1;
//     waves 0-7   ATTENTION of head h = wave on the slab set of step j (the core of tri_att_fwd_kernel: S^T = K Q^T,
//                 softmax over k in the lane + one half-wave exchange, gate, O^T = V^T P^T), plus the X-tile prefetch
//                 (global -> registers at the top of the step, -> LDS at its end);
//   so every SIMD holds two matrix-heavy and two VALU-heavy waves.  ONE barrier per step; after it all 1024 threads store
//   the Q/K/V rows of step j+1 and the O rows of step j as whole 256-byte row pieces (raw buffer stores, out-of-range
//   offsets instead of branches: triplet_common.hpp).
//
// Why roles: the round-1 form kept the weights in the attention waves (128 + ~100 registers: 2 waves per
// SIMD in ONE workgroup per CU, all phases behind the same barrier) and measured 0.77 ms; its ISA shows what that cost --
// 256 registers + 132 bytes of scratch reloaded inside the walk behind `s_waitcnt vmcnt(0)`, i.e. every prefetch was a
// synchronous load.
// Supported: N <= 32, D = 16, H % 8 == 0, 16-bit dtypes, C = 256, no attention dropout.
#include <cstdlib>
#include "triplet_common.hpp"

namespace tgt {

__device__ __forceinline__ f32x4 mma16x32(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mma16x32(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

namespace proj2 {
constexpr int kC = 256, kKS = 16, kRowBytes = kC * 2, kSlots = kRowBytes / 16;      // an X row: 512 bytes = 32 slots of 16
constexpr int kXPitch = kRowBytes + 16;                                             // LDS pitch of an X row: 16 bytes of padding
constexpr int kXTile = 64 * kXPitch;                                                // 33 KB (inward: 64 rows; outward uses 32)
constexpr int kSlab = 32 * 8 * 16 * 2;                                              // 32 rows x (8 heads x 16 d) x 2 B = 8 KB
constexpr int kOffSets = 2 * kXTile, kSet = 3 * kSlab;                              // {Q | K | V} x 2
constexpr int kOffO = kOffSets + 2 * kSet, kOffBias = kOffO + 2 * kSlab;            // O slabs x 2, then the bias floats
constexpr int kLds = kOffBias + 8 * 48 * 4;
constexpr uint32_t kNone = 0xffffffffu;
// Rows PADDED by one 16-byte slot instead of XOR-swizzled: the 16 rows a ds_read_b128 lane group touches land in 16 different
// bank slots (pitch 528 = 33 slots), and a fragment address is  lane base + k-step x constant  -- the k-loop's reads take
// immediate offsets.  (With the XOR form every k-step needs its own address register or three VALU instructions; hipcc chose
// the registers, hoisted them out of the walk and spilled them next to the resident weights.)
__device__ __forceinline__ int xoff(int row, int slot) { return row * kXPitch + (slot << 4); }
}  // namespace proj2

template <typename T, int DIR>
__device__ __forceinline__ void proj2_walk(const tgt_triplet_attention_args& a, const TriCtx& c, const T* x, const T* w,
                                           char* smem, int tid) {
    using namespace proj2;
    constexpr int D = 16, HG = 8;
    using G = TriGeo<T, D, HG>;
    using F = frag_t<T>;
    const int lane = tid & 63, wave = tid >> 6, r = lane & 31, hi = lane >> 5;
#ifdef TGT_PROBES
    const uint32_t ablate = a._pad1;                   // probe builds only (TGT_PROJ_ABLATE): 1 no X loads, 2 no stores, 4 no projection, 8 no attention math
#else
    constexpr uint32_t ablate = 0;                     // (the shipped library ignores _pad1)
#endif
    const bool attend = wave < 8;                      // role
    const int hw = wave & 7;                           // head inside the group
    const int N = c.N;
    char* xbuf = smem;
    char* sets = smem + kOffSets;
    char* oslab = smem + kOffO;
    const float* bias_w = reinterpret_cast<const float*>(smem + kOffBias) + hw * 48;

    // global addressing: X rows of the graph (read), Q/K/V rows (written for the backward), O rows
    const int64_t sz = sizeof(T), Nl = N;
    const __amdgpu_buffer_rsrc_t r_x = graph_rsrc(x, Nl * Nl * kRowBytes, c.b);
    const uint32_t hch = (uint32_t)(c.g * HG * D * sz), lds_ = (uint32_t)(a.ld_qkv[DIR] * sz), ldo_ = (uint32_t)(a.ld_out * sz);
    const __amdgpu_buffer_rsrc_t r_dst = graph_rsrc(a.qkv[DIR], Nl * Nl * a.ld_qkv[DIR] * sz, c.b);
    const __amdgpu_buffer_rsrc_t r_out = graph_rsrc(a.out, Nl * Nl * a.ld_out * sz, c.b);
    // slab s of a step: 0 Q rows (i,j) | 1 K, 2 V partner rows (j,k) inward, (k,j) outward | 3 O rows (i,j)
    const uint32_t q_row = (uint32_t)N * lds_, q_j = lds_;
    const uint32_t kv_row = DIR == 0 ? lds_ : (uint32_t)N * lds_, kv_j = DIR == 0 ? (uint32_t)N * lds_ : lds_;
    const uint32_t o_row = (uint32_t)N * ldo_, o_j = ldo_;
    // this thread's 16-byte chunks of the four slabs of a step: threads 0-511 take Q and V, threads 512-1023 K and O.  Every thread
    // issues the same three stores (two through the Q/K/V resource, one through the O resource; the one that is not its own goes
    // out of range) -- a per-thread choice of the RESOURCE or of the scalar offset would make hipcc emit a waterfall loop around
    // the store.  The offsets are rebuilt from the thread index at every step (a dozen VALU instructions) instead of living in
    // registers through the walk: the projection role has none to spare.
    auto store_step = [&](int jq, bool qkv_live, int jo, bool o_live) {
        // Q/K/V rows of step jq (slab set jq & 1) and O rows of step jo (O slab jo & 1); a dead part goes out of range
        int t_ = tid;
        asm volatile("" : "+v"(t_));                     // (opaque: nothing below is hoisted out of the walk)
        const int srow = (t_ & 511) >> 4, sslot = t_ & 15, shalf = t_ >> 9;
        const int s_lds = G::lds_off(srow, sslot);
        const bool row_ok = srow < N;
        const uint32_t c16 = (uint32_t)sslot * 16u + hch;
        const char* set = sets + (jq & 1) * kSet;
        const uint4 va = *reinterpret_cast<const uint4*>(set + shalf * kSlab + s_lds);
        const uint4 vb = *reinterpret_cast<const uint4*>((shalf == 0 ? set + 2 * kSlab : oslab + (jo & 1) * kSlab) + s_lds);
        const uint32_t vo_a = (uint32_t)srow * (shalf == 0 ? q_row : kv_row) + (uint32_t)jq * (shalf == 0 ? q_j : kv_j) +
                              (uint32_t)((shalf == 0 ? a.q_off[DIR] : a.k_off[DIR]) * sz) + c16;                     // Q (half 0) / K (half 1)
        const uint32_t vo_v = (uint32_t)srow * kv_row + (uint32_t)jq * kv_j + (uint32_t)(a.v_off[DIR] * sz) + c16;     // V (half 0)
        const uint32_t vo_o = (uint32_t)srow * o_row + (uint32_t)jo * o_j + (uint32_t)(a.o_off[DIR] * sz) + c16;       // O (half 1)
        u32x4_t da = {va.x, va.y, va.z, va.w}, db = {vb.x, vb.y, vb.z, vb.w};
        if (ablate & 2) return;
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_raw_buffer_store_b128(da, r_dst, (int)((row_ok && qkv_live) ? vo_a : kNone), 0, TGT_ST_AUX);
        __builtin_amdgcn_raw_buffer_store_b128(db, r_dst, (int)((row_ok && qkv_live && shalf == 0) ? vo_v : kNone), 0, TGT_ST_AUX);
        __builtin_amdgcn_raw_buffer_store_b128(db, r_out, (int)((row_ok && o_live && shalf == 1) ? vo_o : kNone), 0, TGT_ST_AUX);
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- a DropPath-dropped graph: zeros to the O rows, nothing else is read, computed or written (the backward of such a graph
    //      gets a zero d_out and must be given graph_scale too: ops.py)
    if (a.graph_scale && a.graph_scale[c.b] == 0.f) {
        const int srow = (tid & 511) >> 4, sslot = tid & 15;
        const uint32_t vo = (srow < N && tid >= 512) ? (uint32_t)srow * o_row + (uint32_t)(a.o_off[DIR] * sz) + hch + (uint32_t)sslot * 16u : kNone;
        for (int j = 0; j < N; ++j) {
            u32x4_t z = {0, 0, 0, 0};
            __builtin_amdgcn_raw_buffer_store_b128(z, r_out, (int)vo, (int)((uint32_t)j * o_j), TGT_ST_AUX);
        }
        return;
    }

    // The two roles run two SEPARATE loops with the same barrier sequence (B0 B1: third arm; B2: X tiles 0, 1; B3: step 0 projected;
    // then one per step) -- in one shared loop the register allocator would have to keep the 96 weight registers of one role and
    // the ~60 registers of tile state of the other alive together.
    // Hazards with ONE barrier per step (B_j = the barrier of iteration j):
    //  * slab set (j+1)&1 is written by the projection waves in iteration j; it was last read by the attention waves in iteration
    //    j-1 (before B_{j-1}) and by every thread's store of step j-1 right after B_{j-2}: one barrier lies in between.
    //  * X tile j&1 is overwritten (x_commit of tile j+2) at the END of iteration j by the attention waves; the projection waves
    //    read it in iteration j-1 (before B_{j-1}).  X tile (j+1)&1, read in iteration j, was committed before B_{j-1}.
    //  * O slab j&1 is written in iteration j and stored after B_j; it is written again in iteration j+2, after B_{j+1}.
    const int l16 = lane & 15, l4 = lane >> 4;
    // Static wave priority (round 4): the projection role -- the younger half of the workgroup, which loses the per-SIMD issue
    // arbitration by age, and the matrix-pipe-heavy one -- runs at s_setprio 1 for the whole walk: 0.459 -> 0.420 ms at B = 256
    // (profiles/r05n_proj_prio.txt; 2 = the attention role instead: no change).  The condition must be provably wave-uniform
    // (readfirstlane): s_setprio ignores EXEC.
#ifndef TGT_PROJ_PRIO
#define TGT_PROJ_PRIO 1
#endif
    if (TGT_PROJ_PRIO != 0 && __builtin_amdgcn_readfirstlane(wave) >= 8 == (TGT_PROJ_PRIO == 1)) __builtin_amdgcn_s_setprio(1);
    if (attend) {
        // ------------------------------------------------------------------------------------------- attention role
        float biasM[16], gate[16];
        const ThirdArm ta = tri_third_arm(a, DIR);
        arm_stage_load<T, HG, 1>(ta, c.b, DIR, c.g, N, 0, smem, tid);
        __syncthreads();
        arm_stage_read<T, HG, 1, false>(ta, smem, DIR, hw, N, r, hi, 0, 0, biasM, gate);
        __syncthreads();
        // X tile prefetch (512 threads: 4 chunks inward, 2 outward)
        constexpr int kXIt = DIR == 0 ? 4 : 2;
        uint32_t xvo[kXIt];
        int xlo[kXIt];
#pragma unroll
        for (int it = 0; it < kXIt; ++it) {
            const int cidx = it * 512 + tid, row = cidx >> 5, slot = cidx & 31;
            // tile rows 0..31: edge (row, j) -- j enters as a scalar offset;  rows 32..63 (inward only): edge (j, row - 32)
            const int node = row & 31;
            xvo[it] = node < N ? (row < 32 ? (uint32_t)node * (uint32_t)N * kRowBytes : (uint32_t)node * kRowBytes) + (uint32_t)slot * 16u : kNone;
            xlo[it] = xoff(row, slot);
        }
        uint4 px[kXIt];
        auto x_issue = [&](int jx) {                    // (jx >= N: out of range, zeros)
            const bool live = jx < N && !(ablate & 1);
#pragma unroll
            for (int it = 0; it < kXIt; ++it) {
                const bool kvrows = DIR == 0 && it >= 2;
                const uint32_t so = live ? (uint32_t)jx * (kvrows ? (uint32_t)N * kRowBytes : (uint32_t)kRowBytes) : 0u;
                const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r_x, (int)(live ? xvo[it] : kNone), (int)so, 0);
                px[it] = make_uint4(v.x, v.y, v.z, v.w);
            }
        };
        auto x_commit = [&](int jx) {
            char* xt = xbuf + (jx & 1) * kXTile;
#pragma unroll
            for (int it = 0; it < kXIt; ++it) *reinterpret_cast<uint4*>(xt + xlo[it]) = px[it];
        };
        F ident_d[1];
        make_ident_d<T, 1>(ident_d, r, hi);
        x_issue(0);
        x_commit(0);
        x_issue(1);
        x_commit(1);
        __syncthreads();                                // B2
        __syncthreads();                                // B3 (the projection waves produce step 0)
        store_step(0, true, 0, false);
        for (int j = 0; j < N; ++j) {
            x_issue(j + 2);
            asm volatile("" ::: "memory");              // (the prefetch is issued HERE, a whole step ahead of its commit)
            if (!(ablate & 8)) {
                const char* set = sets + (j & 1) * kSet;
                F fq[1], fk[1], fv[1];
                read_frags<T, D, HG>(fq, set, hw, r, hi);
                read_frags<T, D, HG>(fk, set + kSlab, hw, r, hi);
                read_frags<T, D, HG>(fv, set + 2 * kSlab, hw, r, hi);
                f32x16 z0 = {0}, z1 = {0};
                f32x16 st = mma32(fk[0], fq[0], z0);    // S^T[k][i]
                f32x16 vt = mma32(fv[0], ident_d[0], z1);   // V[k][d] -> lane d
                float mx = -INFINITY;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    st[q] = st[q] * a.scale + biasM[q];
                    mx = fmaxf(mx, st[q]);
                }
                mx = fmaxf(mx, xhalf(mx));
                float sum = 0.f;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    st[q] = fast_exp(st[q] - mx);
                    sum += st[q];
                }
                sum += xhalf(sum);
                const float inv = fast_rcp(sum);
#pragma unroll
                for (int q = 0; q < 16; ++q) st[q] = st[q] * inv * gate[q];
                f32x16 o = {0};
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) o = mma32(pack_chunk<T>(vt, cc), pack_chunk<T>(st, cc), o);   // O^T[d][i]
                write_rows<T, D, HG>(oslab + (j & 1) * kSlab, o, hw, r, hi);
            }
            x_commit(j + 2);
            __syncthreads();                            // B_j
            store_step(j + 1, j + 1 < N, j, true);
        }
    } else {
        // ------------------------------------------------------------------------------------------ projection role
        // the head's 48 weight rows as A operands of v_mfma_f32_16x16x32 (16 rows x 32 k per fragment), resident for the whole walk:
        // part 0 = W_Q(h), 1 = W_K(h), 2 = W_V(h); eight k-steps each.  (16x16 tiles keep the accumulators at 8 registers: with one
        // 32x32 [K;V] tile the role needed 16 + 16 for the V copy and hipcc spilled weight fragments.)
        // W_Q(h): A operands of v_mfma_f32_16x16x32 (16 rows x 32 k: 8 fragments); [W_K(h); W_V(h)]: A operands of v_mfma_f32_32x32x16
        // (32 rows x 16 k: 16 fragments) -- one fragment read of the K/V tile feeds a 32-cycle instruction
        F wq[kKS / 2], wkv[kKS];
        {
            const T* rq = w + (int64_t)(a.q_off[DIR] + c.h * D + l16) * kC;
            const T* rkv = w + (int64_t)((r < 16 ? a.k_off[DIR] : a.v_off[DIR]) + c.h * D + (r & 15)) * kC;
#pragma unroll
            for (int s2 = 0; s2 < kKS / 2; ++s2) wq[s2] = load_frag<T>(rq + 32 * s2 + 8 * l4);
#pragma unroll
            for (int s2 = 0; s2 < kKS; ++s2) wkv[s2] = load_frag<T>(rkv + 16 * s2 + 8 * hi);
        }
        auto project = [&](int jn) {                    // Q/K/V rows of step jn from X tile jn & 1 into slab set jn & 1
            const char* xt = xbuf + (jn & 1) * kXTile;
            char* set = sets + (jn & 1) * kSet;
            constexpr int kv0 = DIR == 0 ? 32 : 0;      // tile rows of the K / V partner edges (Q: the edges (i, j), rows 0..31)
            {
                f32x16 kva;                             // [K;V]^T[ch = acc_row(q, hi)][row = r]
#pragma unroll
                for (int q = 0; q < 16; ++q) kva[q] = bias_w[16 + acc_row(q, hi)];
#pragma unroll
                for (int s2 = 0; s2 < kKS; ++s2) {
                    const F xk = load_frag<T>(reinterpret_cast<const T*>(xt + xoff(kv0 + r, 2 * s2 + hi)));
                    kva = mma32(wkv[s2], xk, kva);
                    // (left alone the scheduler hoists every fragment read of the tile to its top -- 64 registers next to the 96 of
                    // resident weights; a few reads in flight cover the LDS latency behind the other waves' work)
                    if ((s2 & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int part = 0; part < 2; ++part)      // K = tile rows 0..15 (registers 0..7), V = rows 16..31 (registers 8..15)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        T tmp[4];
#pragma unroll
                        for (int t = 0; t < 4; ++t) tmp[t] = from_f32<T>(kva[8 * part + 4 * q + t]);
                        uint2 v;
                        __builtin_memcpy(&v, tmp, 8);
                        *reinterpret_cast<uint2*>(set + (1 + part) * kSlab + G::lds_elem(r, hw * D + 8 * q + 4 * hi)) = v;
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                f32x4 q0, q1;                           // Q^T[d = 4 l4 + t][i = l16] and [i = 16 + l16]
#pragma unroll
                for (int t = 0; t < 4; ++t) q0[t] = q1[t] = bias_w[4 * l4 + t];
#pragma unroll
                for (int s2 = 0; s2 < kKS / 2; ++s2) {
                    const F x0 = load_frag<T>(reinterpret_cast<const T*>(xt + xoff(l16, 4 * s2 + l4)));
                    const F x1 = load_frag<T>(reinterpret_cast<const T*>(xt + xoff(16 + l16, 4 * s2 + l4)));
                    q0 = mma16x32(wq[s2], x0, q0);
                    q1 = mma16x32(wq[s2], x1, q1);
                    if (s2 & 1) __builtin_amdgcn_sched_barrier(0);
                }
                T t0[4], t1[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) { t0[t] = from_f32<T>(q0[t]); t1[t] = from_f32<T>(q1[t]); }
                uint2 v0, v1;
                __builtin_memcpy(&v0, t0, 8);
                __builtin_memcpy(&v1, t1, 8);
                *reinterpret_cast<uint2*>(set + G::lds_elem(l16, hw * D + 4 * l4)) = v0;
                *reinterpret_cast<uint2*>(set + G::lds_elem(16 + l16, hw * D + 4 * l4)) = v1;
            }
        };
        __syncthreads();                                // B0
        __syncthreads();                                // B1
        __syncthreads();                                // B2 (X tiles 0 and 1 are in LDS)
        project(0);
        __syncthreads();                                // B3
        store_step(0, true, 0, false);
        for (int j = 0; j < N; ++j) {
            if (j + 1 < N && !(ablate & 4)) project(j + 1);
            __syncthreads();                            // B_j
            store_step(j + 1, j + 1 < N, j, true);
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(1024, 4) tri_att_proj_fwd_kernel(const tgt_triplet_attention_args a, const T* x, const T* w,
                                                                   const T* bias) {
    constexpr int D = 16, HG = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // (unit order = tri_ctx's plain order.  Measured and rejected: the four (direction, head group) units of a graph as consecutive
    // blocks of ONE XCD, so that its X rows come from that XCD's L2 -- 0.59 against 0.555 ms.)
    const TriCtx c = tri_ctx<T, D, HG>(a, wave & 7);
    const int dir = c.dir;
    // bias of the heads: [Q_d | K_d | V_d], 16 each, as floats (projection waves write their head's)
    if (wave >= 8 && lane < 48) {
        const int part = lane >> 4, d = lane & 15;
        const int off = part == 0 ? a.q_off[dir] : (part == 1 ? a.k_off[dir] : a.v_off[dir]);
        reinterpret_cast<float*>(smem + proj2::kOffBias)[(wave & 7) * 48 + lane] = to_f32(bias[off + c.h * D + d]);
    }
    if (dir == 0) proj2_walk<T, 0>(a, c, x, w, smem, tid);
    else proj2_walk<T, 1>(a, c, x, w, smem, tid);
}

template <typename T>
static int launch_proj(const tgt_triplet_attention_args& a, const void* x, const void* w, const void* bias, hipStream_t st) {
    const int grid = a.B * 2 * (a.H / 8);
    static_assert(ArmStage<T, 8, 1>::kBytes <= proj2::kOffO, "arm stage must fit the aliased area");
    static bool attr_set[16] = {};
    if (!dyn_lds_once(attr_set, reinterpret_cast<const void*>(&tri_att_proj_fwd_kernel<T>), proj2::kLds))
        return set_error(TGT_ERR_LAUNCH, "tri_att_proj_fwd_kernel: cannot reserve %d bytes of LDS", proj2::kLds);
#ifdef TGT_PROBES
    static const int ablate = getenv("TGT_PROJ_ABLATE") ? atoi(getenv("TGT_PROJ_ABLATE")) : 0;
    tgt_triplet_attention_args aa = a;
    aa._pad1 = (uint32_t)ablate;
#else
    const tgt_triplet_attention_args& aa = a;          // (_pad1 is padding: never written, never read)
#endif
    hipLaunchKernelGGL((tri_att_proj_fwd_kernel<T>), dim3(grid), dim3(1024), proj2::kLds, st, aa, reinterpret_cast<const T*>(x),
                       reinterpret_cast<const T*>(w), reinterpret_cast<const T*>(bias));
    return check_launch("tri_att_proj_fwd_kernel");
}

int triplet_attention_proj_supported(const tgt_triplet_attention_args* a, int C) {
    return a && a->dropout_p == 0.f && a->N <= 32 && a->D == 16 && a->H % 8 == 0 && (a->dtype == TGT_BF16 || a->dtype == TGT_F16) &&
           C == 256;
}

int triplet_attention_proj_run(const tgt_triplet_attention_args* a, const void* x, int C, const void* w, const void* bias,
                               hipStream_t st) {
    if (!a || !x || !w || !bias) return set_error(TGT_ERR_INVALID, "projected triplet attention: null argument");
    if (a->B < 0 || a->N < 0 || a->H <= 0) return set_error(TGT_ERR_INVALID, "projected triplet attention: bad sizes");
    if (a->B == 0 || a->N == 0) return TGT_OK;
    if (!triplet_attention_proj_supported(a, C))
        return set_error(TGT_ERR_UNSUPPORTED,
                         "projected triplet attention needs N <= 32, D = 16, H %% 8 == 0, a 16-bit dtype, C = 256 and no attention dropout "
                         "(got N=%d D=%d H=%d dtype=%d C=%d)", a->N, a->D, a->H, a->dtype, C);
    for (int dir = 0; dir < 2; ++dir) {
        if (!a->qkv[dir] || !a->out || !a->mask) return set_error(TGT_ERR_INVALID, "projected triplet attention: null tensor");
        if ((a->ld_qkv[dir] * 2) % 16 || (a->q_off[dir] * 2) % 16 || (a->k_off[dir] * 2) % 16 || (a->v_off[dir] * 2) % 16 ||
            (a->ld_out * 2) % 16 || (a->o_off[dir] * 2) % 16 || ((uintptr_t)a->qkv[dir] % 16) || ((uintptr_t)a->out % 16))
            return set_error(TGT_ERR_INVALID, "projected triplet attention: rows/offsets must be 16-byte aligned");
        if ((a->flags & (TGT_TRI_BIASED | TGT_TRI_GATED)) && !a->eg[dir]) return set_error(TGT_ERR_INVALID, "projected triplet attention: eg missing");
    }
    if (((uintptr_t)x | (uintptr_t)w) % 16) return set_error(TGT_ERR_INVALID, "projected triplet attention: x / w must be 16-byte aligned");
    return a->dtype == TGT_BF16 ? launch_proj<bf16_t>(*a, x, w, bias, st) : launch_proj<f16_t>(*a, x, w, bias, st);
}

}  // namespace tgt
