// Triplet aggregate core (TripletAggregate / TripletAggregateUngated) for
// gfx950 -- forward and backward.
//
// Replaces reference lib/tgt/layers/triplet.py:56-70 and :107-123: the
// softmax*gate over the third arm followed by einsum('bikh,bjkdh->bijdh') /
// einsum('bkih,bkjdh->bijdh'), and their autograd backward.  SURVEY App. A.3.
//
// Same decomposition as triplet_attention.hip (workgroup = (graph, direction,
// head group), wave = head, walk over the shared node j), but the weights
// A[i,k] do not depend on j: each wave computes its 32x32 weight tile ONCE
// from E/G/M, keeps it in registers as matrix-core operand fragments, and per
// j only streams the V slab:   O^T[d][i] = sum_k V^T[d][k] A^T[k][i].
// Backward: dA^T[k][i] = sum_j V[j,k,:].dO[i,j,:] accumulates across the walk
// inside the MFMA accumulator; dV^T[d][k] = sum_i dO^T[d][i] A[i][k] per j.
#include "triplet_common.hpp"

namespace tgt {

struct AggCtx {
    int b, dir, g, h, N;
    ThirdArm ta;
};

template <typename T, int D, int HG>
__device__ __forceinline__ AggCtx agg_ctx(const tgt_triplet_aggregate_args& a, int wave) {
    AggCtx c;
    const int ngroups = a.H / HG;
    int bid = blockIdx.x;
    c.g = bid % ngroups;
    bid /= ngroups;
    c.dir = bid & 1;
    c.b = bid >> 1;
    c.h = c.g * HG + wave;
    c.N = a.N;
    const bool use_mask = c.dir == 0 || (a.flags & TGT_TRI_MASK_OUT);
    c.ta = ThirdArm{a.eg[c.dir], a.ld_eg[c.dir], a.e_off[c.dir], a.g_off[c.dir], use_mask ? a.mask : nullptr,
                    true, (a.flags & TGT_TRI_GATED) != 0};
    return c;
}

// buffer-addressed slabs (triplet_common.hpp): rows k of V[j,k] (inward) / V[k,j] (outward) of this head group
template <typename T, int D, int HG>
__device__ __forceinline__ SlabBuf agg_v_slab(const void* tensor, int64_t ld, int off, const AggCtx& c) {
    const int64_t sz = sizeof(T), N = c.N;
    const uint32_t ldb = (uint32_t)(ld * sz);
    return SlabBuf{graph_rsrc(tensor, N * N * ld * sz, c.b), (uint32_t)((off + c.g * HG * D) * sz),
                   c.dir == 0 ? ldb : (uint32_t)N * ldb, c.dir == 0 ? (uint32_t)N * ldb : ldb};
}
// rows i of X[i,j] (the aggregate's output and its gradient)
template <typename T, int D, int HG>
__device__ __forceinline__ SlabBuf agg_o_slab(const void* tensor, int64_t ld, int off, const AggCtx& c) {
    const int64_t sz = sizeof(T), N = c.N;
    const uint32_t ldb = (uint32_t)(ld * sz);
    return SlabBuf{graph_rsrc(tensor, N * N * ld * sz, c.b), (uint32_t)((off + c.g * HG * D) * sz), (uint32_t)N * ldb, ldb};
}

// weights of query tile it / key tile kt in (lane = i) layout: softmax over ALL key tiles
// (the E/G/mask tiles come through the LDS stage: two barriers inside, `lds` aliases the slabs)
template <typename T, int HG, int NT, bool PAD>
__device__ __forceinline__ void agg_weights(const AggCtx& c, int N, int wave, int tid, int r, int hi, int i0, char* lds,
                                            float (&p)[NT][16], float (&gate)[NT][16]) {
    arm_stage_load<T, HG, NT>(c.ta, c.b, c.dir, c.g, N, i0, lds, tid);
    __syncthreads();
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
        arm_stage_read<T, HG, NT, PAD>(c.ta, lds, c.dir, wave, N, r, hi, i0, kt, p[kt], gate[kt]);
    __syncthreads();
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int q = 0; q < 16; ++q) mx = fmaxf(mx, p[kt][q]);
    mx = fmaxf(mx, xhalf(mx));
    if (mx == -INFINITY) mx = 0.f;               // padding column: all weights exactly 0
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            p[kt][q] = fast_exp(p[kt][q] - mx);
            sum += p[kt][q];
        }
    sum += xhalf(sum);
    const float inv = sum > 0.f ? fast_rcp(sum) : 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int q = 0; q < 16; ++q) p[kt][q] *= inv;
}

template <typename T, int D, int HG, int NT>
__global__ void __launch_bounds__(HG * 64) tri_agg_fwd_kernel(const tgt_triplet_aggregate_args a) {
    using G = TriGeo<T, D, HG>;
    using F = frag_t<T>;
    constexpr int KR = 32 * NT;
    // two LDS sets {V | O}: set j&1 is computed on while slab j+1 lands in the other one -> ONE barrier per j
    // (hazards as in triplet_attention.hip)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int kSet = (NT + 1) * G::kSlabBytes;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hi = lane >> 5;
    const AggCtx c = agg_ctx<T, D, HG>(a, wave);
    const int N = c.N;
    F ident_d[G::kDC];
    make_ident_d<T, G::kDC>(ident_d, r, hi);
    const TriDrop drop = tri_drop(a.dropout_p, a.dropout_seed);
    const uint32_t drop_unit = (uint32_t)((c.b * 2 + c.dir) * a.H + c.h);
    const SlabBuf bV = agg_v_slab<T, D, HG>(a.v[c.dir], a.ld_v[c.dir], a.v_off[c.dir], c);
    const SlabBuf bO = agg_o_slab<T, D, HG>(a.out, a.ld_out, a.o_off[c.dir], c);

    for (int it = 0; it < NT; ++it) {
        const int i0 = 32 * it;
        if (i0 >= N) break;
        F pa[NT][2];
        {
            float p[NT][16], gate[NT][16];
            agg_weights<T, HG, NT, false>(c, N, wave, tid, r, hi, i0, smem, p, gate);
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                f32x16 w;
#pragma unroll
                for (int q = 0; q < 16; ++q) w[q] = p[kt][q] * gate[kt][q];
                if (drop.on) {
                    const uint32_t keep = tri_drop_bits(drop, drop_unit, i0 + r, kt, hi);
#pragma unroll
                    for (int q = 0; q < 16; ++q) w[q] = (keep >> q) & 1u ? w[q] * drop.scale : 0.f;
                }
                pa[kt][0] = pack_chunk<T>(w, 0);
                pa[kt][1] = pack_chunk<T>(w, 1);
            }
        }
        uint4 pv[SlabIO<G, KR>::kIters];
        slab_issue<G, KR>(pv, bV, 0, 0, N, tid);
        slab_commit<G, KR>(pv, smem, tid);
        if (N > 1) slab_issue<G, KR>(pv, bV, 1, 0, N, tid);
        __syncthreads();
        for (int j = 0; j < N; ++j) {
            char* sV = smem + (j & 1) * kSet;
            char* sOut = sV + NT * G::kSlabBytes;
            if (j + 1 < N) slab_commit<G, KR>(pv, smem + ((j + 1) & 1) * kSet, tid);
            if (j + 2 < N) slab_issue<G, KR>(pv, bV, j + 2, 0, N, tid);
            f32x16 o = {0};
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                F fv[G::kDC];
                read_frags<T, D, HG>(fv, sV, wave, 32 * kt + r, hi);
                f32x16 vt = {0};
#pragma unroll
                for (int dc = 0; dc < G::kDC; ++dc) vt = mma32(fv[dc], ident_d[dc], vt);
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) o = mma32(pack_chunk<T>(vt, cc), pa[kt][cc], o);
            }
            write_rows<T, D, HG>(sOut, o, wave, r, hi);
            __syncthreads();
            slab_store<G, 32>(sOut, bO, j, i0, N, tid);
        }
        __syncthreads();      // the next query-tile pass re-fills the stage and both sets
    }
}

template <typename T, int D, int HG, int NT>
__global__ void __launch_bounds__(HG * 64) tri_agg_bwd_kernel(const tgt_triplet_aggregate_args a) {
    using G = TriGeo<T, D, HG>;
    using F = frag_t<T>;
    constexpr int KR = 32 * NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int kSet = (NT + 1) * G::kSlabBytes;        // {dO | V}, two sets (see forward)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hi = lane >> 5;
    const AggCtx c = agg_ctx<T, D, HG>(a, wave);
    const int N = c.N;

    F ident_d[G::kDC];
    make_ident_d<T, G::kDC>(ident_d, r, hi);
    const TriDrop drop = tri_drop(a.dropout_p, a.dropout_seed);
    const uint32_t drop_unit = (uint32_t)((c.b * 2 + c.dir) * a.H + c.h);
    const SlabBuf bV = agg_v_slab<T, D, HG>(a.v[c.dir], a.ld_v[c.dir], a.v_off[c.dir], c);
    const SlabBuf dV = agg_v_slab<T, D, HG>(a.d_v[c.dir], a.ld_v[c.dir], a.v_off[c.dir], c);      // d_v mirrors v
    const SlabBuf dO = agg_o_slab<T, D, HG>(a.d_out, a.ld_out, a.o_off[c.dir], c);

    for (int it = 0; it < NT; ++it) {
        const int i0 = 32 * it;
        if (i0 >= N) break;
        // weights in (lane = k, registers = i) layout, as operand fragments over i
        F a2f[NT][2];
        {
            float p[NT][16], gate[NT][16];
            agg_weights<T, HG, NT, true>(c, N, wave, tid, r, hi, i0, smem, p, gate);
            F ident_k[2];
            make_ident_k<T>(ident_k, r, hi);
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                f32x16 w, a2 = {0};
#pragma unroll
                for (int q = 0; q < 16; ++q) w[q] = p[kt][q] * gate[kt][q];
                if (drop.on) {
                    const uint32_t keep = tri_drop_bits(drop, drop_unit, i0 + r, kt, hi);
#pragma unroll
                    for (int q = 0; q < 16; ++q) w[q] = (keep >> q) & 1u ? w[q] * drop.scale : 0.f;
                }
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) a2 = mma32(pack_chunk<T>(w, cc), ident_k[cc], a2);
                a2f[kt][0] = pack_chunk<T>(a2, 0);
                a2f[kt][1] = pack_chunk<T>(a2, 1);
            }
        }
        f32x16 dacc[NT];          // dA^T[k][i], summed over j
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int q = 0; q < 16; ++q) dacc[kt][q] = 0.f;

        uint4 pv[SlabIO<G, KR>::kIters], po[SlabIO<G, 32>::kIters];
        slab_issue<G, 32>(po, dO, 0, i0, N, tid);
        slab_issue<G, KR>(pv, bV, 0, 0, N, tid);
        slab_commit<G, 32>(po, smem, tid);
        slab_commit<G, KR>(pv, smem + G::kSlabBytes, tid);
        if (N > 1) {
            slab_issue<G, 32>(po, dO, 1, i0, N, tid);
            slab_issue<G, KR>(pv, bV, 1, 0, N, tid);
        }
        __syncthreads();
        for (int j = 0; j < N; ++j) {
            char* sO = smem + (j & 1) * kSet;
            char* sV = sO + G::kSlabBytes;
            if (j + 1 < N) {
                char* nO = smem + ((j + 1) & 1) * kSet;
                slab_commit<G, 32>(po, nO, tid);
                slab_commit<G, KR>(pv, nO + G::kSlabBytes, tid);
            }
            if (j + 2 < N) {
                slab_issue<G, 32>(po, dO, j + 2, i0, N, tid);
                slab_issue<G, KR>(pv, bV, j + 2, 0, N, tid);
            }
            // partial dV the previous query-tile pass stored for THIS j (added at the store below)
            uint4 curv[NT > 1 ? SlabIO<G, KR>::kIters : 1];
            if constexpr (NT > 1) {
                if (it > 0) slab_issue<G, KR>(curv, dV, j, 0, N, tid);
            }
            F fo[G::kDC];
            read_frags<T, D, HG>(fo, sO, wave, r, hi);
            f32x16 t2 = {0};
#pragma unroll
            for (int dc = 0; dc < G::kDC; ++dc) t2 = mma32(fo[dc], ident_d[dc], t2);   // dO^T
            F oTf[2] = {pack_chunk<T>(t2, 0), pack_chunk<T>(t2, 1)};
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                F fv[G::kDC];
                read_frags<T, D, HG>(fv, sV, wave, 32 * kt + r, hi);
#pragma unroll
                for (int dc = 0; dc < G::kDC; ++dc) dacc[kt] = mma32(fv[dc], fo[dc], dacc[kt]);
                f32x16 dv = {0};
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) dv = mma32(oTf[cc], a2f[kt][cc], dv);
                // (in place: this wave only reads its own columns, and only of rows it has not overwritten yet)
                write_rows<T, D, HG>(sV, dv, wave, 32 * kt + r, hi);
            }
            __syncthreads();
            bool plain = true;
            if constexpr (NT > 1) {
                if (it > 0) {
                    plain = false;
                    slab_store_add<G, KR, T>(sV, curv, dV, j, 0, N, tid);
                }
            }
            if (plain) slab_store<G, KR>(sV, dV, j, 0, N, tid);
        }
        __syncthreads();      // the stage below aliases both sets

        // softmax*gate backward on the accumulated dA (recompute P, g)
        float p[NT][16], gate[NT][16];
        agg_weights<T, HG, NT, true>(c, N, wave, tid, r, hi, i0, smem, p, gate);
        float delta = 0.f;
        float dG[NT][16];
        if (drop.on) {            // dacc is the gradient of the DROPPED weights
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                const uint32_t keep = tri_drop_bits(drop, drop_unit, i0 + r, kt, hi);
#pragma unroll
                for (int q = 0; q < 16; ++q) dacc[kt][q] = (keep >> q) & 1u ? dacc[kt][q] * drop.scale : 0.f;
            }
        }
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                dG[kt][q] = dacc[kt][q] * p[kt][q] * gate[kt][q] * (1.f - gate[kt][q]);
                dacc[kt][q] *= gate[kt][q];
                delta += p[kt][q] * dacc[kt][q];
            }
        delta += xhalf(delta);
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            float dE[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) dE[q] = p[kt][q] * (dacc[kt][q] - delta);
            arm_stage_put_grad<T, HG, NT>(smem, c.dir, wave, r, hi, kt, dE, dG[kt]);
        }
        __syncthreads();
        arm_stage_store_grad<T, HG, NT>(c.ta, a.d_eg[c.dir], c.b, c.dir, c.g, N, i0, smem, tid);
        __syncthreads();
    }
}

template <typename T, int D, int HG, int NT>
static int launch_agg_nt(const tgt_triplet_aggregate_args& a, bool bwd, hipStream_t st) {
    using G = TriGeo<T, D, HG>;
    const int grid = a.B * 2 * (a.H / HG);
    constexpr int kArm = ArmStage<T, HG, NT>::kBytes;
    constexpr int kLds = 2 * (NT + 1) * G::kSlabBytes > kArm ? 2 * (NT + 1) * G::kSlabBytes : kArm;
    if (!bwd) hipLaunchKernelGGL((tri_agg_fwd_kernel<T, D, HG, NT>), dim3(grid), dim3(G::kThreads), kLds, st, a);
    else      hipLaunchKernelGGL((tri_agg_bwd_kernel<T, D, HG, NT>), dim3(grid), dim3(G::kThreads), kLds, st, a);
    return check_launch(bwd ? "tri_agg_bwd_kernel" : "tri_agg_fwd_kernel");
}
template <typename T, int D, int HG>
static int launch_agg(const tgt_triplet_aggregate_args& a, bool bwd, hipStream_t st) {
    if (a.N <= 32) return launch_agg_nt<T, D, HG, 1>(a, bwd, st);
    return launch_agg_nt<T, D, HG, 2>(a, bwd, st);
}
template <typename T, int D>
static int agg_dispatch_hg(const tgt_triplet_aggregate_args& a, bool bwd, hipStream_t st) {
    if constexpr (D == 16 && sizeof(T) == 2) {
        if (a.H % 8 == 0 && a.N <= 32) return launch_agg_nt<T, D, 8, 1>(a, bwd, st);     // 256-byte row pieces
    }
    if (a.H % 4 == 0) return launch_agg<T, D, 4>(a, bwd, st);
    if constexpr (D * sizeof(T) >= 16) return launch_agg<T, D, 1>(a, bwd, st);
    return set_error(TGT_ERR_UNSUPPORTED, "triplet aggregate: H=%d not a multiple of 4 with D=%d", a.H, D);
}
template <typename T>
static int agg_dispatch_d(const tgt_triplet_aggregate_args& a, bool bwd, hipStream_t st) {
    switch (a.D) {
        case 8: return agg_dispatch_hg<T, 8>(a, bwd, st);
        case 16: return agg_dispatch_hg<T, 16>(a, bwd, st);
        case 32: return agg_dispatch_hg<T, 32>(a, bwd, st);
        default: return set_error(TGT_ERR_UNSUPPORTED, "triplet aggregate: D=%d not in {8,16,32}", a.D);
    }
}

int triplet_aggregate_run(const tgt_triplet_aggregate_args* a, bool bwd, hipStream_t st) {
    if (!a) return set_error(TGT_ERR_INVALID, "triplet aggregate: null args");
    if (a->B < 0 || a->N < 0 || a->H <= 0) return set_error(TGT_ERR_INVALID, "triplet aggregate: bad sizes");
    if (a->B == 0 || a->N == 0) return TGT_OK;
    if (a->N > 64) return set_error(TGT_ERR_UNSUPPORTED, "triplet aggregate: N=%d > 64 not supported", a->N);
    const int64_t esz = a->dtype == TGT_F32 ? 4 : 2;
    for (int dir = 0; dir < 2; ++dir) {
        if (!a->v[dir] || !a->eg[dir] || !a->out || !a->mask) return set_error(TGT_ERR_INVALID, "triplet aggregate: null tensor");
        if ((a->ld_v[dir] * esz) % 16 || (a->v_off[dir] * esz) % 16 || (a->ld_out * esz) % 16 ||
            (a->o_off[dir] * esz) % 16 || ((uintptr_t)a->v[dir] % 16) || ((uintptr_t)a->out % 16))
            return set_error(TGT_ERR_INVALID, "triplet aggregate: rows/offsets must be 16-byte aligned");
        if (bwd && (!a->d_out || !a->d_v[dir] || !a->d_eg[dir] || ((uintptr_t)a->d_v[dir] % 16) || ((uintptr_t)a->d_out % 16)))
            return set_error(TGT_ERR_INVALID, "triplet aggregate bwd: null/misaligned gradient tensor");
    }
    switch (a->dtype) {
        case TGT_F32: return agg_dispatch_d<float>(*a, bwd, st);
        case TGT_BF16: return agg_dispatch_d<bf16_t>(*a, bwd, st);
        case TGT_F16: return agg_dispatch_d<f16_t>(*a, bwd, st);
        default: return set_error(TGT_ERR_INVALID, "triplet aggregate: bad dtype %d", a->dtype);
    }
}

}  // namespace tgt
