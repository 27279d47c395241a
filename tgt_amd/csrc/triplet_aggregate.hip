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
    SlabSrc v;        // rows k of V[j,k] (inward) / V[k,j] (outward)
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
    const int64_t N = a.N, ld = a.ld_v[c.dir], sz = sizeof(T);
    const char* base = reinterpret_cast<const char*>(a.v[c.dir]) +
                       ((int64_t)c.b * N * N * ld + a.v_off[c.dir] + c.g * HG * D) * sz;
    if (c.dir == 0) c.v = {base, ld * sz, N * ld * sz};
    else            c.v = {base, N * ld * sz, ld * sz};
    const bool use_mask = c.dir == 0 || (a.flags & TGT_TRI_MASK_OUT);
    c.ta = ThirdArm{a.eg[c.dir], a.ld_eg[c.dir], a.e_off[c.dir], a.g_off[c.dir], use_mask ? a.mask : nullptr,
                    true, (a.flags & TGT_TRI_GATED) != 0};
    return c;
}

// softmax over k of biasM (column i of this lane), in place -> P
__device__ __forceinline__ void column_softmax(float (&x)[16]) {
    float mx = -INFINITY;
#pragma unroll
    for (int q = 0; q < 16; ++q) mx = fmaxf(mx, x[q]);
    mx = fmaxf(mx, xhalf(mx));
    if (mx == -INFINITY) mx = 0.f;               // padding column: all weights exactly 0
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        x[q] = fast_exp(x[q] - mx);
        sum += x[q];
    }
    sum += xhalf(sum);
    const float inv = sum > 0.f ? __frcp_rn(sum) : 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) x[q] *= inv;
}

template <typename T, int D, int HG>
__global__ void __launch_bounds__(HG * 64) tri_agg_fwd_kernel(const tgt_triplet_aggregate_args a) {
    using G = TriGeo<T, D, HG>;
    using F = frag_t<T>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sV = smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hi = lane >> 5;
    const AggCtx c = agg_ctx<T, D, HG>(a, wave);
    const int N = c.N;

    F pa[2];
    {
        float p[16], gate[16];
        load_third_arm<T, false>(c.ta, c.b, c.dir, c.h, N, r, hi, p, gate);
        column_softmax(p);
        f32x16 w;
#pragma unroll
        for (int q = 0; q < 16; ++q) w[q] = p[q] * gate[q];
        pa[0] = pack_chunk<T>(w, 0);
        pa[1] = pack_chunk<T>(w, 1);
    }
    F ident_d[G::kDC];
    make_ident_d<T, G::kDC>(ident_d, r, hi);

    const int64_t sz = sizeof(T);
    char* obase = reinterpret_cast<char*>(a.out) + ((int64_t)c.b * N * N * a.ld_out + a.o_off[c.dir] + c.g * HG * D) * sz;
    const int64_t o_row = (int64_t)N * a.ld_out * sz, o_j = a.ld_out * sz;

    uint4 pv[G::kIters];
    slab_issue<G>(pv, c.v, 0, N, tid);
    slab_commit<G>(pv, sV, tid);
    __syncthreads();
    for (int j = 0; j < N; ++j) {
        if (j + 1 < N) slab_issue<G>(pv, c.v, j + 1, N, tid);
        F fv[G::kDC];
        read_frags<T, D, HG>(fv, sV, wave, r, hi);
        f32x16 vt = {0}, o = {0};
#pragma unroll
        for (int dc = 0; dc < G::kDC; ++dc) vt = mma32(fv[dc], ident_d[dc], vt);
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) o = mma32(pack_chunk<T>(vt, cc), pa[cc], o);
        write_rows<T, D, HG>(sV, o, wave, r, hi);
        __syncthreads();
        slab_store<G>(sV, obase, o_row, o_j, j, N, tid);
        if (j + 1 < N) slab_commit<G>(pv, sV, tid);
        __syncthreads();
    }
}

template <typename T, int D, int HG>
__global__ void __launch_bounds__(HG * 64) tri_agg_bwd_kernel(const tgt_triplet_aggregate_args a) {
    using G = TriGeo<T, D, HG>;
    using F = frag_t<T>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sV = smem;
    char* sO = smem + G::kSlabBytes;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hi = lane >> 5;
    const AggCtx c = agg_ctx<T, D, HG>(a, wave);
    const int N = c.N;

    F ident_d[G::kDC];
    make_ident_d<T, G::kDC>(ident_d, r, hi);
    // weights in (lane = k, registers = i) layout, as operand fragments over i
    F a2f[2];
    {
        float p[16], gate[16];
        load_third_arm<T, true>(c.ta, c.b, c.dir, c.h, N, r, hi, p, gate);
        column_softmax(p);
        f32x16 w;
#pragma unroll
        for (int q = 0; q < 16; ++q) w[q] = p[q] * gate[q];
        F ident_k[2];
        make_ident_k<T>(ident_k, r, hi);
        f32x16 a2 = {0};
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) a2 = mma32(pack_chunk<T>(w, cc), ident_k[cc], a2);
        a2f[0] = pack_chunk<T>(a2, 0);
        a2f[1] = pack_chunk<T>(a2, 1);
    }

    const int64_t sz = sizeof(T), Nl = N;
    const SlabSrc dO = {reinterpret_cast<const char*>(a.d_out) +
                            ((int64_t)c.b * Nl * Nl * a.ld_out + a.o_off[c.dir] + c.g * HG * D) * sz,
                        Nl * a.ld_out * sz, a.ld_out * sz};
    const int64_t shift = reinterpret_cast<const char*>(a.d_v[c.dir]) - reinterpret_cast<const char*>(a.v[c.dir]);
    char* dv_base = const_cast<char*>(c.v.base) + shift;

    f32x16 dacc = {0};          // dA^T[k][i], summed over j
    uint4 pv[G::kIters], po[G::kIters];
    slab_issue<G>(pv, c.v, 0, N, tid);
    slab_issue<G>(po, dO, 0, N, tid);
    slab_commit<G>(pv, sV, tid);
    slab_commit<G>(po, sO, tid);
    __syncthreads();
    for (int j = 0; j < N; ++j) {
        if (j + 1 < N) {
            slab_issue<G>(pv, c.v, j + 1, N, tid);
            slab_issue<G>(po, dO, j + 1, N, tid);
        }
        F fv[G::kDC], fo[G::kDC];
        read_frags<T, D, HG>(fv, sV, wave, r, hi);
        read_frags<T, D, HG>(fo, sO, wave, r, hi);
#pragma unroll
        for (int dc = 0; dc < G::kDC; ++dc) dacc = mma32(fv[dc], fo[dc], dacc);
        f32x16 t2 = {0}, dv = {0};
#pragma unroll
        for (int dc = 0; dc < G::kDC; ++dc) t2 = mma32(fo[dc], ident_d[dc], t2);   // dO^T
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) dv = mma32(pack_chunk<T>(t2, cc), a2f[cc], dv);
        write_rows<T, D, HG>(sV, dv, wave, r, hi);
        __syncthreads();
        slab_store<G>(sV, dv_base, c.v.row_stride, c.v.j_stride, j, N, tid);
        if (j + 1 < N) {
            slab_commit<G>(pv, sV, tid);
            slab_commit<G>(po, sO, tid);
        }
        __syncthreads();
    }

    // softmax*gate backward on the accumulated dA (recompute P, g)
    float p[16], gate[16], dE[16], dG[16];
    load_third_arm<T, true>(c.ta, c.b, c.dir, c.h, N, r, hi, p, gate);
    column_softmax(p);
    float delta = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        dG[q] = dacc[q] * p[q] * gate[q] * (1.f - gate[q]);
        dacc[q] *= gate[q];
        delta += p[q] * dacc[q];
    }
    delta += xhalf(delta);
#pragma unroll
    for (int q = 0; q < 16; ++q) dE[q] = p[q] * (dacc[q] - delta);
    store_third_arm_grad<T>(c.ta, a.d_eg[c.dir], c.b, c.dir, c.h, N, r, hi, dE, dG);
}

template <typename T, int D, int HG>
static int launch_agg(const tgt_triplet_aggregate_args& a, bool bwd, hipStream_t st) {
    using G = TriGeo<T, D, HG>;
    const int grid = a.B * 2 * (a.H / HG);
    if (!bwd) hipLaunchKernelGGL((tri_agg_fwd_kernel<T, D, HG>), dim3(grid), dim3(G::kThreads), G::kSlabBytes, st, a);
    else      hipLaunchKernelGGL((tri_agg_bwd_kernel<T, D, HG>), dim3(grid), dim3(G::kThreads), 2 * G::kSlabBytes, st, a);
    return check_launch(bwd ? "tri_agg_bwd_kernel" : "tri_agg_fwd_kernel");
}
template <typename T, int D>
static int agg_dispatch_hg(const tgt_triplet_aggregate_args& a, bool bwd, hipStream_t st) {
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
    if (a->B <= 0 || a->N <= 0 || a->H <= 0) return set_error(TGT_ERR_INVALID, "triplet aggregate: bad sizes");
    if (a->N > 32) return set_error(TGT_ERR_UNSUPPORTED, "triplet aggregate: N=%d > 32 not supported yet", a->N);
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
