// Triplet attention core for gfx950 -- forward and backward.
//
// Replaces the two einsum -> +bias -> +mask -> softmax -> *gate -> einsum
// chains of reference lib/tgt/layers/triplet.py:213-246 (and the autograd
// backward of that chain).  Math: SURVEY.md App. A.2 / A.4.
//
// Decomposition (one kernel serves both directions):
//   workgroup = (graph b, direction, group of HG heads); wave w = head g*HG+w.
//   The workgroup walks the shared node j = 0..N-1.  For each j it needs three
//   "slabs": rows Q[i,j], i=0..N-1; the partner rows K/V[j,k] (inward) or
//   K/V[k,j] (outward), k=0..N-1.  Only this group's HG*D channels of each
//   row are touched (HG*D*sizeof(T) >= 128 B for the shipped shapes, i.e.
//   whole cache lines).  Slabs for j+1 are fetched HBM->registers while the
//   waves work on j out of LDS (one LDS buffer; the prefetch lives in VGPRs).
//   The third-arm bias E, gate sigmoid(G+M) and mask M do not depend on j:
//   each wave keeps its 32x32 (i,k) tile of them in registers for the whole
//   walk, and in the backward accumulates dE/dG over j in registers --
//   no atomics, no workspace, deterministic.
//
// Matrix-core mapping (N <= 32 per tile, Dt = 16 -> one 32x32x16 MFMA):
//   S^T[k][i]  = K[k,:] . Q[i,:]          A = K rows, B = Q rows (both read
//                                         as natural d-contiguous fragments)
//   lane (i = l&31, hi = l>>5) then owns column i of S^T: 16 of the 32 k's;
//   its partner lane l^32 owns the other 16 -> softmax over k is an in-lane
//   reduction plus ONE cross-lane exchange.
//   V^T        = V . I                    transposes V through the matrix core
//                                         (exact: x*1 + 0), giving lane d the
//                                         k-vector the P.V product needs
//   O^T[d][i]  = sum_k V^T[d][k] P^T[k][i]
//   The register->row map of an MFMA result equals the k-order of the next
//   MFMA's operand fragment, so no cross-lane data movement is needed at all.
//   Backward uses the same trick to re-layout dS and A (multiply by I).
#include <cstdlib>
#include "triplet_common.hpp"

namespace tgt {

// ---------------------------------------------------------------------------
// forward.  NT = node tiles of 32 (N <= 32*NT).  The workgroup makes one pass
// per query tile `it`; inside a pass the key axis spans all NT tiles.
// ---------------------------------------------------------------------------
template <typename T, int D, int HG, int NT, int PF, bool DROP>
__global__ void __launch_bounds__(HG * 64, (NT == 1 && sizeof(T) == 2) ? 4 : (sizeof(T) == 2 ? 2 : 1)) tri_att_fwd_kernel(const tgt_triplet_attention_args a) {
    using G = TriGeo<T, D, HG>;
    using F = frag_t<T>;
    constexpr int KR = 32 * NT;
    // two LDS sets {Q | K | V}: set j&1 is computed on while j+1 lands in the other one,
    // so ONE barrier per j suffices (see the hazard notes at the loop).
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int kSet = (1 + 2 * NT) * G::kSlabBytes;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hi = lane >> 5;
    const TriCtx c = tri_ctx<T, D, HG>(a, wave);
    const int N = c.N;
    const ThirdArm ta = tri_third_arm(a, c.dir);
    F ident_d[G::kDC];
    make_ident_d<T, G::kDC>(ident_d, r, hi);
    const TriDrop drop = tri_drop(a.dropout_p, a.dropout_seed);
    const uint32_t drop_unit0 = (uint32_t)(((c.b * 2 + c.dir) * a.H + c.h) * N);

    // buffer-addressed slabs (triplet_common.hpp): scalar offsets along j, no vector address arithmetic
    const int64_t sz = sizeof(T), Nl = N;
    const uint32_t hch = (uint32_t)(c.g * HG * D * sz), lds_ = (uint32_t)(a.ld_qkv[c.dir] * sz), ldo_ = (uint32_t)(a.ld_out * sz);
    const __amdgpu_buffer_rsrc_t r_src = graph_rsrc(a.qkv[c.dir], Nl * Nl * a.ld_qkv[c.dir] * sz, c.b);
    const SlabBuf bQ = {r_src, (uint32_t)(a.q_off[c.dir] * sz) + hch, (uint32_t)N * lds_, lds_};
    const SlabBuf bK = {r_src, (uint32_t)(a.k_off[c.dir] * sz) + hch, c.dir == 0 ? lds_ : (uint32_t)N * lds_,
                        c.dir == 0 ? (uint32_t)N * lds_ : lds_};
    const SlabBuf bV = {r_src, (uint32_t)(a.v_off[c.dir] * sz) + hch, bK.row_stride, bK.j_stride};
    const SlabBuf bO = {graph_rsrc(a.out, Nl * Nl * a.ld_out * sz, c.b), (uint32_t)(a.o_off[c.dir] * sz) + hch, (uint32_t)N * ldo_, ldo_};

    // a graph DropPath dropped (graph_scale[b] == 0): the residual add multiplies this branch by zero, so nothing is
    // read or computed; the rows get the zeros that product would give (workgroup-uniform branch)
    if (a.graph_scale && a.graph_scale[c.b] == 0.f) {
        for (int i0 = 0; i0 < N; i0 += 32)
            for (int j = 0; j < N; ++j) slab_store_zero<G, 32>(bO, j, i0, N, tid);
        return;
    }

    for (int it = 0; it < NT; ++it) {
        const int i0 = 32 * it;
        if (i0 >= N) break;
        float biasM[NT][16], gate[NT][16];
        arm_stage_load<T, HG, NT>(ta, c.b, c.dir, c.g, N, i0, smem, tid);
        __syncthreads();
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
            arm_stage_read<T, HG, NT, false>(ta, smem, c.dir, wave, N, r, hi, i0, kt, biasM[kt], gate[kt]);
        __syncthreads();

        // PF register stages of prefetch: stage A holds slab j+1 (j+2 in B) when iteration j starts
        uint4 pqA[SlabIO<G, 32>::kIters], pkA[SlabIO<G, KR>::kIters], pvA[SlabIO<G, KR>::kIters];
        uint4 pqB[PF > 1 ? SlabIO<G, 32>::kIters : 1], pkB[PF > 1 ? SlabIO<G, KR>::kIters : 1],
            pvB[PF > 1 ? SlabIO<G, KR>::kIters : 1];
        slab_issue<G, 32>(pqA, bQ, 0, i0, N, tid);
        slab_issue<G, KR>(pkA, bK, 0, 0, N, tid);
        slab_issue<G, KR>(pvA, bV, 0, 0, N, tid);
        slab_commit<G, 32>(pqA, smem, tid);
        slab_commit<G, KR>(pkA, smem + G::kSlabBytes, tid);
        slab_commit<G, KR>(pvA, smem + (1 + NT) * G::kSlabBytes, tid);
        if (N > 1) {
            slab_issue<G, 32>(pqA, bQ, 1, i0, N, tid);
            slab_issue<G, KR>(pkA, bK, 1, 0, N, tid);
            slab_issue<G, KR>(pvA, bV, 1, 0, N, tid);
        }
        if constexpr (PF > 1) {
            if (N > 2) {
                slab_issue<G, 32>(pqB, bQ, 2, i0, N, tid);
                slab_issue<G, KR>(pkB, bK, 2, 0, N, tid);
                slab_issue<G, KR>(pvB, bV, 2, 0, N, tid);
            }
        }
        __syncthreads();

        // Hazards with one barrier per j (B_j = the barrier of iteration j):
        //  * set[cur^1] is overwritten at the top of iteration j.  Its K/V were last read by
        //    compute(j-1), finished before B_{j-1}; its Q slab holds O(j-1), which is stored after
        //    B_{j-1} by the SAME thread (same chunk map) that now overwrites that chunk.
        //  * compute(j) reads set[cur], committed before B_{j-1}.
        //  * O(j) goes into this wave's own columns of set[cur].Q and is read after B_j.
        auto step = [&](int j, auto& pq, auto& pk, auto& pv) {
            char* sQ = smem + (j & 1) * kSet;
            char* sK = sQ + G::kSlabBytes;
            char* sV = sK + NT * G::kSlabBytes;
            if (j + 1 < N) {
                char* nQ = smem + ((j + 1) & 1) * kSet;
                slab_commit<G, 32>(pq, nQ, tid);
                slab_commit<G, KR>(pk, nQ + G::kSlabBytes, tid);
                slab_commit<G, KR>(pv, nQ + (1 + NT) * G::kSlabBytes, tid);
            }
            if (j + 1 + PF < N) {
                slab_issue<G, 32>(pq, bQ, j + 1 + PF, i0, N, tid);
                slab_issue<G, KR>(pk, bK, j + 1 + PF, 0, N, tid);
                slab_issue<G, KR>(pv, bV, j + 1 + PF, 0, N, tid);
            }
            F fq[G::kDC];
            read_frags<T, D, HG>(fq, sQ, wave, r, hi);
            f32x16 s[NT], vt[NT];
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                F fk[G::kDC], fv[G::kDC];
                read_frags<T, D, HG>(fk, sK, wave, 32 * kt + r, hi);
                read_frags<T, D, HG>(fv, sV, wave, 32 * kt + r, hi);
                f32x16 z0 = {0}, z1 = {0};
#pragma unroll
                for (int dc = 0; dc < G::kDC; ++dc) z0 = mma32(fk[dc], fq[dc], z0);        // S^T[k][i]
#pragma unroll
                for (int dc = 0; dc < G::kDC; ++dc) z1 = mma32(fv[dc], ident_d[dc], z1);   // V[k][d] -> lane d
                s[kt] = z0;
                vt[kt] = z1;
            }
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    s[kt][q] = s[kt][q] * a.scale + biasM[kt][q];
                    mx = fmaxf(mx, s[kt][q]);
                }
            mx = fmaxf(mx, xhalf(mx));
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    s[kt][q] = fast_exp(s[kt][q] - mx);
                    sum += s[kt][q];
                }
            sum += xhalf(sum);
            const float inv = fast_rcp(sum);
            f32x16 o = {0};
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
                for (int q = 0; q < 16; ++q) s[kt][q] = s[kt][q] * inv * gate[kt][q];
                if constexpr (DROP) {     // attention dropout on the gated weights
                    const uint32_t keep = tri_drop_bits(drop, drop_unit0 + j, i0 + r, kt, hi);
#pragma unroll
                    for (int q = 0; q < 16; ++q) s[kt][q] = (keep >> q) & 1u ? s[kt][q] * drop.scale : 0.f;
                }
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) o = mma32(pack_chunk<T>(vt[kt], cc), pack_chunk<T>(s[kt], cc), o);   // O^T[d][i]
            }
            write_rows<T, D, HG>(sQ, o, wave, r, hi);     // in place of this head's Q columns
            __syncthreads();
            slab_store<G, 32>(sQ, bO, j, i0, N, tid);
        };
        for (int j = 0; j < N; j += PF) {
            step(j, pqA, pkA, pvA);
            if constexpr (PF > 1) {
                if (j + 1 < N) step(j + 1, pqB, pkB, pvB);
            }
        }
        __syncthreads();      // the next query-tile pass re-fills both sets
    }
}

// ---------------------------------------------------------------------------
// backward.  Per (b,dir,h,j), with P recomputed:
//   dA^T[k][i] = V[k,:].dO[i,:]         dP = dA*g      delta_i = sum_k P dP
//   dS = P (dP - delta)                  dE += dS       dG += dA P g (1-g)
//   dQ^T[d][i] = s sum_k K^T[d][k] dS^T[k][i]
//   dK^T[d][k] = s sum_i Q^T[d][i] dS[i][k]     dV^T[d][k] = sum_i dO^T[d][i] A[i][k]
// dS and A are produced in (lane = i) layout and re-laid out to (lane = k)
// by a multiply with the identity on the matrix core.
// NT > 1: one pass per query tile; dK/dV sum over query tiles, so pass `it > 0`
// adds its partial to what pass it-1 stored (same thread wrote that address).
// ---------------------------------------------------------------------------
#ifdef TGT_PROBES
// Probe build only (tools/probes/tri_bwd_probe.py; never in the shipped library): per-segment cycle totals of wave 0 of every
// workgroup, and ablation bits (TGT_TRI_BWD_ABLATE -> _pad0: 1 no loads, 2 no stores, 4 no tile math).
static __device__ unsigned long long g_tri_probe[4096 * 8];
__device__ __forceinline__ unsigned long long probe_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define TGT_PROBE_T(i) do { const unsigned long long t_ = probe_now(); pt[i] += t_ - t_last; t_last = t_; } while (0)
#else
#define TGT_PROBE_T(i) do { } while (0)
#endif

// FL >= 0: the BIASED/GATED flags are compile-time (the hot gated+biased instantiation: no
// per-element selects); FL < 0: read from the arguments.
template <typename T, int D, int HG, int NT, int OCC, bool CS, int FL, bool DROP>
// (waves_per_eu(1,1) for the one-wave variants: lets the allocator park values in the 256 AGPRs instead of scratch)
__global__ void __launch_bounds__(HG * 64, OCC) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) tri_att_bwd_kernel(const tgt_triplet_attention_args a) {
    using G = TriGeo<T, D, HG>;
    using F = frag_t<T>;
    constexpr int KR = 32 * NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int kSet = (2 + 2 * NT) * G::kSlabBytes;       // {Q | dO | K | V}, two sets (see forward)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hi = lane >> 5;
#ifdef TGT_PROBES
    unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last = probe_now();
    const unsigned long long t_begin = t_last;
    const int ablate = a._pad0;
#else
    constexpr int ablate = 0;
#endif
    const TriCtx c = tri_ctx<T, D, HG>(a, wave);
    const int N = c.N;
    const ThirdArm ta = tri_third_arm(a, c.dir);
    const bool biased = FL >= 0 ? (FL & TGT_TRI_BIASED) != 0 : ta.biased, gated = FL >= 0 ? (FL & TGT_TRI_GATED) != 0 : ta.gated;
    const TriDrop drop = tri_drop(a.dropout_p, a.dropout_seed);
    const uint32_t drop_unit0 = (uint32_t)(((c.b * 2 + c.dir) * a.H + c.h) * N);
    constexpr float kLog2e = 1.4426950408889634f;
    const float scale2 = a.scale * kLog2e;          // logits in the log2 domain: exp2 without the per-element multiply
    F ident_d[G::kDC], ident_k[2];
    make_ident_d<T, G::kDC>(ident_d, r, hi);
    make_ident_k<T>(ident_k, r, hi);

    const int64_t sz = sizeof(T), Nl = N;
    // gradient slabs mirror the source slabs (same channel offsets) inside d_qkv, whose rows may be
    // longer than the sources' (ld_dqkv: one fused gradient row for the projection's GEMMs)
    const int64_t ldq = a.ld_dqkv[c.dir] ? a.ld_dqkv[c.dir] : a.ld_qkv[c.dir];
    const int64_t lde = a.ld_deg[c.dir] ? a.ld_deg[c.dir] : a.ld_eg[c.dir];
    // buffer-addressed slabs (see triplet_common.hpp): one resource per tensor and graph
    const uint32_t hch = (uint32_t)(c.g * HG * D * sz);
    const uint32_t lds_ = (uint32_t)(a.ld_qkv[c.dir] * sz), ldg_ = (uint32_t)(ldq * sz), ldo_ = (uint32_t)(a.ld_out * sz);
    const __amdgpu_buffer_rsrc_t r_src = graph_rsrc(a.qkv[c.dir], Nl * Nl * a.ld_qkv[c.dir] * sz, c.b);
    const __amdgpu_buffer_rsrc_t r_grd = graph_rsrc(a.d_qkv[c.dir], Nl * Nl * ldq * sz, c.b);
    const __amdgpu_buffer_rsrc_t r_do = graph_rsrc(a.d_out, Nl * Nl * a.ld_out * sz, c.b);
    const uint32_t qo = (uint32_t)(a.q_off[c.dir] * sz) + hch, ko = (uint32_t)(a.k_off[c.dir] * sz) + hch,
                   vo = (uint32_t)(a.v_off[c.dir] * sz) + hch;
    // Q-type rows (i, j): row stride N*ld, j stride ld;  partner rows (j,k) inward / (k,j) outward
    const SlabBuf bQ = {r_src, qo, (uint32_t)N * lds_, lds_};
    const SlabBuf bK = {r_src, ko, c.dir == 0 ? lds_ : (uint32_t)N * lds_, c.dir == 0 ? (uint32_t)N * lds_ : lds_};
    const SlabBuf bV = {r_src, vo, bK.row_stride, bK.j_stride};
    const SlabBuf dO = {r_do, (uint32_t)(a.o_off[c.dir] * sz) + hch, (uint32_t)N * ldo_, ldo_};
    const SlabBuf gQ = {r_grd, qo, (uint32_t)N * ldg_, ldg_};
    const SlabBuf dK = {r_grd, ko, c.dir == 0 ? ldg_ : (uint32_t)N * ldg_, c.dir == 0 ? (uint32_t)N * ldg_ : ldg_};
    const SlabBuf dV = {r_grd, vo, dK.row_stride, dK.j_stride};
    ThirdArm dta = ta;                      // third-arm gradient rows
    dta.ld = lde;

    // optional per-graph column sums of dQ/dK/dV and dE/dG (= bias gradients of the projection):
    // per-thread fp32 accumulators (3 planes: dQ, dK, dV) in LDS behind the slab sets
    constexpr int kPlane = slab_colsum_plane_floats<G, T>();
    constexpr int kArmB = ArmStage<T, HG, NT>::kBytes;
    float* cs = reinterpret_cast<float*>(smem + (2 * kSet > kArmB ? 2 * kSet : kArmB));
    if constexpr (CS)
        for (int t = tid; t < 3 * kPlane + HG * 64; t += HG * 64) cs[t] = 0.f;     // + one dE/dG partial per thread

    // a graph DropPath dropped (graph_scale[b] == 0) receives an all-zero d_out: every gradient of it is exactly zero.
    // Nothing is read or computed; zeros go to its dQ / dK / dV / dE / dG rows, and the column sums below stay zero.
    const bool dead = a.graph_scale && a.graph_scale[c.b] == 0.f;          // workgroup-uniform
    if (dead) {
        for (int it = 0; it < NT; ++it) {
            const int i0 = 32 * it;
            if (i0 >= N) break;
            for (int j = 0; j < N; ++j) {
                slab_store_zero<G, 32>(gQ, j, i0, N, tid);
                if (it == 0) {
                    slab_store_zero<G, KR>(dK, j, 0, N, tid);
                    slab_store_zero<G, KR>(dV, j, 0, N, tid);
                }
            }
            const float zero16[16] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            __syncthreads();
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) arm_stage_put_grad<T, HG, NT>(smem, c.dir, wave, r, hi, kt, zero16, zero16);
            __syncthreads();
            arm_stage_store_grad<T, HG, NT>(dta, a.d_eg[c.dir], c.b, c.dir, c.g, N, i0, smem, tid);
            __syncthreads();
        }
    } else
    for (int it = 0; it < NT; ++it) {
        const int i0 = 32 * it;
        if (i0 >= N) break;
        // The 32 x 32 tile of a wave is 16 values per lane; all element-wise arithmetic below runs on PAIRS (f32x2: v_pk_fma_f32 /
        // v_pk_mul_f32 / v_pk_add_f32 -- two results per VALU issue).  Written on scalars hipcc packed 28 of ~190 operations of
        // the j-loop; the loop is 2 waves per SIMD in lockstep behind one barrier per j, so its VALU phase is issue-bound.
        f32x2 biasM[NT][8], gate[NT][8], dE[NT][8], dG[NT][8];
        arm_stage_load<T, HG, NT>(ta, c.b, c.dir, c.g, N, i0, smem, tid);
        __syncthreads();
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            float b16[16], g16[16];
            arm_stage_read<T, HG, NT, true>(ta, smem, c.dir, wave, N, r, hi, i0, kt, b16, g16);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                // log2 domain.  A masked entry is finfo.min (-3.4e38): times log2(e) it would overflow to
                // -inf and a fully masked row would lose its (reference) uniform softmax -- clamp it
                // back to finfo.min; real -inf (padding past N) stays -inf.
                const float bl = b16[q] * kLog2e;
                b16[q] = (bl == -INFINITY && b16[q] != -INFINITY) ? -3.402823466e38f : bl;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                biasM[kt][k] = f32x2{b16[2 * k], b16[2 * k + 1]};
                gate[kt][k] = f32x2{g16[2 * k], g16[2 * k + 1]};
                dE[kt][k] = dG[kt][k] = f32x2{0.f, 0.f};     // dG accumulates sum_j dA*P; the gate factor is applied once, after the walk
            }
        }
        __syncthreads();

        uint4 pq[SlabIO<G, 32>::kIters], po[SlabIO<G, 32>::kIters];
        uint4 pk[SlabIO<G, KR>::kIters], pv[SlabIO<G, KR>::kIters];
        slab_issue<G, 32>(pq, bQ, 0, i0, N, tid);
        slab_issue<G, 32>(po, dO, 0, i0, N, tid);
        slab_issue<G, KR>(pk, bK, 0, 0, N, tid);
        slab_issue<G, KR>(pv, bV, 0, 0, N, tid);
        slab_commit<G, 32>(pq, smem, tid);
        slab_commit<G, 32>(po, smem + G::kSlabBytes, tid);
        slab_commit<G, KR>(pk, smem + 2 * G::kSlabBytes, tid);
        slab_commit<G, KR>(pv, smem + (2 + NT) * G::kSlabBytes, tid);
        if (N > 1) {
            slab_issue<G, 32>(pq, bQ, 1, i0, N, tid);
            slab_issue<G, 32>(po, dO, 1, i0, N, tid);
            slab_issue<G, KR>(pk, bK, 1, 0, N, tid);
            slab_issue<G, KR>(pv, bV, 1, 0, N, tid);
        }
        __syncthreads();

        for (int j = 0; j < N; ++j) {
            if constexpr (DROP && NT > 1) {
                // The two-tile dropout variants are the most register-starved instantiations, and hipcc
                // (ROCm 7.2) miscompiled the spill of these loop-invariant fragments there (3 dwords to
                // scratch, the 4th parked in an AGPR and never restored: tools/isa_defuse_lint.py).
                // Rebuilding them per j from an opaque copy of r keeps them out of the spill set.
                int rr = r;
                asm volatile("" : "+v"(rr));
                make_ident_d<T, G::kDC>(ident_d, rr, hi);
                make_ident_k<T>(ident_k, rr, hi);
            }
            char* sQ = smem + (j & 1) * kSet;
            char* sO = sQ + G::kSlabBytes;
            char* sK = sQ + 2 * G::kSlabBytes;
            char* sV = sK + NT * G::kSlabBytes;
            TGT_PROBE_T(0);                // (loop overhead + whatever the previous iteration left)
            if (j + 1 < N) {
                char* nQ = smem + ((j + 1) & 1) * kSet;
                slab_commit<G, 32>(pq, nQ, tid);
                slab_commit<G, 32>(po, nQ + G::kSlabBytes, tid);
                slab_commit<G, KR>(pk, nQ + 2 * G::kSlabBytes, tid);
                slab_commit<G, KR>(pv, nQ + (2 + NT) * G::kSlabBytes, tid);
            }
            TGT_PROBE_T(1);                // commit: the wait for the prefetched slabs + 4 LDS writes
            if (j + 2 < N && !(ablate & 1)) {
                slab_issue<G, 32>(pq, bQ, j + 2, i0, N, tid);
                slab_issue<G, 32>(po, dO, j + 2, i0, N, tid);
                slab_issue<G, KR>(pk, bK, j + 2, 0, N, tid);
                slab_issue<G, KR>(pv, bV, j + 2, 0, N, tid);
            }
            // partial dK/dV the previous query-tile pass stored for THIS j (added at the store below)
            uint4 curk[NT > 1 ? SlabIO<G, KR>::kIters : 1], curv[NT > 1 ? SlabIO<G, KR>::kIters : 1];
            if constexpr (NT > 1) {
                if (it > 0) {
                    slab_issue<G, KR>(curk, dK, j, 0, N, tid);
                    slab_issue<G, KR>(curv, dV, j, 0, N, tid);
                }
            }
            TGT_PROBE_T(2);                // prefetch issue
            if (!(ablate & 4)) {
            F fq[G::kDC], fo[G::kDC];
            read_frags<T, D, HG>(fq, sQ, wave, r, hi);
            read_frags<T, D, HG>(fo, sO, wave, r, hi);

            f32x2 s[NT][8], da[NT][8];
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                F fk[G::kDC], fv[G::kDC];
                read_frags<T, D, HG>(fk, sK, wave, 32 * kt + r, hi);
                read_frags<T, D, HG>(fv, sV, wave, 32 * kt + r, hi);
                f32x16 z0 = {0}, z1 = {0};
#pragma unroll
                for (int dc = 0; dc < G::kDC; ++dc) z0 = mma32(fk[dc], fq[dc], z0);         // S^T[k][i]
#pragma unroll
                for (int dc = 0; dc < G::kDC; ++dc) z1 = mma32(fv[dc], fo[dc], z1);         // dA^T[k][i]
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    s[kt][k] = f32x2{z0[2 * k], z0[2 * k + 1]};
                    da[kt][k] = f32x2{z1[2 * k], z1[2 * k + 1]};
                }
            }

            // softmax statistics are recomputed (in-lane values + the partner lane); saving a
            // log-sum-exp instead would lose log(sum) next to a finfo.min-sized row maximum.
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    s[kt][k] = s[kt][k] * scale2 + biasM[kt][k];
                    mx = fmaxf(mx, fmaxf(s[kt][k].x, s[kt][k].y));
                }
            mx = fmaxf(mx, xhalf(mx));
            if (mx == -INFINITY) mx = 0.f;           // padding column: every weight is exactly 0
            f32x2 sum2 = {0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const f32x2 t = s[kt][k] - mx;
                    s[kt][k] = f32x2{fast_exp2(t.x), fast_exp2(t.y)};
                    sum2 += s[kt][k];
                }
            float sum = sum2.x + sum2.y;
            sum += xhalf(sum);
            const float inv = sum > 0.f ? fast_rcp(sum) : 0.f;
            // attention dropout: the kept weights were scaled by 1/(1-p), so dA (and A below) carry the mask
            uint32_t keep[DROP ? NT : 1];
            if constexpr (DROP) {
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) {
                    keep[kt] = tri_drop_bits(drop, drop_unit0 + j, i0 + r, kt, hi);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        da[kt][k].x = (keep[kt] >> (2 * k)) & 1u ? da[kt][k].x * drop.scale : 0.f;
                        da[kt][k].y = (keep[kt] >> (2 * k + 1)) & 1u ? da[kt][k].y * drop.scale : 0.f;
                    }
                }
            }
            // s -> P;  da -> dP = dA * g;  delta_i = sum_k P dP
            f32x2 delta2 = {0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const f32x2 p = s[kt][k] * inv;
                    const f32x2 dp = da[kt][k] * gate[kt][k];
                    delta2 += p * dp;
                    if (gated) dG[kt][k] += da[kt][k] * p;
                    s[kt][k] = p;
                    da[kt][k] = dp;
                }
            float delta = delta2.x + delta2.y;
            delta += xhalf(delta);

            f32x16 qT = {0}, oT = {0};
#pragma unroll
            for (int dc = 0; dc < G::kDC; ++dc) qT = mma32(fq[dc], ident_d[dc], qT);   // Q^T
#pragma unroll
            for (int dc = 0; dc < G::kDC; ++dc) oT = mma32(fo[dc], ident_d[dc], oT);   // dO^T
            F qTf[2] = {pack_chunk<T>(qT, 0), pack_chunk<T>(qT, 1)};
            F oTf[2] = {pack_chunk<T>(oT, 0), pack_chunk<T>(oT, 1)};

            // one key tile at a time from here on (short live ranges): dS, A -> dQ partial, dK, dV
            f32x16 dq = {0};
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                F dsf[2], af[2];
                {
                    f32x16 att, dsv;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const f32x2 ds = s[kt][k] * (da[kt][k] - delta);
                        if (biased) dE[kt][k] += ds;
                        f32x2 at = s[kt][k] * gate[kt][k];
                        if constexpr (DROP) {
                            at.x = (keep[kt] >> (2 * k)) & 1u ? at.x * drop.scale : 0.f;
                            at.y = (keep[kt] >> (2 * k + 1)) & 1u ? at.y * drop.scale : 0.f;
                        }
                        const f32x2 dss = ds * a.scale;
                        att[2 * k] = at.x; att[2 * k + 1] = at.y;
                        dsv[2 * k] = dss.x; dsv[2 * k + 1] = dss.y;
                    }
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc) {
                        dsf[cc] = pack_chunk<T>(dsv, cc);
                        af[cc] = pack_chunk<T>(att, cc);
                    }
                }
                {   // dQ^T[d][i] += sum_k K^T[d][k] dS^T[k][i]   (K^T through the identity trick)
                    F fk[G::kDC];
                    read_frags<T, D, HG>(fk, sK, wave, 32 * kt + r, hi);
                    f32x16 kT = {0};
#pragma unroll
                    for (int dc = 0; dc < G::kDC; ++dc) kT = mma32(fk[dc], ident_d[dc], kT);
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc) dq = mma32(pack_chunk<T>(kT, cc), dsf[cc], dq);
                }
                // re-layout dS, A to lane = k:  X[i][k] = sum_kk X^T-frag[i][kk] I[kk][k]
                f32x16 ds2 = {0}, a2 = {0}, dk = {0}, dv = {0};
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    ds2 = mma32(dsf[cc], ident_k[cc], ds2);
                    a2 = mma32(af[cc], ident_k[cc], a2);
                }
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    dk = mma32(qTf[cc], pack_chunk<T>(ds2, cc), dk);   // dK^T[d][k] = sum_i Q^T[d][i] dS[i][k]
                    dv = mma32(oTf[cc], pack_chunk<T>(a2, cc), dv);    // dV^T[d][k] = sum_i dO^T[d][i] A[i][k]
                }
                // NOTE: this wave still has to read the K fragments of the NEXT key tile from sK,
                // and write_rows only touches the rows of THIS tile, so in-place is safe
                write_rows<T, D, HG>(sK, dk, wave, 32 * kt + r, hi);
                write_rows<T, D, HG>(sV, dv, wave, 32 * kt + r, hi);
            }
            write_rows<T, D, HG>(sQ, dq, wave, r, hi);
            }
            TGT_PROBE_T(3);                // tile math (LDS fragment reads .. result rows written to LDS)
            __syncthreads();
            TGT_PROBE_T(4);                // barrier
            bool plain = true;
            if (ablate & 2) {
            } else
            if constexpr (CS) {
                slab_store_sum<G, 32, T, false>(sQ, pq, gQ, j, i0, N, tid, cs);
                if constexpr (NT > 1) {
                    if (it > 0) {
                        plain = false;
                        slab_store_sum<G, KR, T, true>(sK, curk, dK, j, 0, N, tid, cs + kPlane);
                        slab_store_sum<G, KR, T, true>(sV, curv, dV, j, 0, N, tid, cs + 2 * kPlane);
                    }
                }
                if (plain) {
                    slab_store_sum<G, KR, T, false>(sK, pk, dK, j, 0, N, tid, cs + kPlane);
                    slab_store_sum<G, KR, T, false>(sV, pv, dV, j, 0, N, tid, cs + 2 * kPlane);
                }
            } else {
                slab_store<G, 32>(sQ, gQ, j, i0, N, tid);
                if constexpr (NT > 1) {
                    if (it > 0) {
                        plain = false;
                        slab_store_add<G, KR, T>(sK, curk, dK, j, 0, N, tid);
                        slab_store_add<G, KR, T>(sV, curv, dV, j, 0, N, tid);
                    }
                }
                if (plain) {
                    slab_store<G, KR>(sK, dK, j, 0, N, tid);
                    slab_store<G, KR>(sV, dV, j, 0, N, tid);
                }
            }
            TGT_PROBE_T(5);                // result stores (+ column sums)
        }
        __syncthreads();      // the next query-tile pass re-fills both sets
        // third-arm gradients of this query tile (summed over j in registers) leave through LDS
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            float e16[16], g16[16];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                f32x2 gg = dG[kt][k];
                if (gated) gg *= gate[kt][k] * (1.f - gate[kt][k]);      // d sigmoid, once per tile
                e16[2 * k] = dE[kt][k].x; e16[2 * k + 1] = dE[kt][k].y;
                g16[2 * k] = gg.x; g16[2 * k + 1] = gg.y;
            }
            arm_stage_put_grad<T, HG, NT>(smem, c.dir, wave, r, hi, kt, e16, g16);
        }
        __syncthreads();
        {
            const float part = arm_stage_store_grad<T, HG, NT>(dta, a.d_eg[c.dir], c.b, c.dir, c.g, N, i0, smem, tid);
            if constexpr (CS) cs[3 * kPlane + tid] += part;
        }
        __syncthreads();
    }
    if constexpr (CS) {
        // (the last barrier above also ordered every wave's accumulator updates)
        float* row = a.d_qkv_colsum[c.dir] + (int64_t)c.b * ldq + c.g * HG * D;
        slab_colsum_finish<G, T>(cs, row + a.q_off[c.dir], tid);
        slab_colsum_finish<G, T>(cs + kPlane, row + a.k_off[c.dir], tid);
        slab_colsum_finish<G, T>(cs + 2 * kPlane, row + a.v_off[c.dir], tid);
        if (biased || gated) {
            constexpr int kVals = ArmStage<T, HG, NT>::kVals;      // E of the HG heads, then G
            if (tid < kVals) {      // (the barrier that ended the last pass ordered the partials)
                float v = 0.f;
                for (int t = tid; t < HG * 64; t += kVals) v += cs[3 * kPlane + t];
                float* erow = a.d_eg_colsum[c.dir] + (int64_t)c.b * lde;
                if (tid < HG) { if (biased) erow[a.e_off[c.dir] + c.g * HG + tid] = v; }
                else if (gated) erow[a.g_off[c.dir] + c.g * HG + tid - HG] = v;
            }
        }
    }
#ifdef TGT_PROBES
    if (tid == 0 && blockIdx.x < 4096) {
        pt[7] = probe_now() - t_begin;
#pragma unroll
        for (int i = 0; i < 8; ++i) g_tri_probe[blockIdx.x * 8 + i] = pt[i];
    }
#endif
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
#ifndef TGT_NT2_OCC
#define TGT_NT2_OCC 1      // measured: capping the two-tile backward at 256 registers (2 waves per SIMD) spills -- 7.0 ms against 1.6 ms
#endif
template <typename T, int D, int HG, int NT>
static int launch_tri_nt(const tgt_triplet_attention_args& a_in, bool bwd, hipStream_t st) {
    using G = TriGeo<T, D, HG>;
#ifdef TGT_PROBES
    tgt_triplet_attention_args a = a_in;
    a._pad0 = getenv("TGT_TRI_BWD_ABLATE") ? atoi(getenv("TGT_TRI_BWD_ABLATE")) : 0;
#else
    const tgt_triplet_attention_args& a = a_in;
#endif
    const int grid = a.B * 2 * (a.H / HG);
    constexpr int kArm = ArmStage<T, HG, NT>::kBytes;
    constexpr int kFwdLds = 2 * (1 + 2 * NT) * G::kSlabBytes > kArm ? 2 * (1 + 2 * NT) * G::kSlabBytes : kArm;
    // (+ the column-sum accumulators, 3 planes of 16/sizeof(T) fp32 per thread, when requested)
    constexpr int kBwdLds = 2 * (2 + 2 * NT) * G::kSlabBytes > kArm ? 2 * (2 + 2 * NT) * G::kSlabBytes : kArm;
    const bool drop = a.dropout_p > 0.f;
    if (!bwd) {
        // (a prefetch depth of 2 was measured neutral in round 1 and is gone)
        if (drop)
            hipLaunchKernelGGL((tri_att_fwd_kernel<T, D, HG, NT, 1, true>), dim3(grid), dim3(G::kThreads), kFwdLds, st, a);
        else
            hipLaunchKernelGGL((tri_att_fwd_kernel<T, D, HG, NT, 1, false>), dim3(grid), dim3(G::kThreads),
                               kFwdLds, st, a);
    } else {
        // one node tile: registers capped for 2 waves per SIMD (kOcc)
        const bool cs = a.d_qkv_colsum[0] != nullptr;
        constexpr int kCs = (3 * slab_colsum_plane_floats<G, T>() + G::kThreads) * 4;
        constexpr int kOcc = NT == 1 ? 2 : 1;
        constexpr int kBG = TGT_TRI_BIASED | TGT_TRI_GATED;
        if (drop) {                 // attention dropout (p = 0 in every shipped config): one generic variant each
            if (cs)
                hipLaunchKernelGGL((tri_att_bwd_kernel<T, D, HG, NT, kOcc, true, -1, true>), dim3(grid), dim3(G::kThreads),
                                   kBwdLds + kCs, st, a);
            else
                hipLaunchKernelGGL((tri_att_bwd_kernel<T, D, HG, NT, kOcc, false, -1, true>), dim3(grid), dim3(G::kThreads),
                                   kBwdLds, st, a);
        } else if (NT == 1) {
            if (cs && HG == 8 && (a.flags & kBG) == kBG)         // the training hot path: flags compiled in
                hipLaunchKernelGGL((tri_att_bwd_kernel<T, D, HG, NT, kOcc, true, (HG == 8 ? kBG : -1), false>), dim3(grid),
                                   dim3(G::kThreads), kBwdLds + kCs, st, a);
            else if (cs)
                hipLaunchKernelGGL((tri_att_bwd_kernel<T, D, HG, NT, kOcc, true, -1, false>), dim3(grid),
                                   dim3(G::kThreads), kBwdLds + kCs, st, a);
            else
                hipLaunchKernelGGL((tri_att_bwd_kernel<T, D, HG, NT, kOcc, false, -1, false>), dim3(grid),
                                   dim3(G::kThreads), kBwdLds, st, a);
        } else {
            // two node tiles (N in 33..64): TGT_NT2_OCC waves per SIMD (register cap 512 / TGT_NT2_OCC)
            constexpr int kOcc2 = NT == 2 ? TGT_NT2_OCC : 1;
            if (cs)
                hipLaunchKernelGGL((tri_att_bwd_kernel<T, D, HG, NT, kOcc2, true, -1, false>), dim3(grid), dim3(G::kThreads),
                                   kBwdLds + kCs, st, a);
            else
                hipLaunchKernelGGL((tri_att_bwd_kernel<T, D, HG, NT, kOcc2, false, -1, false>), dim3(grid), dim3(G::kThreads),
                                   kBwdLds, st, a);
        }
    }
    return check_launch(bwd ? "tri_att_bwd_kernel" : "tri_att_fwd_kernel");
}
template <typename T, int D, int HG>
static int launch_tri(const tgt_triplet_attention_args& a, bool bwd, hipStream_t st) {
    if (a.N <= 32) return launch_tri_nt<T, D, HG, 1>(a, bwd, st);
    return launch_tri_nt<T, D, HG, 2>(a, bwd, st);
}

template <typename T, int D>
static int dispatch_hg(const tgt_triplet_attention_args& a, bool bwd, hipStream_t st) {
    if constexpr (D == 16 && sizeof(T) == 2) {         // 8 heads per workgroup: 256-byte row pieces
        if (a.H % 8 == 0 && a.N <= 32) return launch_tri_nt<T, D, 8, 1>(a, bwd, st);
    }
    if (a.H % 4 == 0) return launch_tri<T, D, 4>(a, bwd, st);
    if constexpr (D * sizeof(T) >= 16) return launch_tri<T, D, 1>(a, bwd, st);
    return set_error(TGT_ERR_UNSUPPORTED, "triplet attention: H=%d not a multiple of 4 with D=%d", a.H, D);
}

template <typename T>
static int dispatch_d(const tgt_triplet_attention_args& a, bool bwd, hipStream_t st) {
    switch (a.D) {
        case 8: return dispatch_hg<T, 8>(a, bwd, st);
        case 16: return dispatch_hg<T, 16>(a, bwd, st);
        case 32: return dispatch_hg<T, 32>(a, bwd, st);
        default: return set_error(TGT_ERR_UNSUPPORTED, "triplet attention: D=%d not in {8,16,32}", a.D);
    }
}

// The three dtypes are separate translation units in the build (TGT_TRI_INST: bit 0 fp32, bit 1
// bf16, bit 2 fp16, bit 3 the argument checks + dispatch) so that they compile in parallel; one
// TU with everything when the macro is not given.
#ifndef TGT_TRI_INST
#define TGT_TRI_INST 15
#endif
int tri_att_run_f32(const tgt_triplet_attention_args& a, bool bwd, hipStream_t st);
int tri_att_run_bf16(const tgt_triplet_attention_args& a, bool bwd, hipStream_t st);
int tri_att_run_f16(const tgt_triplet_attention_args& a, bool bwd, hipStream_t st);
#if TGT_TRI_INST & 1
int tri_att_run_f32(const tgt_triplet_attention_args& a, bool bwd, hipStream_t st) { return dispatch_d<float>(a, bwd, st); }
#endif
#if TGT_TRI_INST & 2
int tri_att_run_bf16(const tgt_triplet_attention_args& a, bool bwd, hipStream_t st) { return dispatch_d<bf16_t>(a, bwd, st); }
#ifdef TGT_PROBES
}  // namespace tgt
// probe build only: the segment cycle totals the last bf16 backward launch left (8 per workgroup, see TGT_PROBE_T)
extern "C" int tgt_probe_read(void* dst, int n_u64) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(tgt::g_tri_probe), (size_t)n_u64 * 8);
}
namespace tgt {
#endif
#endif
#if TGT_TRI_INST & 4
int tri_att_run_f16(const tgt_triplet_attention_args& a, bool bwd, hipStream_t st) { return dispatch_d<f16_t>(a, bwd, st); }
#endif

#if TGT_TRI_INST & 8
bool tri_att16_fwd_eligible(const tgt_triplet_attention_args& a);
int tri_att16_fwd_run(const tgt_triplet_attention_args& a, hipStream_t st);
bool tri_att16_bwd_eligible(const tgt_triplet_attention_args& a);
int tri_att16_bwd_run(const tgt_triplet_attention_args& a, hipStream_t st);
bool tri_att_bwd2_eligible(const tgt_triplet_attention_args& a);       // triplet_attention_bwd2.hip: 16-bit, D = 16, N <= 32, H % 8 == 0
int tri_att_bwd2_run(const tgt_triplet_attention_args& a, hipStream_t st);

int triplet_attention_run(const tgt_triplet_attention_args* a, bool bwd, hipStream_t st) {
    if (!a) return set_error(TGT_ERR_INVALID, "triplet attention: null args");

    if (a->B < 0 || a->N < 0 || a->H <= 0) return set_error(TGT_ERR_INVALID, "triplet attention: bad sizes B=%d N=%d H=%d", a->B, a->N, a->H);
    if (a->B == 0 || a->N == 0) return TGT_OK;                 // empty batch: nothing to do
    if (a->N > 64) return set_error(TGT_ERR_UNSUPPORTED, "triplet attention: N=%d > 64 not supported", a->N);
    if (!(a->dropout_p >= 0.f && a->dropout_p < 1.f)) return set_error(TGT_ERR_INVALID, "triplet attention: dropout_p=%f outside [0,1)", a->dropout_p);
    const int64_t esz = a->dtype == TGT_F32 ? 4 : 2;
    for (int dir = 0; dir < 2; ++dir) {
        if (!a->qkv[dir] || !a->out || !a->mask) return set_error(TGT_ERR_INVALID, "triplet attention: null tensor");
        if ((a->ld_qkv[dir] * esz) % 16 || (a->q_off[dir] * esz) % 16 || (a->k_off[dir] * esz) % 16 ||
            (a->v_off[dir] * esz) % 16 || (a->ld_out * esz) % 16 || (a->o_off[dir] * esz) % 16 ||
            ((uintptr_t)a->qkv[dir] % 16) || ((uintptr_t)a->out % 16))
            return set_error(TGT_ERR_INVALID, "triplet attention: rows/offsets must be 16-byte aligned");
        if ((a->flags & (TGT_TRI_BIASED | TGT_TRI_GATED)) && !a->eg[dir]) return set_error(TGT_ERR_INVALID, "triplet attention: eg missing");
        if (bwd) {
            if ((a->d_qkv_colsum[dir] != nullptr) != (a->d_qkv_colsum[0] != nullptr) ||
                ((a->flags & (TGT_TRI_BIASED | TGT_TRI_GATED)) && (a->d_eg_colsum[dir] != nullptr) != (a->d_qkv_colsum[0] != nullptr)))
                return set_error(TGT_ERR_INVALID, "triplet attention bwd: d_qkv_colsum / d_eg_colsum must be all set or all NULL");
            if (a->ld_dqkv[dir] < 0 || a->ld_deg[dir] < 0 || (a->ld_dqkv[dir] * esz) % 16 || (a->ld_deg[dir] * esz) % 16)
                return set_error(TGT_ERR_INVALID, "triplet attention bwd: ld_dqkv / ld_deg must be 0 or 16-byte-aligned row lengths");
            if (!a->d_out || !a->d_qkv[dir] || ((uintptr_t)a->d_qkv[dir] % 16) || ((uintptr_t)a->d_out % 16))
                return set_error(TGT_ERR_INVALID, "triplet attention bwd: null/misaligned gradient tensor");
            if ((a->flags & (TGT_TRI_BIASED | TGT_TRI_GATED)) && !a->d_eg[dir]) return set_error(TGT_ERR_INVALID, "triplet attention bwd: d_eg missing");
        }
    }
    if (!bwd && tri_att16_fwd_eligible(*a)) return tri_att16_fwd_run(*a, st);      // 33 <= N <= 64: 16-wide tiles (triplet_attention16.hip)
    if (bwd && tri_att16_bwd_eligible(*a)) return tri_att16_bwd_run(*a, st);
    if (bwd && tri_att_bwd2_eligible(*a)) return tri_att_bwd2_run(*a, st);
    switch (a->dtype) {
        case TGT_F32: return tri_att_run_f32(*a, bwd, st);
        case TGT_BF16: return tri_att_run_bf16(*a, bwd, st);
        case TGT_F16: return tri_att_run_f16(*a, bwd, st);
        default: return set_error(TGT_ERR_INVALID, "triplet attention: bad dtype %d", a->dtype);
    }
}
#endif

}  // namespace tgt
