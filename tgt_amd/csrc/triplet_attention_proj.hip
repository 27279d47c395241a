// Triplet attention forward with the Q/K/V projection fused in -- gfx950.
//
// Reference lib/tgt/layers/triplet.py:210-246: `lin_QKV_in/out(e_ln)` followed by the two
// einsum -> softmax -> gate -> einsum chains.  The unfused path writes the 1536 projected
// channels of every edge (0.8 GB at B=256) with a library GEMM and reads them straight back in
// the attention kernel.  Here the workgroup that walks node j projects the 64 edge rows it
// needs itself, on the matrix cores, and feeds the attention core from registers; Q/K/V still
// go to HBM once (the backward kernel reads them), but nothing is read back in the forward,
// and the tall-skinny GEMM (K = 256: four k-iterations per 256x256 tile, 555 TFLOP/s in the
// library) disappears.
//
//   workgroup = (graph b, direction, 8 heads), wave = head h (as triplet_attention.hip)
//   per j:  X_q = e_ln rows (i,j), i < 32        X_kv = rows (j,k) inward / (k,j) outward
//           Q^T[d][i]   = W_Q(h)  . X_q^T          \  A = weight rows (resident in VGPRs for
//           [K|V]^T[.][k] = [W_K(h);W_V(h)] . X_kv^T /  the whole walk), B = e_ln rows from LDS
//   The projection result is produced TRANSPOSED (channel in registers, edge row in the
//   lane), which is at once (a) the operand-fragment layout the attention MFMAs want --
//   pack the accumulator to bf16, no LDS round trip -- and (b) the layout write_rows() turns
//   into full rows for the coalesced Q/K/V store.
//   Bias: the accumulators start from the bias instead of zero.
//   e_ln tiles (64 rows x 512 B) are double-buffered in LDS, XOR-swizzled by row so that the
//   32 lanes of a fragment read hit 32 different 16-byte slots; HBM -> VGPR -> LDS prefetch of
//   tile j+1 runs under the matrix work of tile j.  One barrier per j (same hazard argument as
//   the unfused kernel: two slab sets, two tile buffers).
// Supported: N <= 32, D = 16, H % 8 == 0, 16-bit dtypes, C in {64, 128, 256}.
//
// STATUS (round 1, measured on MI355X, B=256 N=32 C=256): correct (tests/test_hip_ops.py), but
// 0.77 ms against 0.39 ms (library GEMM) + 0.24 ms (attention kernel) for the pair it replaces,
// so the host keeps it OFF by default (TGT_TRI_PROJ=1 turns it on).  Where the time goes, from
// switching parts off: skeleton (core + LDS + barriers) 0.24, + projection MFMAs 0.35, + Q/K/V/O
// stores 0.17, + tile loads 0.10 -- the phases add up instead of overlapping: the 128 VGPRs of
// resident weights leave 2 waves/SIMD in ONE workgroup per CU, all meeting at the same
// barrier.  4-head workgroups (two per CU, single-buffered tile) measured slower still (1.11 vs
// 0.92 ms on one box).  Next: wave-specialised producer/consumer roles.  See DESIGN.md section 4.1a.
#include <cstdlib>
#include "triplet_common.hpp"

namespace tgt {

template <typename T, int KS>
struct ProjGeo {
    static constexpr int C = 16 * KS;
    static constexpr int kRowBytes = C * (int)sizeof(T);
    static constexpr int kSlots = kRowBytes / 16;
    static constexpr int kTileBytes = 64 * kRowBytes;
    // XOR swizzle of the 16-byte slot by the row: the 32 rows of one fragment read (same logical
    // slot) land in 32 different slots
    __device__ static __forceinline__ int off(int row, int slot) {
        return row * kRowBytes + ((slot ^ (row & (kSlots - 1))) << 4);
    }
};

// number of 16-byte chunks per thread for one e_ln tile: rows * slots / 512 threads
template <typename T, int KS, int DIR>
struct XStage {
    using P = ProjGeo<T, KS>;
    static constexpr int kRows = DIR == 0 ? 64 : 32;
    static constexpr int kChunks = kRows * P::kSlots;
    static constexpr int kIters = (kChunks + 511) / 512;
};

template <typename T, int KS, int DIR, int IT = XStage<T, KS, DIR>::kIters>
__device__ __forceinline__ void xtile_issue(uint4 (&pre)[IT], const char* xg, int N, int j, int tid) {
    using P = ProjGeo<T, KS>;
    using S = XStage<T, KS, DIR>;
#pragma unroll
    for (int it = 0; it < S::kIters; ++it) {
        const int c = it * 512 + tid;
        const int row = c / P::kSlots, slot = c % P::kSlots;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (c < S::kChunks) {
            // tile rows 0..31: edge (row, j);  rows 32..63 (inward only): edge (j, row - 32)
            const int p = row < 32 ? row : j, q = row < 32 ? j : row - 32;
            if (p < N && q < N)
                v = *reinterpret_cast<const uint4*>(xg + ((int64_t)p * N + q) * P::kRowBytes + slot * 16);
        }
        pre[it] = v;
    }
}
template <typename T, int KS, int DIR, int IT = XStage<T, KS, DIR>::kIters>
__device__ __forceinline__ void xtile_commit(const uint4 (&pre)[IT], char* tile, int tid) {
    using P = ProjGeo<T, KS>;
    using S = XStage<T, KS, DIR>;
#pragma unroll
    for (int it = 0; it < S::kIters; ++it) {
        const int c = it * 512 + tid;
        const int row = c / P::kSlots, slot = c % P::kSlots;
        if (c < S::kChunks) *reinterpret_cast<uint4*>(tile + P::off(row, slot)) = pre[it];
    }
}

template <typename T, int KS, int DIR>
__device__ __forceinline__ void proj_walk(const tgt_triplet_attention_args& a, const TriCtx& c, const char* xg,
                                          const frag_t<T> (&wq)[KS], const frag_t<T> (&wkv)[KS], const float* bias_w,
                                          char* smem, int tid) {
    constexpr int D = 16, HG = 8;
    using G = TriGeo<T, D, HG>;
    using F = frag_t<T>;
    using P = ProjGeo<T, KS>;
    constexpr int kSet = 4 * G::kSlabBytes;            // {Q | K | V | O} rows of one j
    char* xbuf = smem;                                  // 2 tiles
    char* slabs = smem + 2 * P::kTileBytes;             // 2 sets
    const int lane = tid & 63, wave = tid >> 6, r = lane & 31, hi = lane >> 5;
    const int N = c.N;
    const ThirdArm ta = tri_third_arm(a, DIR);
    F ident_k[2];
    make_ident_k<T>(ident_k, r, hi);

    // buffer-addressed result slabs (triplet_common.hpp): Q rows (i,j); K / V partner rows (j,k) inward, (k,j) outward
    const int64_t sz = sizeof(T), Nl = N;
    const uint32_t hch = (uint32_t)(c.g * HG * D * sz), lds_ = (uint32_t)(a.ld_qkv[DIR] * sz), ldo_ = (uint32_t)(a.ld_out * sz);
    const __amdgpu_buffer_rsrc_t r_dst = graph_rsrc(a.qkv[DIR], Nl * Nl * a.ld_qkv[DIR] * sz, c.b);
    const SlabBuf bQ = {r_dst, (uint32_t)(a.q_off[DIR] * sz) + hch, (uint32_t)N * lds_, lds_};
    const SlabBuf bK = {r_dst, (uint32_t)(a.k_off[DIR] * sz) + hch, DIR == 0 ? lds_ : (uint32_t)N * lds_, DIR == 0 ? (uint32_t)N * lds_ : lds_};
    const SlabBuf bV = {r_dst, (uint32_t)(a.v_off[DIR] * sz) + hch, bK.row_stride, bK.j_stride};
    const SlabBuf bO = {graph_rsrc(a.out, Nl * Nl * a.ld_out * sz, c.b), (uint32_t)(a.o_off[DIR] * sz) + hch, (uint32_t)N * ldo_, ldo_};

    float biasM[16], gate[16];
    arm_stage_load<T, HG, 1>(ta, c.b, DIR, c.g, N, 0, smem, tid);
    __syncthreads();
    arm_stage_read<T, HG, 1, false>(ta, smem, DIR, wave, N, r, hi, 0, 0, biasM, gate);
    __syncthreads();

    uint4 px[XStage<T, KS, DIR>::kIters];
    xtile_issue<T, KS, DIR>(px, xg, N, 0, tid);
    xtile_commit<T, KS, DIR>(px, xbuf, tid);
    if (N > 1) xtile_issue<T, KS, DIR>(px, xg, N, 1, tid);
    __syncthreads();

    for (int j = 0; j < N; ++j) {
        const char* xt = xbuf + (j & 1) * P::kTileBytes;
        char* sQ = slabs + (j & 1) * kSet;
        char* sK = sQ + G::kSlabBytes;
        char* sV = sK + G::kSlabBytes;
        char* sO = sV + G::kSlabBytes;
        if (j + 1 < N) xtile_commit<T, KS, DIR>(px, xbuf + ((j + 1) & 1) * P::kTileBytes, tid);
        if (j + 2 < N) xtile_issue<T, KS, DIR>(px, xg, N, j + 2, tid);

        // ---- projection: accumulators start from the bias (rows = channel, lane = edge row)
        f32x16 qa, kva;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            qa[q] = q < 8 ? bias_w[acc_row(q, hi)] : 0.f;          // Q_d, d < 16 (rows 16..31 unused)
            kva[q] = bias_w[16 + acc_row(q, hi)];                   // K_d | V_d
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const F xq = load_frag<T>(reinterpret_cast<const T*>(xt + P::off(r, 2 * s + hi)));
            F xk = xq;
            if constexpr (DIR == 0) xk = load_frag<T>(reinterpret_cast<const T*>(xt + P::off(32 + r, 2 * s + hi)));
            qa = mma32(wq[s], xq, qa);
            kva = mma32(wkv[s], xk, kva);
        }
        // rows for the backward kernel, and the attention operands (same rounding)
        f32x16 va;
#pragma unroll
        for (int q = 0; q < 8; ++q) va[q] = kva[8 + q];
#pragma unroll
        for (int q = 8; q < 16; ++q) va[q] = 0.f;
        write_rows<T, D, HG>(sQ, qa, wave, r, hi);
        write_rows<T, D, HG>(sK, kva, wave, r, hi);
        write_rows<T, D, HG>(sV, va, wave, r, hi);
        const F fq = pack_chunk<T>(qa, 0), fk = pack_chunk<T>(kva, 0), fv = pack_chunk<T>(kva, 1);

        // ---- attention core (as tri_att_fwd_kernel; fragment k-order = accumulator order)
        f32x16 z0 = {0}, z1 = {0};
        f32x16 st = mma32(fk, fq, z0);                 // S^T[k][i]
        f32x16 vt = mma32(fv, ident_k[0], z1);         // V[k][d] -> lane d
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
        write_rows<T, D, HG>(sO, o, wave, r, hi);
        __syncthreads();
        slab_store<G, 32>(sQ, bQ, j, 0, N, tid);
        slab_store<G, 32>(sK, bK, j, 0, N, tid);
        slab_store<G, 32>(sV, bV, j, 0, N, tid);
        slab_store<G, 32>(sO, bO, j, 0, N, tid);
    }
}

template <typename T, int KS>
__global__ void __launch_bounds__(512, 2) tri_att_proj_fwd_kernel(const tgt_triplet_attention_args a, const T* x,
                                                                  const T* w, const T* bias) {
    constexpr int D = 16, HG = 8, C = 16 * KS;
    using F = frag_t<T>;
    using P = ProjGeo<T, KS>;
    using G = TriGeo<T, D, HG>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hi = lane >> 5;
    const TriCtx c = tri_ctx<T, D, HG>(a, wave);
    const int dir = c.dir;

    // bias of this head: [Q_d | K_d | V_d], 16 each, as floats behind the tiles and slab sets
    float* bias_w = reinterpret_cast<float*>(smem + 2 * P::kTileBytes + 8 * G::kSlabBytes) + wave * 48;
    if (lane < 48) {
        const int part = lane >> 4, d = lane & 15;
        const int off = part == 0 ? a.q_off[dir] : (part == 1 ? a.k_off[dir] : a.v_off[dir]);
        bias_w[lane] = to_f32(bias[off + c.h * D + d]);
    }
    // weight rows of this head as A operands, resident for the whole walk:
    //   Q tile rows 0..15 = W_Q(h) (rows 16..31 zero); K|V tile rows 0..15 = W_K(h), 16..31 = W_V(h)
    F wq[KS], wkv[KS];
    {
        const int d = r & 15;
        const T* rq = w + (int64_t)(a.q_off[dir] + c.h * D + d) * C;
        const T* rkv = w + (int64_t)((r < 16 ? a.k_off[dir] : a.v_off[dir]) + c.h * D + d) * C;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            wq[s] = r < 16 ? load_frag<T>(rq + 16 * s + 8 * hi) : zero_frag<T>();
            wkv[s] = load_frag<T>(rkv + 16 * s + 8 * hi);
        }
    }
    const char* xg = reinterpret_cast<const char*>(x) + (int64_t)c.b * c.N * c.N * P::kRowBytes;
    if (dir == 0) proj_walk<T, KS, 0>(a, c, xg, wq, wkv, bias_w, smem, tid);
    else proj_walk<T, KS, 1>(a, c, xg, wq, wkv, bias_w, smem, tid);
}

template <typename T, int KS>
static int launch_proj(const tgt_triplet_attention_args& a, const void* x, const void* w, const void* bias, hipStream_t st) {
    using G = TriGeo<T, 16, 8>;
    using P = ProjGeo<T, KS>;
    const int grid = a.B * 2 * (a.H / 8);
    constexpr int kLds = 2 * P::kTileBytes + 8 * G::kSlabBytes + 8 * 48 * 4;
    static_assert(ArmStage<T, 8, 1>::kBytes <= 2 * P::kTileBytes + 8 * G::kSlabBytes, "arm stage must fit the aliased area");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tri_att_proj_fwd_kernel<T, KS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
        attr_set = true;
    }
    hipLaunchKernelGGL((tri_att_proj_fwd_kernel<T, KS>), dim3(grid), dim3(512), kLds, st, a, reinterpret_cast<const T*>(x),
                       reinterpret_cast<const T*>(w), reinterpret_cast<const T*>(bias));
    return check_launch("tri_att_proj_fwd_kernel");
}

template <typename T>
static int dispatch_ks(const tgt_triplet_attention_args& a, int C, const void* x, const void* w, const void* bias, hipStream_t st) {
    switch (C) {
        case 64: return launch_proj<T, 4>(a, x, w, bias, st);
        case 128: return launch_proj<T, 8>(a, x, w, bias, st);
        case 256: return launch_proj<T, 16>(a, x, w, bias, st);
        default: return set_error(TGT_ERR_UNSUPPORTED, "projected triplet attention: C=%d not in {64,128,256}", C);
    }
}

int triplet_attention_proj_supported(const tgt_triplet_attention_args* a, int C) {
    return a && a->dropout_p == 0.f && a->N <= 32 && a->D == 16 && a->H % 8 == 0 && (a->dtype == TGT_BF16 || a->dtype == TGT_F16) &&
           (C == 64 || C == 128 || C == 256);
}

int triplet_attention_proj_run(const tgt_triplet_attention_args* a, const void* x, int C, const void* w, const void* bias,
                               hipStream_t st) {
    if (!a || !x || !w || !bias) return set_error(TGT_ERR_INVALID, "projected triplet attention: null argument");
    if (a->B < 0 || a->N < 0 || a->H <= 0) return set_error(TGT_ERR_INVALID, "projected triplet attention: bad sizes");
    if (a->B == 0 || a->N == 0) return TGT_OK;
    if (!triplet_attention_proj_supported(a, C))
        return set_error(TGT_ERR_UNSUPPORTED,
                         "projected triplet attention needs N <= 32, D = 16, H %% 8 == 0, a 16-bit dtype and C in {64,128,256} "
                         "(got N=%d D=%d H=%d dtype=%d C=%d)", a->N, a->D, a->H, a->dtype, C);
    for (int dir = 0; dir < 2; ++dir) {
        if (!a->qkv[dir] || !a->out || !a->mask) return set_error(TGT_ERR_INVALID, "projected triplet attention: null tensor");
        if ((a->ld_qkv[dir] * 2) % 16 || (a->q_off[dir] * 2) % 16 || (a->k_off[dir] * 2) % 16 || (a->v_off[dir] * 2) % 16 ||
            (a->ld_out * 2) % 16 || (a->o_off[dir] * 2) % 16 || ((uintptr_t)a->qkv[dir] % 16) || ((uintptr_t)a->out % 16))
            return set_error(TGT_ERR_INVALID, "projected triplet attention: rows/offsets must be 16-byte aligned");
        if ((a->flags & (TGT_TRI_BIASED | TGT_TRI_GATED)) && !a->eg[dir]) return set_error(TGT_ERR_INVALID, "projected triplet attention: eg missing");
    }
    if (((uintptr_t)x | (uintptr_t)w) % 16) return set_error(TGT_ERR_INVALID, "projected triplet attention: x / w must be 16-byte aligned");
    return a->dtype == TGT_BF16 ? dispatch_ks<bf16_t>(*a, C, x, w, bias, st) : dispatch_ks<f16_t>(*a, C, x, w, bias, st);
}

}  // namespace tgt
