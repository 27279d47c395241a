// Triplet attention FORWARD on 16-wide node tiles (v_mfma_f32_16x16x16) for 33 <= N <= 64 -- BASELINE config 4.
//
// Replaces the einsum -> +bias -> +mask -> softmax -> *gate -> einsum chains of reference
// lib/tgt/layers/triplet.py:213-246 for graphs padded to more than 32 nodes.  Same arithmetic and the same
// decomposition as triplet_attention.hip (workgroup = (graph, direction, group of HG heads), walk over the shared
// node j, slabs of Q rows [i,j] and partner K / V rows through two LDS sets, one barrier per j); what changes is
// the tile: 32-wide tiles pad a 48-node graph to 64 x 64 (two query-tile passes over the walk, each against two key
// tiles: 4 tile pairs for 2.25 tile pairs of work, K / V slabs streamed twice, 128 registers of third-arm state per
// wave).  Here a WAVE owns (head, block of 16 queries) and all NQ = ceil(N/16) key blocks:
//   S^T[key][query] per key block = one 16x16x16 MFMA (K rows x Q rows, depth D = 16): lane (query = l & 15,
//   g = l >> 4) holds keys 4g..4g+3 of the block -- softmax over keys = in-lane values + two lane exchanges
//   (l ^ 16, l ^ 32); V^T = V . I through the matrix core; O^T[d][query] = sum_blocks V^T P^T.  With 16x16x16
//   the accumulator layout (lane = column, rows 4g+q) IS the operand layout (lane = row or column, k = 4g+t): no
//   permuted k-order, no shuffles.
// One pass over the walk whatever N <= 64 is: HG * NQ waves, every lane works on real elements (N = 48: 9 key-block
// products per (head, j) against 16 for the padded 32-wide tiles), third-arm state 2 * NQ * 4 registers per wave.
// The backward stays on the two-tile kernel of triplet_attention.hip for now.
#include <cstdlib>
#include "triplet_common.hpp"

namespace tgt {
namespace t16 {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

template <typename T> struct F4;
template <> struct F4<bf16_t> { typedef s16x4 type; };
template <> struct F4<f16_t> { typedef h16x4 type; };
template <typename T> using frag4_t = typename F4<T>::type;

// C[m][n] += sum_kk A[m][kk] B[kk][n], kk in [0,16): lane l = (x = l & 15, g = l >> 4) supplies A[m = x][kk = 4g + t] /
// B[kk = 4g + t][n = x], t = 0..3, and holds C[m = 4g + q][n = x], q = 0..3
__device__ __forceinline__ f32x4 mma16(s16x4 a, s16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mma16(h16x4 a, h16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); }

template <typename T> struct V4;
template <> struct V4<bf16_t> { typedef __attribute__((ext_vector_type(4))) __bf16 type; };
template <> struct V4<f16_t> { typedef __attribute__((ext_vector_type(4))) _Float16 type; };
template <typename T>
__device__ __forceinline__ frag4_t<T> pack4(const f32x4& v) {
    typename V4<T>::type t;                 // (vector element assignment: hipcc pairs the conversions into two v_cvt_pk)
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = from_f32<T>(v[i]);
    frag4_t<T> f;
    __builtin_memcpy(&f, &t, 8);
    return f;
}
template <typename T>
__device__ __forceinline__ frag4_t<T> ident4(int x, int g) {     // B[kk][n] = (kk == n)
    T t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = from_f32<T>(4 * g + i == x ? 1.f : 0.f);
    frag4_t<T> f;
    __builtin_memcpy(&f, t, 8);
    return f;
}

// gfx950's transposed LDS read: a 16-lane group reads a [4 rows][16 columns] block of 16-bit elements (lane p supplies the address of
// row p >> 2, columns 4 (p & 3) ..) and lane p receives COLUMN p of it -- an operand fragment of the transposed tile, straight
// from a row-major image (round 4: replaces the X . I matrix-core transposes and the 2-byte transposed exchange stores)
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
template <typename T>
__device__ __forceinline__ frag4_t<T> tr_frag(const char* p) {
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_s16x4*>((uint32_t)(uintptr_t)p));
    frag4_t<T> f;
    __builtin_memcpy(&f, &v, 8);
    return f;
}
// exchange block [16 rows][16 columns] of 16-bit values, 8-byte quads XOR-swizzled with (row >> 2): conflict-free for the 8-byte
// stores in accumulator layout (lane = row) and for the transposed reads (lane = column)
__device__ __forceinline__ int xblk_off(int row, int quad) { return row * 32 + ((quad ^ ((row >> 2) & 3)) << 3); }

// slab geometry: 16 * NQ rows of HG heads x 16 channels; the slab_* templates of triplet_common.hpp take it as they take TriGeo
template <typename T, int HG, int NQ>
struct Geo16 {
    static constexpr int kThreads = HG * NQ * 64;
    static constexpr int kRows = 16 * NQ;
    static constexpr int kRowBytes = HG * 16 * (int)sizeof(T);
    static constexpr int kSlots = kRowBytes / 16;
    static constexpr int kSlabBytes = kRows * kRowBytes;
    static constexpr int kRowsPerBankRow = kRowBytes >= 256 ? 1 : 256 / kRowBytes;
    static constexpr int kSwzMask = (kSlots < 16 ? kSlots : 16) - 1;
    __device__ static __forceinline__ int lds_off(int row, int slot) {
        const int f = (row / kRowsPerBankRow) & kSwzMask;
        return row * kRowBytes + ((slot ^ f) << 4);
    }
    __device__ static __forceinline__ int lds_elem(int row, int col) {
        const int bo = col * (int)sizeof(T);
        return lds_off(row, bo >> 4) + (bo & 15);
    }
};

// operand fragment of head hw for slab row `row`: channels 4g .. 4g+3
template <typename T, typename G>
__device__ __forceinline__ frag4_t<T> frag_of(const char* slab, int row, int hw, int g) {
    frag4_t<T> f;
    const uint2 raw = *reinterpret_cast<const uint2*>(slab + G::lds_elem(row, hw * 16 + 4 * g));
    __builtin_memcpy(&f, &raw, 8);
    return f;
}

// third-arm stage: the (x, y) pair records [E of the HG heads | G of the HG heads] of the whole graph + the mask, in memory
// order (x, y); element (query i, key k) of direction 0 sits at (x, y) = (i, k), of direction 1 at (k, i)
template <typename T, int HG, int NQ>
struct Arm16 {
    static constexpr int R = 16 * NQ, kVals = 2 * HG, kRec = kVals * (int)sizeof(T);
    static constexpr int kPitch = R * kRec + 4, kMPitch = R * 4 + 4;
    static constexpr int kOffM = ((R * kPitch + 15) / 16) * 16;
    static constexpr int kBytes = kOffM + R * kMPitch;
};

template <typename T, int HG, int NQ>
__global__ void __launch_bounds__(HG * NQ * 64) tri_att16_fwd_kernel(const tgt_triplet_attention_args a) {
    using G = Geo16<T, HG, NQ>;
    using A = Arm16<T, HG, NQ>;
    using F = frag4_t<T>;
    constexpr int R = G::kRows, kSet = 3 * G::kSlabBytes;        // {Q | K | V}, two sets
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x16 = lane & 15, g = lane >> 4, hw = wave % HG, qb = wave / HG;
    const int ngroups = a.H / HG;
    int bid = blockIdx.x;
    const int grp = bid % ngroups;
    bid /= ngroups;
    const int dir = bid & 1, b = bid >> 1, N = a.N;
    const bool biased = (a.flags & TGT_TRI_BIASED) != 0, gated = (a.flags & TGT_TRI_GATED) != 0;

    // a graph DropPath dropped (graph_scale[b] == 0, see tgt_hip.h): nothing is read or computed, its rows get zeros
    if (a.graph_scale && a.graph_scale[b] == 0.f) {
        const int64_t sz0 = sizeof(T);
        const uint32_t ld0 = (uint32_t)(a.ld_out * sz0);
        const SlabBuf zO = {graph_rsrc(a.out, (int64_t)N * N * a.ld_out * sz0, b), (uint32_t)(a.o_off[dir] * sz0) + (uint32_t)(grp * HG * 16 * sz0),
                            (uint32_t)N * ld0, ld0};
        for (int j = 0; j < N; ++j) slab_store_zero<G, R>(zO, j, 0, N, tid);
        return;
    }

    // ---- third arm of this wave: bias + mask and gate for (query 16 qb + x16, keys 16 kb + 4 g + q) ----
    float biasM[NQ][4], gate[NQ][4];
    {
        const T* eg = reinterpret_cast<const T*>(a.eg[dir]);
        const int64_t ld = a.ld_eg[dir];
        for (int idx = tid; idx < R * R * A::kVals; idx += G::kThreads) {
            const int v = idx % A::kVals, p = idx / A::kVals, y = p % R, x = p / R;
            T val = from_f32<T>(0.f);
            if (x < N && y < N) {
                const bool is_e = v < HG;
                if (is_e ? biased : gated)
                    val = eg[(((int64_t)b * N + x) * N + y) * ld + (is_e ? a.e_off[dir] + grp * HG + v : a.g_off[dir] + grp * HG + v - HG)];
            }
            *reinterpret_cast<T*>(smem + x * A::kPitch + y * A::kRec + v * (int)sizeof(T)) = val;
        }
        for (int idx = tid; idx < R * R; idx += G::kThreads) {
            const int y = idx % R, x = idx / R;
            float m = 0.f;
            if (x < N && y < N && a.mask) m = a.mask[((int64_t)b * N + x) * N + y];
            *reinterpret_cast<float*>(smem + A::kOffM + x * A::kMPitch + y * 4) = m;
        }
        __syncthreads();
        const int i = 16 * qb + x16;
#pragma unroll
        for (int kb = 0; kb < NQ; ++kb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = 16 * kb + 4 * g + q;
                const int xx = dir == 0 ? i : k, yy = dir == 0 ? k : i;
                const char* pp = smem + xx * A::kPitch + yy * A::kRec;
                const float e = to_f32(*reinterpret_cast<const T*>(pp + hw * (int)sizeof(T)));
                const float gl = to_f32(*reinterpret_cast<const T*>(pp + (HG + hw) * (int)sizeof(T)));
                const float m = *reinterpret_cast<const float*>(smem + A::kOffM + xx * A::kMPitch + yy * 4);
                biasM[kb][q] = k < N ? e + m : -INFINITY;
                gate[kb][q] = (i < N && k < N) ? (gated ? fast_sigmoid(gl + m) : 1.f) : 0.f;
            }
        __syncthreads();
    }

    // ---- slabs (buffer-addressed, triplet_common.hpp) ----
    const int64_t sz = sizeof(T), Nl = N;
    const uint32_t hch = (uint32_t)(grp * HG * 16 * sz), lds_ = (uint32_t)(a.ld_qkv[dir] * sz), ldo_ = (uint32_t)(a.ld_out * sz);
    const __amdgpu_buffer_rsrc_t r_src = graph_rsrc(a.qkv[dir], Nl * Nl * a.ld_qkv[dir] * sz, b);
    const SlabBuf bQ = {r_src, (uint32_t)(a.q_off[dir] * sz) + hch, (uint32_t)N * lds_, lds_};
    const SlabBuf bK = {r_src, (uint32_t)(a.k_off[dir] * sz) + hch, dir == 0 ? lds_ : (uint32_t)N * lds_, dir == 0 ? (uint32_t)N * lds_ : lds_};
    const SlabBuf bV = {r_src, (uint32_t)(a.v_off[dir] * sz) + hch, bK.row_stride, bK.j_stride};
    const SlabBuf bO = {graph_rsrc(a.out, Nl * Nl * a.ld_out * sz, b), (uint32_t)(a.o_off[dir] * sz) + hch, (uint32_t)N * ldo_, ldo_};

    uint4 pq[SlabIO<G, R>::kIters], pk[SlabIO<G, R>::kIters], pv[SlabIO<G, R>::kIters];
    slab_issue<G, R>(pq, bQ, 0, 0, N, tid);
    slab_issue<G, R>(pk, bK, 0, 0, N, tid);
    slab_issue<G, R>(pv, bV, 0, 0, N, tid);
    slab_commit<G, R>(pq, smem, tid);
    slab_commit<G, R>(pk, smem + G::kSlabBytes, tid);
    slab_commit<G, R>(pv, smem + 2 * G::kSlabBytes, tid);
    if (N > 1) {
        slab_issue<G, R>(pq, bQ, 1, 0, N, tid);
        slab_issue<G, R>(pk, bK, 1, 0, N, tid);
        slab_issue<G, R>(pv, bV, 1, 0, N, tid);
    }
    __syncthreads();

    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    // one barrier per j: hazards as in triplet_attention.hip (the other set is rewritten at the top of the iteration by the
    // thread that last read those chunks; O goes into this wave's own rows / columns of the Q slab)
    for (int j = 0; j < N; ++j) {
        char* sQ = smem + (j & 1) * kSet;
        char* sK = sQ + G::kSlabBytes;
        char* sV = sK + G::kSlabBytes;
        if (j + 1 < N) {
            char* nQ = smem + ((j + 1) & 1) * kSet;
            slab_commit<G, R>(pq, nQ, tid);
            slab_commit<G, R>(pk, nQ + G::kSlabBytes, tid);
            slab_commit<G, R>(pv, nQ + 2 * G::kSlabBytes, tid);
        }
        if (j + 2 < N) {
            slab_issue<G, R>(pq, bQ, j + 2, 0, N, tid);
            slab_issue<G, R>(pk, bK, j + 2, 0, N, tid);
            slab_issue<G, R>(pv, bV, j + 2, 0, N, tid);
        }
        const F fq = frag_of<T, G>(sQ, 16 * qb + x16, hw, g);
        f32x4 s[NQ];
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < NQ; ++kb) {
            const F fk = frag_of<T, G>(sK, 16 * kb + x16, hw, g);
            s[kb] = mma16(fk, fq, z);             // S^T[key][query]
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                s[kb][q] = s[kb][q] * a.scale + biasM[kb][q];
                mx = fmaxf(mx, s[kb][q]);
            }
        }
        mx = xor16_max(mx);
        mx = fmaxf(mx, xhalf(mx));
        float sum = 0.f;
#pragma unroll
        for (int kb = 0; kb < NQ; ++kb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                s[kb][q] = fast_exp(s[kb][q] - mx);
                sum += s[kb][q];
            }
        sum = xor16_sum(sum);
        sum += xhalf(sum);
        const float inv = fast_rcp(sum);
        f32x4 o = z;
#pragma unroll
        for (int kb = 0; kb < NQ; ++kb) {
#pragma unroll
            for (int q = 0; q < 4; ++q) s[kb][q] = s[kb][q] * inv * gate[kb][q];
            // V^T of the key block (lane d, keys 4g..4g+3) straight from the slab, transposed by the read (round 4: was V . I on the matrix core + a pack)
            o = mma16(tr_frag<T>(sV + G::lds_elem(16 * kb + 4 * g + (x16 >> 2), hw * 16 + 4 * (x16 & 3))), pack4<T>(s[kb]), o);          // O^T[d][query]
        }
        {   // lane (query, g) holds channels 4g .. 4g+3 of its query: 8 bytes into the head's columns of the Q slab
            const F of = pack4<T>(o);
            uint2 raw;
            __builtin_memcpy(&raw, &of, 8);
            *reinterpret_cast<uint2*>(sQ + G::lds_elem(16 * qb + x16, hw * 16 + 4 * g)) = raw;
        }
        __syncthreads();
        slab_store<G, R>(sQ, bO, j, 0, N, tid);
    }
}

// ---------------------------------------------------------------------------
// backward on the same tiles, N <= 48 (NQ = 3: 12 waves, up to 170 registers).
// Per j, two phases with a barrier between them:
//   phase 1, wave = (head, QUERY block): S, dA for every key block, softmax recomputed, dS = P (dA g - delta), A = P g;
//            dE / dG accumulate in registers; dQ^T = s sum_blocks K^T dS^T goes into the head's columns of the Q slab;
//            the wave leaves Q^T, dO^T (operand layout) and its dS / A blocks (transposed on the way: 2-byte stores)
//            in an LDS exchange area;
//   phase 2, wave = (head, KEY block): dK^T = s sum_query-blocks Q^T dS, dV^T = sum dO^T A from the exchange area --
//            complete sums over all queries, no partial tiles -- into the head's columns of the K / V slabs.
// Stores, fused gradient row (ld_dqkv / ld_deg) and the in-kernel bias-gradient column sums as in triplet_attention.hip;
// the per-thread column accumulators live in registers here (24 of them: this kernel has the room).
// ---------------------------------------------------------------------------
template <typename T, int HG, int NQ, bool CS>
__global__ void __launch_bounds__(HG * NQ * 64) tri_att16_bwd_kernel(const tgt_triplet_attention_args a) {
    using G = Geo16<T, HG, NQ>;
    using A = Arm16<T, HG, NQ>;
    using F = frag4_t<T>;
    constexpr int R = G::kRows, kSet = 4 * G::kSlabBytes;        // {Q | dO | K | V}, THREE sets: the slabs land in LDS two steps ahead (LDS-DMA)
    constexpr int kBlk = 16 * 16 * (int)sizeof(T);               // one 16x16 exchange block
    constexpr int kXWave = (2 + 2 * NQ) * kBlk;                  // per (head, query block): Q^T, dO^T, dS[kb], A[kb]
    constexpr int kOffX = 3 * kSet;
    constexpr int kPieces = G::kSlabBytes / 1024;                // 1 KB (one wave instruction) pieces per slab
    static_assert(G::kSlabBytes % 1024 == 0 && 4 * kPieces == 2 * HG * NQ, "every wave loads exactly two pieces of a set");
    constexpr int E = 16 / (int)sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x16 = lane & 15, g = lane >> 4, hw = wave % HG, qb = wave / HG;
    const int pr = x16 >> 2, c2 = x16 & 3;          // the address a lane supplies to a transposed read: row pr, column quad c2 of its group's block
    const int ngroups = a.H / HG;
    int bid = blockIdx.x;
    const int grp = bid % ngroups;
    bid /= ngroups;
    const int dir = bid & 1, b = bid >> 1, N = a.N;
    const bool biased = (a.flags & TGT_TRI_BIASED) != 0, gated = (a.flags & TGT_TRI_GATED) != 0;

    float biasM[NQ][4], gate[NQ][4], dE[NQ][4], dG[NQ][4];
    {
        const T* eg = reinterpret_cast<const T*>(a.eg[dir]);
        const int64_t ld = a.ld_eg[dir];
        for (int idx = tid; idx < R * R * A::kVals; idx += G::kThreads) {
            const int v = idx % A::kVals, p = idx / A::kVals, y = p % R, x = p / R;
            T val = from_f32<T>(0.f);
            if (x < N && y < N) {
                const bool is_e = v < HG;
                if (is_e ? biased : gated)
                    val = eg[(((int64_t)b * N + x) * N + y) * ld + (is_e ? a.e_off[dir] + grp * HG + v : a.g_off[dir] + grp * HG + v - HG)];
            }
            *reinterpret_cast<T*>(smem + x * A::kPitch + y * A::kRec + v * (int)sizeof(T)) = val;
        }
        for (int idx = tid; idx < R * R; idx += G::kThreads) {
            const int y = idx % R, x = idx / R;
            float m = 0.f;
            if (x < N && y < N && a.mask) m = a.mask[((int64_t)b * N + x) * N + y];
            *reinterpret_cast<float*>(smem + A::kOffM + x * A::kMPitch + y * 4) = m;
        }
        __syncthreads();
        const int i = 16 * qb + x16;
#pragma unroll
        for (int kb = 0; kb < NQ; ++kb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = 16 * kb + 4 * g + q;
                const int xx = dir == 0 ? i : k, yy = dir == 0 ? k : i;
                const char* pp = smem + xx * A::kPitch + yy * A::kRec;
                const float e = to_f32(*reinterpret_cast<const T*>(pp + hw * (int)sizeof(T)));
                const float gl = to_f32(*reinterpret_cast<const T*>(pp + (HG + hw) * (int)sizeof(T)));
                const float m = *reinterpret_cast<const float*>(smem + A::kOffM + xx * A::kMPitch + yy * 4);
                const bool valid = i < N && k < N;
                biasM[kb][q] = valid ? e + m : -INFINITY;          // padding queries get weight exactly 0: they feed sums over queries
                gate[kb][q] = valid ? (gated ? fast_sigmoid(gl + m) : 1.f) : 0.f;
                dE[kb][q] = dG[kb][q] = 0.f;
            }
        __syncthreads();
    }

    const int64_t sz = sizeof(T), Nl = N;
    const int64_t ldq = a.ld_dqkv[dir] ? a.ld_dqkv[dir] : a.ld_qkv[dir];
    const int64_t lde = a.ld_deg[dir] ? a.ld_deg[dir] : a.ld_eg[dir];
    const uint32_t hch = (uint32_t)(grp * HG * 16 * sz);
    const uint32_t lds_ = (uint32_t)(a.ld_qkv[dir] * sz), ldg_ = (uint32_t)(ldq * sz), ldo_ = (uint32_t)(a.ld_out * sz);
    const __amdgpu_buffer_rsrc_t r_src = graph_rsrc(a.qkv[dir], Nl * Nl * a.ld_qkv[dir] * sz, b);
    const __amdgpu_buffer_rsrc_t r_grd = graph_rsrc(a.d_qkv[dir], Nl * Nl * ldq * sz, b);
    const __amdgpu_buffer_rsrc_t r_do = graph_rsrc(a.d_out, Nl * Nl * a.ld_out * sz, b);
    const uint32_t qo = (uint32_t)(a.q_off[dir] * sz) + hch, ko = (uint32_t)(a.k_off[dir] * sz) + hch, vo = (uint32_t)(a.v_off[dir] * sz) + hch;
    const SlabBuf bQ = {r_src, qo, (uint32_t)N * lds_, lds_};
    const SlabBuf bK = {r_src, ko, dir == 0 ? lds_ : (uint32_t)N * lds_, dir == 0 ? (uint32_t)N * lds_ : lds_};
    const SlabBuf bV = {r_src, vo, bK.row_stride, bK.j_stride};
    const SlabBuf bO = {r_do, (uint32_t)(a.o_off[dir] * sz) + hch, (uint32_t)N * ldo_, ldo_};
    const SlabBuf gQ = {r_grd, qo, (uint32_t)N * ldg_, ldg_};
    const SlabBuf gK = {r_grd, ko, dir == 0 ? ldg_ : (uint32_t)N * ldg_, dir == 0 ? (uint32_t)N * ldg_ : ldg_};
    const SlabBuf gV = {r_grd, vo, gK.row_stride, gK.j_stride};

    static_assert(SlabIO<G, R>::kIters == 1, "one chunk per thread");
    const bool has_chunk = tid < SlabIO<G, R>::kChunks;
    const int crow = tid / G::kSlots, cslot = tid % G::kSlots;
    // column sums of dQ, dK, dV (the projection's bias gradient) in registers, from the fp32 accumulators: dQ^T of this wave's
    // query block (lane = query, d = 4g + q), dK^T / dV^T of its key block (lane = key); lanes and blocks are summed after the walk
    f32x4 cq = {0.f, 0.f, 0.f, 0.f}, ck = {0.f, 0.f, 0.f, 0.f}, cv = {0.f, 0.f, 0.f, 0.f};
    // a graph DropPath dropped (graph_scale[b] == 0) receives an all-zero d_out: zeros to its gradient rows, nothing read or
    // computed; dE / dG (zero-initialised above) and the column sums leave through the common tail below
    const bool dead = a.graph_scale && a.graph_scale[b] == 0.f;          // workgroup-uniform
    if (dead) {
        for (int j = 0; j < N; ++j) {
            slab_store_zero<G, R>(gQ, j, 0, N, tid);
            slab_store_zero<G, R>(gK, j, 0, N, tid);
            slab_store_zero<G, R>(gV, j, 0, N, tid);
        }
    } else {
    // Loads (round 4, as triplet_attention_bwd2.hip): buffer_load ... lds straight into set (step % 3), two steps ahead; each wave
    // owns two 1 KB pieces of a set (piece t = wave and wave + #waves of the 4 * kPieces pieces {Q | dO | K | V}); a lane's 16 bytes
    // land at physical chunk (row, slot'), i.e. it fetches slot' ^ swizzle(row) of the row (the LDS image keeps Geo16's layout);
    // rows past N and steps past the end are out of range (no bytes written: the sets are zero-filled first when N < R).
    // Inline assembly with hand-placed waits: hipcc would order every LDS read behind a builtin LDS-DMA with vmcnt(0).
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const uint32_t sbase = (uint32_t)(uintptr_t)smem;
    const uint64_t src_base = (uint64_t)(uintptr_t)a.qkv[dir] + (uint64_t)b * (uint64_t)(Nl * Nl * a.ld_qkv[dir] * sz);
    const uint64_t do_base = (uint64_t)(uintptr_t)a.d_out + (uint64_t)b * (uint64_t)(Nl * Nl * a.ld_out * sz);
    const uint32_t src_bytes = (uint32_t)(Nl * Nl * a.ld_qkv[dir] * sz), do_bytes = (uint32_t)(Nl * Nl * a.ld_out * sz);
    uint32_t l_off[2], l_voff[2], l_chan[2], l_js[2];
    bool l_do[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = wave_u + u * HG * NQ, slab = t / kPieces, piece = t % kPieces;      // (wave-uniform)
        const int c = piece * 64 + lane, row = c / G::kSlots, pslot = c % G::kSlots;
        const int f = (row / G::kRowsPerBankRow) & G::kSwzMask;
        const SlabBuf& sb = slab == 0 ? bQ : (slab == 1 ? bO : (slab == 2 ? bK : bV));
        l_off[u] = (uint32_t)(slab * G::kSlabBytes + piece * 1024);
        l_voff[u] = row < N ? (uint32_t)row * sb.row_stride + (uint32_t)((pslot ^ f) << 4) : 0x80000000u;
        l_chan[u] = sb.chan;
        l_js[u] = sb.j_stride;
        l_do[u] = slab == 1;
    }
    auto dma = [&](int jj, int set) {
        const bool live = jj < N;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const uint64_t base = l_do[u] ? do_base : src_base;
            const u32x4_t rs = {(uint32_t)base, (uint32_t)(base >> 32) & 0xffffu, live ? (l_do[u] ? do_bytes : src_bytes) : 0u, 0x00020000u};
            const uint32_t lds = sbase + (uint32_t)(set * kSet) + l_off[u];
            const uint32_t so = l_chan[u] + (uint32_t)jj * l_js[u];
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds"
                         :: "s"(lds), "v"(l_voff[u]), "s"(rs), "s"(so) : "memory");
        }
    };
    // every thread issues three result stores per step (out of range where it owns no chunk / no row): the wait counts below
    // are the same number in every wave
    const bool own = has_chunk && crow < N;
    const uint32_t kOOB = 0x80000000u;
    const uint32_t wQ = own ? (uint32_t)crow * gQ.row_stride + (uint32_t)cslot * 16u : kOOB;
    const uint32_t wK = own ? (uint32_t)crow * gK.row_stride + (uint32_t)cslot * 16u : kOOB;
    const int chunk = has_chunk ? G::lds_off(crow, cslot) : 0;
    if (N < R) {
        const uint4 z4 = make_uint4(0, 0, 0, 0);
        for (int o = tid * 16; o < 3 * kSet; o += G::kThreads * 16) *reinterpret_cast<uint4*>(smem + o) = z4;
        __syncthreads();
    }
    dma(0, 0);
    dma(1, 1);
    {
        const u32x4_t z4 = {0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 3; ++t) __builtin_amdgcn_raw_buffer_store_b128(z4, r_grd, (int)kOOB, 16 * t, TGT_ST_AUX);
    }
    // queue: loads(0) x2, loads(1) x2, 3 stores
    asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" ::: "memory");

#ifndef TGT_T16_PRIO
#define TGT_T16_PRIO 1             // static wave priority: the younger half of the workgroup at s_setprio 1 (three alternating pairs at N = 48: 0.692 / 0.690 / 0.699 against 0.697 / 0.714 / 0.704 ms)
#endif
    if (TGT_T16_PRIO == 1 && wave_u >= HG * NQ / 2) __builtin_amdgcn_s_setprio(1);
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    char* xw = smem + kOffX + (hw * NQ + qb) * kXWave;          // this wave's exchange blocks (phase 1)
    int cur = 0;
    for (int j = 0; j < N; ++j) {
        char* sQ = smem + cur * kSet;
        char* sO = sQ + G::kSlabBytes;
        char* sK = sQ + 2 * G::kSlabBytes;
        char* sV = sQ + 3 * G::kSlabBytes;
        // ---- phase 1: (head, query block) ----
        {
            const F fq = frag_of<T, G>(sQ, 16 * qb + x16, hw, g), fo = frag_of<T, G>(sO, 16 * qb + x16, hw, g);
            f32x4 s[NQ], da[NQ];
            float mx = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < NQ; ++kb) {
                const F fk = frag_of<T, G>(sK, 16 * kb + x16, hw, g), fv = frag_of<T, G>(sV, 16 * kb + x16, hw, g);
                s[kb] = mma16(fk, fq, z);                 // S^T[key][query]
                da[kb] = mma16(fv, fo, z);                // dA^T[key][query]
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    s[kb][q] = s[kb][q] * a.scale + biasM[kb][q];
                    mx = fmaxf(mx, s[kb][q]);
                }
            }
            mx = xor16_max(mx);
            mx = fmaxf(mx, xhalf(mx));
            if (mx == -INFINITY) mx = 0.f;                // padding query: every weight is exactly 0
            float sum = 0.f;
#pragma unroll
            for (int kb = 0; kb < NQ; ++kb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    s[kb][q] = fast_exp(s[kb][q] - mx);
                    sum += s[kb][q];
                }
            sum = xor16_sum(sum);
            sum += xhalf(sum);
            const float inv = sum > 0.f ? fast_rcp(sum) : 0.f;
            float delta = 0.f;
#pragma unroll
            for (int kb = 0; kb < NQ; ++kb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float p = s[kb][q] * inv, dp = da[kb][q] * gate[kb][q];
                    delta += p * dp;
                    if (gated) dG[kb][q] += da[kb][q] * p;
                    s[kb][q] = p;
                    da[kb][q] = dp;
                }
            delta = xor16_sum(delta);
            delta += xhalf(delta);
            // Q^T, dO^T in operand layout (lane d, queries 4g..4g+3) for phase 2: a copy, because this wave's dQ rows replace its
            // Q rows in the slab before the other waves of the head get there
            {
                const F qTf = tr_frag<T>(sQ + G::lds_elem(16 * qb + 4 * g + pr, hw * 16 + 4 * c2));
                const F oTf = tr_frag<T>(sO + G::lds_elem(16 * qb + 4 * g + pr, hw * 16 + 4 * c2));
                uint2 r0, r1;
                __builtin_memcpy(&r0, &qTf, 8);
                __builtin_memcpy(&r1, &oTf, 8);
                *reinterpret_cast<uint2*>(xw + (x16 * 16 + 4 * g) * (int)sizeof(T)) = r0;
                *reinterpret_cast<uint2*>(xw + kBlk + (x16 * 16 + 4 * g) * (int)sizeof(T)) = r1;
            }
            f32x4 dq = z;
#pragma unroll
            for (int kb = 0; kb < NQ; ++kb) {
                f32x4 dsv, atv;
                char* xs = xw + (2 + kb) * kBlk;
                char* xa = xw + (2 + NQ + kb) * kBlk;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float ds = s[kb][q] * (da[kb][q] - delta);
                    if (biased) dE[kb][q] += ds;
                    dsv[q] = ds * a.scale;
                    atv[q] = s[kb][q] * gate[kb][q];
                }
                // block[query][key], as the accumulator holds it (lane = query, keys 4g..4g+3): one 8-byte store each; phase 2
                // reads them transposed
                const F dsf = pack4<T>(dsv), atf = pack4<T>(atv);
                uint2 w0, w1;
                __builtin_memcpy(&w0, &dsf, 8);
                __builtin_memcpy(&w1, &atf, 8);
                *reinterpret_cast<uint2*>(xs + xblk_off(x16, g)) = w0;
                *reinterpret_cast<uint2*>(xa + xblk_off(x16, g)) = w1;
                // K^T of the key block (lane d, keys 4g..4g+3) straight from the slab, transposed by the read
                dq = mma16(tr_frag<T>(sK + G::lds_elem(16 * kb + 4 * g + pr, hw * 16 + 4 * c2)), dsf, dq);          // dQ^T[d][query]
            }
            const F dqf = pack4<T>(dq);
            uint2 raw;
            __builtin_memcpy(&raw, &dqf, 8);
            *reinterpret_cast<uint2*>(sQ + G::lds_elem(16 * qb + x16, hw * 16 + 4 * g)) = raw;
            if constexpr (CS) cq += dq;
        }
        __syncthreads();
        // the slabs of step j + 2 into the set of step j - 1: every thread read its result chunks of that set before it got here
        dma(j + 2, cur == 0 ? 2 : cur - 1);
        // ---- phase 2: (head, KEY block = this wave's block index) ----
        {
            const int kb = qb;
            f32x4 dk = z, dv = z;
#pragma unroll
            for (int q2 = 0; q2 < NQ; ++q2) {
                const char* xo = smem + kOffX + (hw * NQ + q2) * kXWave;
                F qTf, oTf, dsb, ab;
                const uint2 r0 = *reinterpret_cast<const uint2*>(xo + (x16 * 16 + 4 * g) * (int)sizeof(T));
                const uint2 r1 = *reinterpret_cast<const uint2*>(xo + kBlk + (x16 * 16 + 4 * g) * (int)sizeof(T));
                __builtin_memcpy(&qTf, &r0, 8);
                __builtin_memcpy(&oTf, &r1, 8);
                // dS / A of (query block q2, key block kb) as the B operand: lane = key, queries 4g..4g+3 -- transposed reads
                dsb = tr_frag<T>(xo + (2 + kb) * kBlk + xblk_off(4 * g + pr, c2));
                ab = tr_frag<T>(xo + (2 + NQ + kb) * kBlk + xblk_off(4 * g + pr, c2));
                dk = mma16(qTf, dsb, dk);                 // dK^T[d][key] += Q^T[d][queries] dS[queries][key]
                dv = mma16(oTf, ab, dv);                  // dV^T[d][key] += dO^T[d][queries] A[queries][key]
            }
            const F dkf = pack4<T>(dk), dvf = pack4<T>(dv);
            uint2 r0, r1;
            __builtin_memcpy(&r0, &dkf, 8);
            __builtin_memcpy(&r1, &dvf, 8);
            *reinterpret_cast<uint2*>(sK + G::lds_elem(16 * kb + x16, hw * 16 + 4 * g)) = r0;
            *reinterpret_cast<uint2*>(sV + G::lds_elem(16 * kb + x16, hw * 16 + 4 * g)) = r1;
            if constexpr (CS) { ck += dk; cv += dv; }
        }
        // queue: loads(j+1) x2, stores(j-1) x3, loads(j+2) x2 -- the slabs of step j + 1 have landed for every wave behind this
        asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        {
            const u32x4_t o0 = *reinterpret_cast<const u32x4_t*>(sQ + chunk);
            const u32x4_t o2 = *reinterpret_cast<const u32x4_t*>(sK + chunk);
            const u32x4_t o3 = *reinterpret_cast<const u32x4_t*>(sV + chunk);
            __builtin_amdgcn_raw_buffer_store_b128(o0, r_grd, (int)wQ, (int)(gQ.chan + (uint32_t)j * gQ.j_stride), TGT_ST_AUX);
            __builtin_amdgcn_raw_buffer_store_b128(o2, r_grd, (int)wK, (int)(gK.chan + (uint32_t)j * gK.j_stride), TGT_ST_AUX);
            __builtin_amdgcn_raw_buffer_store_b128(o3, r_grd, (int)wK, (int)(gV.chan + (uint32_t)j * gV.j_stride), TGT_ST_AUX);
        }
        cur = cur == 2 ? 0 : cur + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the out-of-range loads past the end of the walk)
    }      // (!dead)
    __syncthreads();
    // ---- third-arm gradients: through the stage image, then 2-byte scatter (as arm_stage_store_grad) ----
    {
        const int i = 16 * qb + x16;
#pragma unroll
        for (int kb = 0; kb < NQ; ++kb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = 16 * kb + 4 * g + q;
                const int xx = dir == 0 ? i : k, yy = dir == 0 ? k : i;
                char* pp = smem + xx * A::kPitch + yy * A::kRec;
                const float gq = gate[kb][q];
                *reinterpret_cast<T*>(pp + hw * (int)sizeof(T)) = from_f32<T>(dE[kb][q]);
                *reinterpret_cast<T*>(pp + (HG + hw) * (int)sizeof(T)) = from_f32<T>(dG[kb][q] * gq * (1.f - gq));
            }
    }
    __syncthreads();
    float part = 0.f;
    if (biased || gated) {
        static_assert(G::kThreads % A::kVals == 0, "a thread must own one E/G column");
        T* deg = reinterpret_cast<T*>(a.d_eg[dir]);
        for (int idx = tid; idx < R * R * A::kVals; idx += G::kThreads) {
            const int v = idx % A::kVals, p = idx / A::kVals, y = p % R, x = p / R;
            const bool is_e = v < HG;
            if (x < N && y < N && (is_e ? biased : gated)) {
                const T val = *reinterpret_cast<const T*>(smem + x * A::kPitch + y * A::kRec + v * (int)sizeof(T));
                deg[(((int64_t)b * N + x) * N + y) * lde + (is_e ? a.e_off[dir] + grp * HG + v : a.g_off[dir] + grp * HG + v - HG)] = val;
                part += to_f32(val);
            }
        }
    }
    if constexpr (CS) {
        __syncthreads();                                         // the stage image is dead: its space takes the fold
        float* cs = reinterpret_cast<float*>(smem);              // [tensor][head][block][16 d], then one dE / dG partial per thread
        constexpr int kFold = 3 * HG * NQ * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float vq = group_sum<16>(cq[q]), vk = group_sum<16>(ck[q]), vv = group_sum<16>(cv[q]);
            if (x16 == 0) {
                cs[((0 * HG + hw) * NQ + qb) * 16 + 4 * g + q] = vq;
                cs[((1 * HG + hw) * NQ + qb) * 16 + 4 * g + q] = vk;
                cs[((2 * HG + hw) * NQ + qb) * 16 + 4 * g + q] = vv;
            }
        }
        cs[kFold + tid] = part;
        __syncthreads();
        float* row = a.d_qkv_colsum[dir] + (int64_t)b * ldq + grp * HG * 16;
        if (tid < 3 * HG * 16) {
            const int t = tid / (HG * 16), col = tid % (HG * 16), h = col / 16, d = col % 16;
            float v = 0.f;
#pragma unroll
            for (int blk = 0; blk < NQ; ++blk) v += cs[((t * HG + h) * NQ + blk) * 16 + d];
            row[(t == 0 ? a.q_off[dir] : (t == 1 ? a.k_off[dir] : a.v_off[dir])) + col] = v;
        }
        if ((biased || gated) && tid < A::kVals) {
            float v = 0.f;
            for (int t = tid; t < G::kThreads; t += A::kVals) v += cs[kFold + t];
            float* erow = a.d_eg_colsum[dir] + (int64_t)b * lde;
            if (tid < HG) { if (biased) erow[a.e_off[dir] + grp * HG + tid] = v; }
            else if (gated) erow[a.g_off[dir] + grp * HG + tid - HG] = v;
        }
    }
}

template <typename T, int HG, int NQ>
static int launch_bwd(const tgt_triplet_attention_args& a, hipStream_t st) {
    using G = Geo16<T, HG, NQ>;
    constexpr int E = 16 / (int)sizeof(T);
    constexpr int kArm = Arm16<T, HG, NQ>::kBytes;
    constexpr int kWalk = 3 * 4 * G::kSlabBytes + HG * NQ * (2 + 2 * NQ) * 16 * 16 * (int)sizeof(T);
    constexpr int kCs = (3 * HG * NQ * 16 + G::kThreads) * 4;
    constexpr int kLds = (kArm > kWalk ? kArm : kWalk) > kCs ? (kArm > kWalk ? kArm : kWalk) : kCs;
    static_assert(kLds <= 160 * 1024, "LDS");
    const bool cs = a.d_qkv_colsum[0] != nullptr;
    const int grid = a.B * 2 * (a.H / HG);
    if (cs) {
        static bool attr_set[16] = {};                 // per device (common.hpp: dyn_lds_once)
        if (!dyn_lds_once(attr_set, reinterpret_cast<const void*>(&tri_att16_bwd_kernel<T, HG, NQ, true>), kLds))
            return set_error(TGT_ERR_LAUNCH, "tri_att16_bwd_kernel: cannot reserve %d bytes of LDS", kLds);
        hipLaunchKernelGGL((tri_att16_bwd_kernel<T, HG, NQ, true>), dim3(grid), dim3(G::kThreads), kLds, st, a);
    } else {
        static bool attr_set[16] = {};                 // per device (common.hpp: dyn_lds_once)
        if (!dyn_lds_once(attr_set, reinterpret_cast<const void*>(&tri_att16_bwd_kernel<T, HG, NQ, false>), kLds))
            return set_error(TGT_ERR_LAUNCH, "tri_att16_bwd_kernel: cannot reserve %d bytes of LDS", kLds);
        hipLaunchKernelGGL((tri_att16_bwd_kernel<T, HG, NQ, false>), dim3(grid), dim3(G::kThreads), kLds, st, a);
    }
    return check_launch("tri_att16_bwd_kernel");
}

template <typename T, int HG, int NQ>
static int launch(const tgt_triplet_attention_args& a, hipStream_t st) {
    using G = Geo16<T, HG, NQ>;
    constexpr int kArm = Arm16<T, HG, NQ>::kBytes, kSlabs = 2 * 3 * G::kSlabBytes;
    constexpr int kLds = kArm > kSlabs ? kArm : kSlabs;
    static_assert(kLds <= 160 * 1024, "LDS");
    static bool attr_set[16] = {};                 // per device (common.hpp: dyn_lds_once)
    if (!dyn_lds_once(attr_set, reinterpret_cast<const void*>(&tri_att16_fwd_kernel<T, HG, NQ>), kLds))
        return set_error(TGT_ERR_LAUNCH, "tri_att16_fwd_kernel: cannot reserve %d bytes of LDS", kLds);
    hipLaunchKernelGGL((tri_att16_fwd_kernel<T, HG, NQ>), dim3(a.B * 2 * (a.H / HG)), dim3(G::kThreads), kLds, st, a);
    return check_launch("tri_att16_fwd_kernel");
}

template <typename T>
static int run(const tgt_triplet_attention_args& a, hipStream_t st) {
    return a.N <= 48 ? launch<T, 4, 3>(a, st) : launch<T, 4, 4>(a, st);
}

}  // namespace t16

// backward on 16-wide tiles: as the forward (N <= 48: 4 heads x 3 blocks; 49..64: 2 heads x 4 blocks)
// The same kernels for 17 <= N <= 32 (two blocks, 8 heads x 2 waves) were measured in round 2 and are gone: forward 254.8 -> 242.5 us
// but backward 484 -> 617 us (128 registers at 16 waves spill, two barriers per j); N <= 32 stays on the 32-wide kernels.
bool tri_att16_bwd_eligible(const tgt_triplet_attention_args& a) {
    return (a.dtype == TGT_BF16 || a.dtype == TGT_F16) && a.D == 16 && ((a.N > 32 && a.N <= 48 && a.H % 4 == 0) || (a.N > 48 && a.N <= 64 && a.H % 2 == 0)) &&
           !(a.dropout_p > 0.f);
}
int tri_att16_bwd_run(const tgt_triplet_attention_args& a, hipStream_t st) {
    // four blocks: 2 heads x 4 waves (8 waves, up to 256 registers; 64-byte row pieces) -- 16 waves would leave 128 registers
    if (a.N > 48) return a.dtype == TGT_BF16 ? t16::launch_bwd<bf16_t, 2, 4>(a, st) : t16::launch_bwd<f16_t, 2, 4>(a, st);
    return a.dtype == TGT_BF16 ? t16::launch_bwd<bf16_t, 4, 3>(a, st) : t16::launch_bwd<f16_t, 4, 3>(a, st);
}

// forward on 16-wide tiles: 16-bit, D = 16, 33 <= N <= 64, H a multiple of 4, no attention dropout
bool tri_att16_fwd_eligible(const tgt_triplet_attention_args& a) {
    return (a.dtype == TGT_BF16 || a.dtype == TGT_F16) && a.D == 16 && a.N > 32 && a.N <= 64 && a.H % 4 == 0 && !(a.dropout_p > 0.f);
}
int tri_att16_fwd_run(const tgt_triplet_attention_args& a, hipStream_t st) {
    return a.dtype == TGT_BF16 ? t16::run<bf16_t>(a, st) : t16::run<f16_t>(a, st);
}

}  // namespace tgt
