// Triplet attention backward for the hot shapes (16-bit, D = 16, N <= 32, H % 8 == 0, no attention dropout) -- round-4 rebuild.
//
// Same math and the same workgroup decomposition as tri_att_bwd_kernel (triplet_attention.hip: workgroup = (graph, direction,
// 8 heads), wave = head, walk over the shared node j, slabs {Q | dO | K | V} of 32 rows x 256 B in two LDS sets, third-arm tiles
// and dE / dG accumulators in registers), reference lib/tgt/layers/triplet.py:213-246 (autograd of that chain), SURVEY App. A.4.
// What changed, and why (tools/probes/tri_bwd_probe.py: the old kernel spends ~1900 cycles per j in tile math, ~1500 waiting at
// the barrier for the partner wave of its SIMD, ~2500 around the stores + LDS column sums):
//   * no identity-matrix MFMAs.  The old kernel re-laid out Q, dO, K (operands) and dS, A (computed tiles) by multiplying with I
//     on the matrix core: 7 of its 15 MFMAs per (head, j) plus 40 v_cvt_pk.  Here
//       - K^T, Q^T, dO^T come straight out of the slabs with gfx950's transposed LDS read (ds_read_b64_tr_b16: a 16-lane
//         group reads a [4 rows][16 columns] block and every lane receives one COLUMN of it),
//       - dS and A go through a per-wave 2 x 2 KB exchange tile in LDS (written in accumulator layout with 8-byte stores, read
//         back transposed by the same instruction); no barrier: one wave, LDS operations of a wave are ordered,
//       - dK^T / dV^T are v_mfma_f32_16x16x32 (M = d = 16, K = i = 32: one instruction per 16-key tile, no half-empty 32x32
//         tiles): 8 MFMAs per (head, j), 160 matrix-pipe cycles instead of 480.
//   * column sums (the projection's bias gradient) accumulate in 16 REGISTERS from the fp32 accumulators (the old kernel kept
//     48 KB of per-thread accumulators in LDS: 6 ds_read_b128 + 6 ds_write_b128 + 48 VALU per thread and j); reduced across
//     lanes once, after the walk.  (Sums of the unrounded fp32 results: at least as close to the exact sums as sums of the
//     rounded rows.)
//   * straight-line j-loop: rows past N and steps past the end of the walk are out-of-range buffer accesses (loads return 0,
//     stores are dropped), so there is no branch around a memory operation and the waits are exact counts.
//   * slab swizzle: 16-byte slot index XOR rotl4(row): 8 consecutive rows put one head's 32 bytes on 8 different 32-byte bank
//     groups (conflict-free transposed reads), rows r / r + 8 still differ (conflict-free ds_read_b128 of the operand rows).
#include <cstdlib>
#include "triplet_common.hpp"

namespace tgt {
namespace bwd2 {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int HG = 8, D = 16, kThreads = 512;
constexpr int kRow = 256;                  // bytes of one slab row (8 heads x 16 channels x 2 bytes)
constexpr int kSlab = 32 * kRow;           // 8 KB
constexpr int kSet = 4 * kSlab;            // {Q | dO | K | V}
constexpr int kXch = 2 * 2048;             // per wave: dS tile + A tile, each [2 key tiles][32 i][16 k] 16-bit
// LDS: NS slab sets | 8 exchange tiles | one dE / dG partial per thread.  NS = 2: slabs prefetched into registers and committed
// by ds_write (one step ahead); NS = 3 (DMA): slabs land in LDS directly (buffer_load ... lds), two steps ahead.
template <bool DMA> struct Lay {
#ifndef TGT_BWD2_DEPTH
#define TGT_BWD2_DEPTH 2             // LDS-DMA: slabs land this many steps ahead (DEPTH + 1 slab sets; 3 = all 160 KB of LDS)
#endif
    static constexpr int kDepth = TGT_BWD2_DEPTH;
    static constexpr int kSets = DMA ? kDepth + 1 : 2;
    static constexpr int kOffXch = kSets * kSet;
    static constexpr int kOffPart = kOffXch + HG * kXch;
    static constexpr int kLds = kOffPart;          // (the per-thread dE / dG partials of the epilogue live in the dead exchange tiles)
};
typedef __attribute__((address_space(3))) void lds_void;
#if TGT_LD_AUX == 2
#define TGT_BWD2_LD_POLICY " nt"           // the streaming-load policy of common.hpp, spelled for the inline-assembly loads
#else
#define TGT_BWD2_LD_POLICY ""
#endif

__device__ __forceinline__ int swz(int row) { return ((row & 7) << 1) | ((row >> 3) & 1); }
// byte offset inside a slab of byte column `cb` of row `row`
__device__ __forceinline__ int slab_off(int row, int cb) { return row * kRow + ((((cb >> 4) ^ swz(row)) & 15) << 4) + (cb & 15); }

__device__ __forceinline__ s16x4 tr_read(uint32_t addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_s16x4*>(addr));
}
template <typename T>
__device__ __forceinline__ frag_t<T> frag_of(s16x4 lo, s16x4 hi) {
    frag_t<T> f;
    __builtin_memcpy(&f, &lo, 8);
    __builtin_memcpy(reinterpret_cast<char*>(&f) + 8, &hi, 8);
    return f;
}
__device__ __forceinline__ f32x4 mma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mma16(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

template <typename T> struct Vec4;
template <> struct Vec4<bf16_t> { typedef __attribute__((ext_vector_type(4))) __bf16 type; };
template <> struct Vec4<f16_t> { typedef __attribute__((ext_vector_type(4))) _Float16 type; };
// four fp32 -> four 16-bit values (vector element assignment: hipcc pairs them into two v_cvt_pk)
template <typename T>
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
    typename Vec4<T>::type t;
    t[0] = from_f32<T>(a); t[1] = from_f32<T>(b); t[2] = from_f32<T>(c); t[3] = from_f32<T>(d);
    uint2 v;
    __builtin_memcpy(&v, &t, 8);
    return v;
}
template <typename T>
__device__ __forceinline__ uint2 frag_half(const frag_t<T>& f, int h) {
    uint2 v;
    __builtin_memcpy(&v, reinterpret_cast<const char*>(&f) + 8 * h, 8);
    return v;
}

// Third-arm staging with 16-byte global accesses.  A pair's record [E of the 8 heads | G of the 8 heads] is two 16-byte pieces of
// its E/G row; the generic arm_stage_load / arm_stage_store_grad of triplet_common.hpp move them as sixteen 2-byte values per pair
// (256 + 256 wave instructions per workgroup around a walk of 768 16-byte stores).  The LDS image is ArmStage's (records 4-byte
// aligned: four dword accesses per piece), so arm_stage_read / arm_stage_put_grad work on it unchanged.
// Requires 16-byte aligned pieces: tri_att_bwd2_eligible checks the row lengths and offsets.
template <typename T>
__device__ __forceinline__ void arm_load16(const ThirdArm& ta, int b, int dir, int g, int N, char* lds, int tid) {
    using A = ArmStage<T, HG, 1>;
    const int pitch = A::pitch(dir), mpitch = A::mpitch(dir);
    const char* eg = reinterpret_cast<const char*>(ta.eg);
    for (int idx = tid; idx < 32 * 32 * 2; idx += kThreads) {
        const int half = idx & 1, pr = idx >> 1, yy = pr & 31, xx = pr >> 5;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (xx < N && yy < N && (half == 0 ? ta.biased : ta.gated))
            v = *reinterpret_cast<const uint4*>(eg + ((((int64_t)b * N + xx) * N + yy) * ta.ld + (half == 0 ? ta.e_off : ta.g_off) + g * HG) * 2);
        uint32_t* dst = reinterpret_cast<uint32_t*>(lds + xx * pitch + yy * A::kPairBytes + half * 16);
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
    for (int idx = tid; idx < 32 * 32; idx += kThreads) {
        const int yy = idx & 31, xx = idx >> 5;
        float m = 0.f;
        if (xx < N && yy < N && ta.mask) m = ta.mask[((int64_t)b * N + xx) * N + yy];
        *reinterpret_cast<float*>(lds + A::kOffM + xx * mpitch + yy * 4) = m;
    }
}
// the gradient records back to the E/G gradient rows; sums[h] += what this thread stored for head h of ITS piece (tid & 1: 0 = E, 1 = G;
// the stored, i.e. rounded, values)
template <typename T>
__device__ __forceinline__ void arm_store16(const ThirdArm& ta, void* d_eg, int b, int dir, int g, int N, const char* lds, int tid,
                                            float (&sums)[HG]) {
    using A = ArmStage<T, HG, 1>;
    const int pitch = A::pitch(dir);
    char* deg = reinterpret_cast<char*>(d_eg);
    static_assert(kThreads % 2 == 0, "a thread must own one piece (E or G)");
#pragma nounroll
    for (int idx = tid; idx < 32 * 32 * 2; idx += kThreads) {
        const int half = idx & 1, pr = idx >> 1, yy = pr & 31, xx = pr >> 5;
        if (xx < N && yy < N && (half == 0 ? ta.biased : ta.gated)) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(lds + xx * pitch + yy * A::kPairBytes + half * 16);
            const uint4 v = make_uint4(src[0], src[1], src[2], src[3]);
            *reinterpret_cast<uint4*>(deg + ((((int64_t)b * N + xx) * N + yy) * ta.ld + (half == 0 ? ta.e_off : ta.g_off) + g * HG) * 2) = v;
            T t[HG];
            __builtin_memcpy(t, &v, 16);
#pragma unroll
            for (int h = 0; h < HG; ++h) sums[h] += to_f32(t[h]);
        }
    }
}

// FL >= 0: BIASED / GATED compiled in (the training instantiation); FL < 0: read from the arguments
template <typename T, bool CS, int FL, bool DMA>
__global__ void __launch_bounds__(kThreads, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) tri_att_bwd2_kernel(const tgt_triplet_attention_args a) {
    using F = frag_t<T>;
    constexpr int kOffXch = Lay<DMA>::kOffXch, kOffPart = Lay<DMA>::kOffPart;
    using G = TriGeo<T, D, HG>;            // (third-arm staging only)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t sbase = (uint32_t)(uintptr_t)smem;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hi = lane >> 5;
    const int p16 = lane & 15, g16 = lane >> 4;
    const TriCtx c = tri_ctx<T, D, HG>(a, wave);
    const int N = c.N;
    const ThirdArm ta = tri_third_arm(a, c.dir);
    const bool biased = FL >= 0 ? (FL & TGT_TRI_BIASED) != 0 : ta.biased, gated = FL >= 0 ? (FL & TGT_TRI_GATED) != 0 : ta.gated;
    constexpr float kLog2e = 1.4426950408889634f;
    const float scale2 = a.scale * kLog2e;
#ifdef TGT_PROBES
    const int ablate = a._pad0;            // probe builds only (results are wrong): 1 no loads, 2 no stores, 4 no tile math
#else
    constexpr int ablate = 0;
#endif

    const int64_t Nl = N;
    const int64_t ldq = a.ld_dqkv[c.dir] ? a.ld_dqkv[c.dir] : a.ld_qkv[c.dir];
    const int64_t lde = a.ld_deg[c.dir] ? a.ld_deg[c.dir] : a.ld_eg[c.dir];
    const uint32_t hch = (uint32_t)(c.g * HG * D * 2);
    const uint32_t lds_ = (uint32_t)(a.ld_qkv[c.dir] * 2), ldg_ = (uint32_t)(ldq * 2), ldo_ = (uint32_t)(a.ld_out * 2);
    const __amdgpu_buffer_rsrc_t r_src = graph_rsrc(a.qkv[c.dir], Nl * Nl * a.ld_qkv[c.dir] * 2, c.b);
    const __amdgpu_buffer_rsrc_t r_grd = graph_rsrc(a.d_qkv[c.dir], Nl * Nl * ldq * 2, c.b);
    const __amdgpu_buffer_rsrc_t r_do = graph_rsrc(a.d_out, Nl * Nl * a.ld_out * 2, c.b);
    // the same graphs with zero records: every access is out of range (steps past the end of the walk)
    const __amdgpu_buffer_rsrc_t r_src0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.qkv[c.dir]), 0, 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_do0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.d_out), 0, 0, 0x00020000);
    const uint32_t qo = (uint32_t)(a.q_off[c.dir] * 2) + hch, ko = (uint32_t)(a.k_off[c.dir] * 2) + hch, vo = (uint32_t)(a.v_off[c.dir] * 2) + hch;
    const uint32_t oo = (uint32_t)(a.o_off[c.dir] * 2) + hch;
    // Q-type rows (i, j): row stride N*ld, j stride ld;  partner rows (j,k) inward / (k,j) outward
    const bool inward = c.dir == 0;
#ifdef TGT_PROBES
    const bool contig = (ablate & 8) != 0;       // probe: every slab = 32 CONSECUTIVE rows (wrong pairs; how much does the row stride cost?)
#else
    constexpr bool contig = false;
#endif
    const uint32_t sQr = contig ? lds_ : (uint32_t)N * lds_, sQj = contig ? (uint32_t)N * lds_ : lds_;
    const uint32_t sKr = (inward || contig) ? lds_ : (uint32_t)N * lds_, sKj = (inward || contig) ? (uint32_t)N * lds_ : lds_;
    const uint32_t gQr = contig ? ldg_ : (uint32_t)N * ldg_, gQj = contig ? (uint32_t)N * ldg_ : ldg_;
    const uint32_t gKr = (inward || contig) ? ldg_ : (uint32_t)N * ldg_, gKj = (inward || contig) ? (uint32_t)N * ldg_ : ldg_;
    const uint32_t oQr = contig ? ldo_ : (uint32_t)N * ldo_, oQj = contig ? (uint32_t)N * ldo_ : ldo_;
    ThirdArm dta = ta;
    dta.ld = lde;

    // this thread's 16-byte chunk of every slab: row tid / 16, PHYSICAL slot tid % 16 of the LDS image (lane-linear inside a
    // wave: what an LDS-DMA load writes), i.e. slot (tid % 16) ^ swz(row) of the row in memory; rows past N are out of range
    const int crow = tid >> 4, cslot = tid & 15;
    const uint32_t lslot16 = (uint32_t)((cslot ^ swz(crow)) & 15) * 16u;
    const uint32_t kOOB = 0x80000000u;
    const bool rowok = crow < N;
    const uint32_t vQ = rowok ? (uint32_t)crow * sQr + lslot16 : kOOB;
    const uint32_t vK = rowok ? (uint32_t)crow * sKr + lslot16 : kOOB;
    const uint32_t vO = rowok ? (uint32_t)crow * oQr + lslot16 : kOOB;
    const uint32_t wQ = rowok ? (uint32_t)crow * gQr + lslot16 : kOOB;
    const uint32_t wK = rowok ? (uint32_t)crow * gKr + lslot16 : kOOB;
    const int chunk = crow * kRow + cslot * 16;

    float* part = reinterpret_cast<float*>(smem + kOffXch);        // after the walk: [thread][head] partial dE / dG column sums (16 KB of the dead exchange tiles)
    // (s_setprio 1 for the younger half of the workgroup: measured neutral, profiles/r05o_ab_prio.txt -- the kernel is memory-bound)

    // a graph DropPath dropped (graph_scale[b] == 0) receives an all-zero d_out: zeros to its gradient rows and column sums
    const bool dead = a.graph_scale && a.graph_scale[c.b] == 0.f;          // workgroup-uniform
    if (dead) {
        const u32x4_t z = {0, 0, 0, 0};
        for (int j = 0; j < N; ++j) {
            __builtin_amdgcn_raw_buffer_store_b128(z, r_grd, (int)wQ, (int)(qo + (uint32_t)j * gQj), TGT_ST_AUX);
            __builtin_amdgcn_raw_buffer_store_b128(z, r_grd, (int)wK, (int)(ko + (uint32_t)j * gKj), TGT_ST_AUX);
            __builtin_amdgcn_raw_buffer_store_b128(z, r_grd, (int)wK, (int)(vo + (uint32_t)j * gKj), TGT_ST_AUX);
        }
        const float zero16[16] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        arm_stage_put_grad<T, HG, 1>(smem, c.dir, wave, r, hi, 0, zero16, zero16);
        __syncthreads();
        float unused[HG] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        arm_store16<T>(dta, a.d_eg[c.dir], c.b, c.dir, c.g, N, smem, tid, unused);
        if constexpr (CS) {
            float* row = a.d_qkv_colsum[c.dir] + (int64_t)c.b * ldq + c.g * HG * D;
            if (tid < HG * D) {
                row[a.q_off[c.dir] + tid] = 0.f;
                row[a.k_off[c.dir] + tid] = 0.f;
                row[a.v_off[c.dir] + tid] = 0.f;
            }
            if (tid < 2 * HG && (biased || gated)) {
                float* erow = a.d_eg_colsum[c.dir] + (int64_t)c.b * lde;
                if (tid < HG) { if (biased) erow[a.e_off[c.dir] + c.g * HG + tid] = 0.f; }
                else if (gated) erow[a.g_off[c.dir] + c.g * HG + tid - HG] = 0.f;
            }
        }
        return;
    }

    // third-arm tile of this head (accumulator layout), staged through LDS once (aliases the slab sets)
    f32x2 biasM[8], gate[8], dE[8], dG[8];
    arm_load16<T>(ta, c.b, c.dir, c.g, N, smem, tid);
    __syncthreads();
    {
        float b16[16], g16[16];
        arm_stage_read<T, HG, 1, true>(ta, smem, c.dir, wave, N, r, hi, 0, 0, b16, g16);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            // log2 domain; a masked entry (finfo.min) times log2(e) would overflow to -inf: clamp it back (see the old kernel)
            const float bl = b16[q] * kLog2e;
            b16[q] = (bl == -INFINITY && b16[q] != -INFINITY) ? -3.402823466e38f : bl;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            biasM[k] = f32x2{b16[2 * k], b16[2 * k + 1]};
            gate[k] = f32x2{g16[2 * k], g16[2 * k + 1]};
            dE[k] = dG[k] = f32x2{0.f, 0.f};
        }
    }
    __syncthreads();

    // per-lane LDS addresses (bytes from the start of a set / of this wave's exchange tiles), all loop-invariant
    const int hb = wave * 32;                                          // this head's first byte in a slab row
    const uint32_t a_frag = (uint32_t)slab_off(r, hb + 16 * hi);       // operand rows: 8 channels of row r
    // K^T for dQ (32x32x16 A operand, lane = (d = lane & 15, hi); lanes with bit 4 set mirror their partner: rows >= 16 of the result are unused).
    // chunk c' holds the keys of accumulator register groups m = c' and m = c' + 2: k = 8c' + 4hi + t and 8c' + 16 + 4hi + t
    const int c2 = p16 & 3, pr = p16 >> 2;
    const uint32_t a_kt0 = (uint32_t)slab_off(4 * hi + pr, hb + 8 * c2);
    const uint32_t a_kt1 = (uint32_t)slab_off(8 + 4 * hi + pr, hb + 8 * c2);
    // Q^T / dO^T for dK / dV (16x16x32 A operand, lane = (d = lane & 15, g = lane >> 4)): rows i = 16u + 4g + t
    const uint32_t a_qt = (uint32_t)slab_off(4 * g16 + pr, hb + 8 * c2);
    // exchange tile [kt][i][quad ^ ((i >> 2) & 3)][4]: writes in accumulator layout (lane = (i = r, hi), register group m)
    const uint32_t xw = (uint32_t)(kOffXch + wave * kXch + r * 32 + ((hi ^ ((r >> 2) & 3)) << 3));     // m even; m odd: ^ 16
    // ... transposed reads as the 16x16x32 B operand (lane = (k = lane & 15, g)): rows i = 16u + 4g + t of key tile kt
    const uint32_t xr = (uint32_t)(kOffXch + wave * kXch + (4 * g16 + pr) * 32 + ((c2 ^ g16) << 3));
    // result rows: dQ^T from the 32x32 accumulator (lane = (i = r, hi): d = 4hi + t and 8 + 4hi + t), dK^T / dV^T from the
    // 16x16 accumulators (lane = (k = lane & 15, g): d = 4g + t)
    const uint32_t w_q = (uint32_t)slab_off(r, hb + 8 * hi);           // second group: ^ 16
    const uint32_t w_kv = (uint32_t)slab_off(p16, hb + 8 * g16);       // key tile 1: + 16 rows

    // column sums of dQ / dK / dV over i (k) and j, in accumulator layout
    f32x2 csq[4], csk[2], csv[2];
    if constexpr (CS) {
#pragma unroll
        for (int k = 0; k < 4; ++k) csq[k] = f32x2{0.f, 0.f};
        csk[0] = csk[1] = csv[0] = csv[1] = f32x2{0.f, 0.f};
    }

    auto issue = [&](u32x4_t (&pre)[4], int jj) {
        const bool live = jj < N && !(ablate & 1);                      // (scalar: past the end every load is out of range)
        const __amdgpu_buffer_rsrc_t rs = live ? r_src : r_src0, ro = live ? r_do : r_do0;
        pre[0] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)vQ, (int)(qo + (uint32_t)jj * sQj), TGT_LD_AUX);
        pre[1] = __builtin_amdgcn_raw_buffer_load_b128(ro, (int)vO, (int)(oo + (uint32_t)jj * oQj), TGT_LD_AUX);
        pre[2] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)vK, (int)(ko + (uint32_t)jj * sKj), TGT_LD_AUX);
        pre[3] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)vK, (int)(vo + (uint32_t)jj * sKj), TGT_LD_AUX);
    };
    auto commit = [&](const u32x4_t (&pre)[4], int set) {
        char* s = smem + set * kSet + chunk;
#pragma unroll
        for (int t = 0; t < 4; ++t) *reinterpret_cast<u32x4_t*>(s + t * kSlab) = pre[t];
    };
    // DMA: the four slabs of step jj straight into set `set` (each wave: its 4 rows x 256 B of every slab, lane-linear).
    // Inline assembly on purpose: hipcc orders every later LDS read behind a builtin LDS-DMA load with s_waitcnt vmcnt(0)
    // (it cannot tell the sets apart), which would serialise the walk; the waits are placed by hand at the barrier instead.
    // (s_nop: one wait state between a SALU write of M0 and the LDS-DMA that uses it.)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const uint64_t src_base = (uint64_t)(uintptr_t)a.qkv[c.dir] + (uint64_t)c.b * (uint64_t)(Nl * Nl * a.ld_qkv[c.dir] * 2);
    const uint64_t do_base = (uint64_t)(uintptr_t)a.d_out + (uint64_t)c.b * (uint64_t)(Nl * Nl * a.ld_out * 2);
    const uint32_t src_bytes = (uint32_t)(Nl * Nl * a.ld_qkv[c.dir] * 2), do_bytes = (uint32_t)(Nl * Nl * a.ld_out * 2);
    auto dma = [&](int jj, int set) {
        const bool live = jj < N && !(ablate & 1);                      // (scalar: past the end every load is out of range)
        const u32x4_t rs = {(uint32_t)src_base, (uint32_t)(src_base >> 32) & 0xffffu, live ? src_bytes : 0u, 0x00020000u};
        const u32x4_t ro = {(uint32_t)do_base, (uint32_t)(do_base >> 32) & 0xffffu, live ? do_bytes : 0u, 0x00020000u};
        const uint32_t l0 = sbase + (uint32_t)(set * kSet + wave_u * 1024);
#define TGT_DMA16(LDSADDR, VOFF, RSRC, SOFF)                                                                                   \
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen" TGT_BWD2_LD_POLICY " lds"                                  \
                     :: "s"(LDSADDR), "v"(VOFF), "s"(RSRC), "s"(SOFF) : "memory")      /* (M0 is not live across statements in this kernel: no other LDS-DMA, movrel or GDS use) */
        TGT_DMA16(l0, vQ, rs, qo + (uint32_t)jj * sQj);
        TGT_DMA16(l0 + kSlab, vO, ro, oo + (uint32_t)jj * oQj);
        TGT_DMA16(l0 + 2 * kSlab, vK, rs, ko + (uint32_t)jj * sKj);
        TGT_DMA16(l0 + 3 * kSlab, vK, rs, vo + (uint32_t)jj * sKj);
#undef TGT_DMA16
    };
    auto dummy_stores = [&]() {
        // three dropped (out-of-range) stores: the loop is entered with the queue it has on its back edge -- 4 loads, then 3
        // stores -- so the waits of the walk are exact counts on both paths
        const u32x4_t z = {0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 3; ++t) __builtin_amdgcn_raw_buffer_store_b128(z, r_grd, (int)kOOB, 16 * t, TGT_ST_AUX);     // (distinct, or hipcc folds them)
    };

    u32x4_t pre[4];
    if constexpr (DMA) {
        // rows past N are never written by the loads (out of range) and stay zero through the walk (their results are zeros)
        if (N < 32) {
            const u32x4_t z = {0, 0, 0, 0};
            for (int o = tid * 16; o < Lay<true>::kSets * kSet; o += kThreads * 16) *reinterpret_cast<u32x4_t*>(smem + o) = z;
            __syncthreads();
        }
        // the loop is entered with the queue it has on its back edge: {4 loads, 3 stores} per step in flight
        dma(0, 0);
#pragma unroll
        for (int d = 1; d < Lay<true>::kDepth; ++d) {
            dma(d, d);
            dummy_stores();
        }
        // queue: loads(0) x4, then {loads(d) x4, 3 stores} for d = 1 .. DEPTH-1
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(7 * (Lay<true>::kDepth - 1)) : "memory");
    } else {
        issue(pre, 0);
        commit(pre, 0);
        issue(pre, 1);
        dummy_stores();
        __syncthreads();
    }

    int cur = 0;
    for (int j = 0; j < N; ++j) {
        if constexpr (DMA) {
            // set of step j + 2 = set of step j - 1: every wave finished reading it before the last barrier, and this thread's
            // loads write the chunks this thread itself read for the stores of step j - 1
            dma(j + Lay<true>::kDepth, cur == 0 ? Lay<true>::kSets - 1 : cur - 1);
        } else {
            // Hazards (one barrier per j, as in the old kernel): set cur^1 holds the results of step j-1; this thread read its
            // own chunk of them for the stores at the end of step j-1 and now overwrites that same chunk.
            commit(pre, cur ^ 1);
            issue(pre, j + 2);
        }

        const uint32_t sQ = sbase + (uint32_t)(cur * kSet), sO = sQ + kSlab, sK = sQ + 2 * kSlab, sV = sQ + 3 * kSlab;
        char* const gQs = smem + cur * kSet;
        if (!(ablate & 4)) {
        const F fq = load_frag<T>(reinterpret_cast<const T*>(gQs + a_frag));
        const F fo = load_frag<T>(reinterpret_cast<const T*>(gQs + kSlab + a_frag));
        const F fk = load_frag<T>(reinterpret_cast<const T*>(gQs + 2 * kSlab + a_frag));
        const F fv = load_frag<T>(reinterpret_cast<const T*>(gQs + 3 * kSlab + a_frag));
        // transposed operands (16 rows further: + 4096 bytes, same swizzle)
        const F kt0 = frag_of<T>(tr_read(sK + a_kt0), tr_read(sK + a_kt0 + 4096));
        const F kt1 = frag_of<T>(tr_read(sK + a_kt1), tr_read(sK + a_kt1 + 4096));
        const F qt = frag_of<T>(tr_read(sQ + a_qt), tr_read(sQ + a_qt + 4096));
        const F ot = frag_of<T>(tr_read(sO + a_qt), tr_read(sO + a_qt + 4096));

        f32x2 s[8], da[8];
        {
            const f32x16 zero = {0};
            const f32x16 z0 = mma32(fk, fq, zero);         // S^T[k][i]
            const f32x16 z1 = mma32(fv, fo, zero);         // dA^T[k][i]
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                s[k] = f32x2{z0[2 * k], z0[2 * k + 1]};
                da[k] = f32x2{z1[2 * k], z1[2 * k + 1]};
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            s[k] = s[k] * scale2 + biasM[k];
            mx = fmaxf(mx, fmaxf(s[k].x, s[k].y));
        }
        mx = fmaxf(mx, xhalf(mx));
        if (mx == -INFINITY) mx = 0.f;           // padding column: every weight is exactly 0
        f32x2 sum2 = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const f32x2 t = s[k] - mx;
            s[k] = f32x2{fast_exp2(t.x), fast_exp2(t.y)};
            sum2 += s[k];
        }
        float sum = sum2.x + sum2.y;
        sum += xhalf(sum);
        const float inv = sum > 0.f ? fast_rcp(sum) : 0.f;
        f32x2 delta2 = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const f32x2 p = s[k] * inv;
            const f32x2 dp = da[k] * gate[k];
            delta2 += p * dp;
            if (gated) dG[k] += da[k] * p;
            s[k] = p;
            da[k] = dp;
        }
        float delta = delta2.x + delta2.y;
        delta += xhalf(delta);

        // dS (times the logit scale) and A in accumulator layout -> 16-bit operand fragments.  Register group m (registers 4m ..
        // 4m + 3 = 4 consecutive keys 8m + 4hi + t); chunk c' of the dQ contraction = groups c' and c' + 2.
        F dsf[2], atf[2];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const f32x2 ds0 = s[2 * m] * (da[2 * m] - delta), ds1 = s[2 * m + 1] * (da[2 * m + 1] - delta);
            if (biased) { dE[2 * m] += ds0; dE[2 * m + 1] += ds1; }
            const f32x2 at0 = s[2 * m] * gate[2 * m], at1 = s[2 * m + 1] * gate[2 * m + 1];
            const f32x2 e0 = ds0 * a.scale, e1 = ds1 * a.scale;
            const int cc = m & 1, o = 4 * (m >> 1);
            dsf[cc][o] = from_f32<T>(e0.x); dsf[cc][o + 1] = from_f32<T>(e0.y); dsf[cc][o + 2] = from_f32<T>(e1.x); dsf[cc][o + 3] = from_f32<T>(e1.y);
            atf[cc][o] = from_f32<T>(at0.x); atf[cc][o + 1] = from_f32<T>(at0.y); atf[cc][o + 2] = from_f32<T>(at1.x); atf[cc][o + 3] = from_f32<T>(at1.y);
        }
        // exchange tile: group m -> key tile m >> 1, quad 2(m & 1) + hi
        {
            char* x0 = smem + xw;
            char* x1 = smem + (xw ^ 16u);
            *reinterpret_cast<uint2*>(x0) = frag_half<T>(dsf[0], 0);
            *reinterpret_cast<uint2*>(x1) = frag_half<T>(dsf[1], 0);
            *reinterpret_cast<uint2*>(x0 + 1024) = frag_half<T>(dsf[0], 1);
            *reinterpret_cast<uint2*>(x1 + 1024) = frag_half<T>(dsf[1], 1);
            *reinterpret_cast<uint2*>(x0 + 2048) = frag_half<T>(atf[0], 0);
            *reinterpret_cast<uint2*>(x1 + 2048) = frag_half<T>(atf[1], 0);
            *reinterpret_cast<uint2*>(x0 + 3072) = frag_half<T>(atf[0], 1);
            *reinterpret_cast<uint2*>(x1 + 3072) = frag_half<T>(atf[1], 1);
        }
        // dQ^T[d][i] = sum_k K^T[d][k] dS^T[k][i] from the registers (runs while the exchange tile settles)
        f32x16 dq = {0};
        dq = mma32(kt0, dsf[0], dq);
        dq = mma32(kt1, dsf[1], dq);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // dK^T[d][k] = sum_i Q^T[d][i] dS[i][k],  dV^T[d][k] = sum_i dO^T[d][i] A[i][k]   (one 16x16x32 per key tile)
        f32x4 dk[2], dv[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const uint32_t x = sbase + xr + (uint32_t)(kt * 1024);
            const F bds = frag_of<T>(tr_read(x), tr_read(x + 512));
            const F bat = frag_of<T>(tr_read(x + 2048), tr_read(x + 2048 + 512));
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            dk[kt] = mma16(qt, bds, z);
            dv[kt] = mma16(ot, bat, z);
        }
        // results into the slabs of this set (this head's columns: no other wave touches them), as 16-bit rows
        *reinterpret_cast<uint2*>(gQs + w_q) = pack4<T>(dq[0], dq[1], dq[2], dq[3]);
        *reinterpret_cast<uint2*>(gQs + (w_q ^ 16u)) = pack4<T>(dq[4], dq[5], dq[6], dq[7]);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            *reinterpret_cast<uint2*>(gQs + 2 * kSlab + w_kv + kt * 4096) = pack4<T>(dk[kt][0], dk[kt][1], dk[kt][2], dk[kt][3]);
            *reinterpret_cast<uint2*>(gQs + 3 * kSlab + w_kv + kt * 4096) = pack4<T>(dv[kt][0], dv[kt][1], dv[kt][2], dv[kt][3]);
        }
        if constexpr (CS) {
#pragma unroll
            for (int k = 0; k < 4; ++k) csq[k] += f32x2{dq[2 * k], dq[2 * k + 1]};
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                csk[k] += f32x2{dk[0][2 * k], dk[0][2 * k + 1]} + f32x2{dk[1][2 * k], dk[1][2 * k + 1]};
                csv[k] += f32x2{dv[0][2 * k], dv[0][2 * k + 1]} + f32x2{dv[1][2 * k], dv[1][2 * k + 1]};
            }
        }
        }
        if constexpr (DMA) {
            // queue: loads(j+1) x4, stores(j-1) x3, loads(j+2) x4 -- the slabs of step j + 1 have landed for every wave behind this
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(7 * (Lay<true>::kDepth - 1)) : "memory");
        } else {
            __syncthreads();
        }
        if (!(ablate & 2)) {
            const char* sres = smem + cur * kSet + chunk;
            const u32x4_t o0 = *reinterpret_cast<const u32x4_t*>(sres);
            const u32x4_t o2 = *reinterpret_cast<const u32x4_t*>(sres + 2 * kSlab);
            const u32x4_t o3 = *reinterpret_cast<const u32x4_t*>(sres + 3 * kSlab);
            __builtin_amdgcn_raw_buffer_store_b128(o0, r_grd, (int)wQ, (int)(qo + (uint32_t)j * gQj), TGT_ST_AUX);
            __builtin_amdgcn_raw_buffer_store_b128(o2, r_grd, (int)wK, (int)(ko + (uint32_t)j * gKj), TGT_ST_AUX);
            __builtin_amdgcn_raw_buffer_store_b128(o3, r_grd, (int)wK, (int)(vo + (uint32_t)j * gKj), TGT_ST_AUX);
        } else if constexpr (DMA) {
            dummy_stores();             // (probe builds: keep the queue shape the wait counts assume)
        }
        if constexpr (DMA) cur = cur == Lay<true>::kSets - 1 ? 0 : cur + 1; else cur ^= 1;
    }
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the out-of-range loads past the end of the walk)
    __syncthreads();
    // third-arm gradients (summed over j in registers) leave through LDS
    {
        float e16[16], g16v[16];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            f32x2 gg = dG[k];
            if (gated) gg *= gate[k] * (1.f - gate[k]);      // d sigmoid, once per tile
            e16[2 * k] = dE[k].x; e16[2 * k + 1] = dE[k].y;
            g16v[2 * k] = gg.x; g16v[2 * k + 1] = gg.y;
        }
        arm_stage_put_grad<T, HG, 1>(smem, c.dir, wave, r, hi, 0, e16, g16v);
    }
    __syncthreads();
    {
        float pv[HG] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        arm_store16<T>(dta, a.d_eg[c.dir], c.b, c.dir, c.g, N, smem, tid, pv);
        if constexpr (CS) {
            // (the exchange tiles are dead: their space takes one partial per thread and head)
#pragma unroll
            for (int h = 0; h < HG; ++h) part[tid * HG + h] = pv[h];
        }
    }
    if constexpr (CS) {
        float* row = a.d_qkv_colsum[c.dir] + (int64_t)c.b * ldq + c.g * HG * D + wave * D;
        // dQ: lane (i = r, hi), register q: d = (q & 3) + 8 (q >> 2) + 4 hi -> sum over the 32 lanes of a half
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float v = group_sum<32>(q & 1 ? csq[q >> 1].y : csq[q >> 1].x);
            if (r == 0) row[a.q_off[c.dir] + (q & 3) + 8 * (q >> 2) + 4 * hi] = v;
        }
        // dK / dV: lane (k = lane & 15, g), register t: d = 4g + t -> sum over the 16 lanes of a row
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float vk = group_sum<16>(t & 1 ? csk[t >> 1].y : csk[t >> 1].x);
            const float vv = group_sum<16>(t & 1 ? csv[t >> 1].y : csv[t >> 1].x);
            if (p16 == 0) {
                row[a.k_off[c.dir] + 4 * g16 + t] = vk;
                row[a.v_off[c.dir] + 4 * g16 + t] = vv;
            }
        }
        __syncthreads();
        if (biased || gated) {
            constexpr int kVals = 2 * HG;                  // E of the 8 heads, then G
            if (tid < kVals) {
                float v = 0.f;
                for (int t = tid / HG; t < kThreads; t += 2) v += part[t * HG + tid % HG];        // threads of this piece (t & 1), fixed order
                float* erow = a.d_eg_colsum[c.dir] + (int64_t)c.b * lde;
                if (tid < HG) { if (biased) erow[a.e_off[c.dir] + c.g * HG + tid] = v; }
                else if (gated) erow[a.g_off[c.dir] + c.g * HG + tid - HG] = v;
            }
        }
    }
}

template <typename T, bool CS, int FL>
static int launch_one(const tgt_triplet_attention_args& a_in, hipStream_t st) {
#ifdef TGT_PROBES
    tgt_triplet_attention_args a = a_in;
    a._pad0 = getenv("TGT_TRI_BWD_ABLATE") ? atoi(getenv("TGT_TRI_BWD_ABLATE")) : 0;
#else
    const tgt_triplet_attention_args& a = a_in;
#endif
    static const bool dma = !(getenv("TGT_TRI_BWD2_DMA") && atoi(getenv("TGT_TRI_BWD2_DMA")) == 0);       // A/B knob
    if (dma) {
        static bool attr_set[16] = {};
        constexpr int kLds = Lay<true>::kLds;
        if (!dyn_lds_once(attr_set, reinterpret_cast<const void*>(&tri_att_bwd2_kernel<T, CS, FL, true>), kLds))
            return set_error(TGT_ERR_LAUNCH, "tri_att_bwd2_kernel: cannot reserve %d bytes of LDS", kLds);
        hipLaunchKernelGGL((tri_att_bwd2_kernel<T, CS, FL, true>), dim3(a.B * 2 * (a.H / HG)), dim3(kThreads), kLds, st, a);
    } else {
        static bool attr_set[16] = {};
        constexpr int kLds = Lay<false>::kLds;
        if (!dyn_lds_once(attr_set, reinterpret_cast<const void*>(&tri_att_bwd2_kernel<T, CS, FL, false>), kLds))
            return set_error(TGT_ERR_LAUNCH, "tri_att_bwd2_kernel: cannot reserve %d bytes of LDS", kLds);
        hipLaunchKernelGGL((tri_att_bwd2_kernel<T, CS, FL, false>), dim3(a.B * 2 * (a.H / HG)), dim3(kThreads), kLds, st, a);
    }
    return check_launch("tri_att_bwd2_kernel");
}
template <typename T>
static int launch(const tgt_triplet_attention_args& a, hipStream_t st) {
    constexpr int kBG = TGT_TRI_BIASED | TGT_TRI_GATED;
    const bool cs = a.d_qkv_colsum[0] != nullptr;
    if (cs && (a.flags & kBG) == kBG) return launch_one<T, true, kBG>(a, st);
    if (cs) return launch_one<T, true, -1>(a, st);
    return launch_one<T, false, -1>(a, st);
}

}  // namespace bwd2

bool tri_att_bwd2_eligible(const tgt_triplet_attention_args& a) {
    static const bool off = getenv("TGT_TRI_BWD2") && atoi(getenv("TGT_TRI_BWD2")) == 0;      // A/B knob: 0 = the round-1..3 kernel
    if (off || !((a.dtype == TGT_BF16 || a.dtype == TGT_F16) && a.D == 16 && a.N <= 32 && a.H % 8 == 0 && !(a.dropout_p > 0.f))) return false;
    // the third-arm tiles travel as 16-byte pieces (arm_load16 / arm_store16): row lengths and column offsets in whole pieces
    if (a.flags & (TGT_TRI_BIASED | TGT_TRI_GATED))
        for (int dir = 0; dir < 2; ++dir) {
            const int64_t lde = a.ld_deg[dir] ? a.ld_deg[dir] : a.ld_eg[dir];
            if (a.ld_eg[dir] % 8 || lde % 8 || a.e_off[dir] % 8 || a.g_off[dir] % 8 || ((uintptr_t)a.eg[dir] | (uintptr_t)a.d_eg[dir]) % 16)
                return false;
        }
    return true;
}
int tri_att_bwd2_run(const tgt_triplet_attention_args& a, hipStream_t st) {
    return a.dtype == TGT_BF16 ? bwd2::launch<bf16_t>(a, st) : bwd2::launch<f16_t>(a, st);
}

}  // namespace tgt
