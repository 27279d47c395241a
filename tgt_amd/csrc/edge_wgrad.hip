// Data gradient AND weight gradient of a 256 -> 256 edge Linear in ONE pass over the rows -- gfx950 (round 6).
//
// The backward of the edge FFN (reference lib/tgt/layers/layers.py:155-160 under autograd, wired at :284-290) is, per Linear,
//     dX = dY W            (data gradient, with the activation's / LayerNorm's backward as its epilogue: edge_rows_kernel)
//     dW = dY^T X          (weight gradient: a library batched GEMM over row chunks + a closing sum)
// and the second product re-reads dY and X (2 x 134 MB at the BASELINE shape) only to contract them over the rows the first
// kernel already had on chip.  Here the row kernel keeps dY's tile, RECOMPUTES X's tile from the epilogue operand it loads anyway
//     TGT_EPI_GELU_BWD   X = dropout(gelu(pre)) * sample_scale      (bit-identical to what lin_W1's launch stored)
//     TGT_EPI_LN_BWD     X = LayerNorm(s; gamma, beta)               (from s, mean, rstd)
// and accumulates dW = sum_tiles dY_tile^T X_tile on the matrix cores next to the data gradient: X and dY are not read again.
//
// What that costs on a CU (VERDICT r5 item 1 asked for exactly this form, DESIGN r5 section 8 item 1b had rejected it by arithmetic):
//   * dW is 256 x 256 fp32 = 256 KB per persistent workgroup: 8 waves x 128 accumulator registers.  So the workgroup is
//     8 waves x 256 registers (2 per SIMD) instead of 16 x 128, there are no separate wave roles -- every wave walks the phases
//         P1  data gradient of its 32 output columns (k-loop over LDS) -> staging tile S, + its dY^T fragments (transposed reads)
//         P2  row phase on S (thread = row x 16-byte chunk: epilogue, result row to HBM) and X written back into S in place
//         P3  dW[32 rows of this wave][256] += dY^T X   (X^T fragments by ds_read_b64_tr_b16 from S)
//     with three barriers per 32-row tile -- and W^T (128 KB) lives in LDS, not in registers:
//     LDS = W^T 128 KB | dY tile 16 KB | S 16 KB = all 160 KB.
//   * Both tiles are read along rows (ds_read_b128, the k-loop) AND along columns (ds_read_b64_tr_b16), so the 16-byte slot swizzle
//     is a 4-bit permutation of the row bits chosen to be conflict-free for: the k-loop's 16-lane groups, the 4-rows x 64-byte
//     blocks of a transposed read, the 8-lane groups of the accumulator's ds_write_b128 (wg_f below).
//   * Every workgroup ends with a 256 KB partial: 64 MB per launch at 256 workgroups (written here, read once more by the closing
//     tgt_sum_planes).  That is 1 E of traffic against the 2 E of re-reads it replaces -- the fusion's ceiling is ~1 E per Linear.
#include <cstdlib>
#include "common.hpp"
#include "edge_common.hpp"

namespace tgt {

int edge_linear_parts(int64_t M, int N);

namespace wg {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int K = 256, N = 256, KS = 16, kBM = 32, kPass = 2, kThreads = 512;
constexpr int kRow = 512;                                   // bytes of a tile row (256 16-bit values)
constexpr int kWBytes = N * kRow, kTile = kBM * kRow;
constexpr int kOffA = kWBytes, kOffS = kOffA + kTile, kLds = kOffS + kTile;      // 163840 = the whole LDS of a CU
constexpr uint32_t kNone = 0xffffffffu;

// 16-byte slot swizzle: slot ^ wg_f(row), wg_f a permutation of the low four row bits (r3 r2 r1 r0) -> (r1^r3, r0, r1, r2):
//   * bijective on row & 15: the 16 rows of a ds_read_b128 lane group ({0-3,12-15,20-27} ...) hit 16 different slots;
//   * rows R .. R+3 (R % 4 == 0) differ in bits 3:2: the four 64-byte row segments of a transposed read's 32-lane half fall into
//     the four 64-byte bank quarters;
//   * rows R .. R+7 (R % 8 == 0) differ in bits 2:0: the eight lanes of a ds_write_b128 group (accumulator rows r .. r+7, one slot)
//     cover a whole 128-byte bank window.
__device__ __forceinline__ int wg_f(int row) {
    return ((((row >> 1) ^ (row >> 3)) & 1) << 3) | ((row & 1) << 2) | (row & 2) | ((row >> 2) & 1);
}
__device__ __forceinline__ int wg_off(int row, int slot) { return row * kRow + ((slot ^ wg_f(row)) << 4); }
// byte column cb (a multiple of 8) of a row
__device__ __forceinline__ int wg_offb(int row, int cb) { return row * kRow + ((((cb >> 4) ^ wg_f(row)) & 31) << 4) + (cb & 15); }

__device__ __forceinline__ s16x4 tr_read(uint32_t addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_s16x4*>(addr));
}
typedef __attribute__((address_space(3))) u32x4_t lds_u32x4;
template <typename T>
__device__ __forceinline__ frag_t<T> lds_frag(uint32_t addr) {
    const u32x4_t raw = *reinterpret_cast<lds_u32x4*>(addr);
    frag_t<T> f;
    __builtin_memcpy(&f, &raw, 16);
    return f;
}
template <typename T>
__device__ __forceinline__ frag_t<T> frag_of(s16x4 lo, s16x4 hi) {
    frag_t<T> f;
    __builtin_memcpy(&f, &lo, 8);
    __builtin_memcpy(reinterpret_cast<char*>(&f) + 8, &hi, 8);
    return f;
}

template <typename T, int EPI>
__global__ void __launch_bounds__(kThreads, 2) edge_rows_wgrad_kernel(const tgt_edge_linear_args a, const uint64_t* seed_ctr) {
    using F = frag_t<T>;
    static_assert(EPI == EPI_GELU_BWD || EPI == EPI_LN_BWD, "backward epilogues only");
    constexpr bool kOp2 = EPI == EPI_LN_BWD;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hi = lane >> 5;
    const int p16 = lane & 15, r16 = (lane >> 4) & 1;
    const int64_t row_tiles = (a.M + kBM - 1) / kBM;
    if (blockIdx.x >= row_tiles) return;                   // (host: grid <= row_tiles, so every launched workgroup writes its dW plane)
    const int n_tiles = (int)((row_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x);
    auto tile_of = [&](int s) { return (int64_t)blockIdx.x + (int64_t)s * gridDim.x; };
    char* const sA = smem + kOffA;
    char* const sS = smem + kOffS;
    const int n0 = wave * 32;                              // this wave's 32 output columns (P1) = its 32 rows of dW (P3)

    // ---- W^T (256 x 256) -> LDS, once
    {
        const char* W = reinterpret_cast<const char*>(a.w);
        const int64_t ldw_b = a.ldw * 2;
        uint4 wv[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int pc = q * kThreads + tid;
            wv[q] = *reinterpret_cast<const uint4*>(W + (int64_t)(pc >> 5) * ldw_b + (pc & 31) * 16);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int pc = q * kThreads + tid;
            *reinterpret_cast<uint4*>(smem + wg_off(pc >> 5, pc & 31)) = wv[q];
        }
    }

    // ---- row-phase constants of this thread: (row rsub [+16], 16-byte chunk ch)
    const int ch = tid & 31, rsub = tid >> 5;
    const uint32_t thresh = a.dropout_p <= 0.f ? 0u : (uint32_t)fminf(65535.f, fmaxf(1.f, rintf(a.dropout_p * 65536.f)));
    const float inv_keep = a.dropout_p <= 0.f ? 1.f : 1.f / (1.f - a.dropout_p);
    constexpr int kCs = EPI == EPI_LN_BWD ? 4 : 1;
    f32x2 cs_g[kCs], cs_b[kCs], cs_x[4];
#pragma unroll
    for (int j = 0; j < kCs; ++j) cs_g[j] = cs_b[j] = rp_splat(0.f);
#pragma unroll
    for (int j = 0; j < 4; ++j) cs_x[j] = rp_splat(0.f);
    const float* scale_ptr = EPI == EPI_GELU_BWD ? a.out_scale : a.row_scale;
    const bool has_scale = scale_ptr != nullptr;
    const FastDiv per_sample((uint32_t)(has_scale ? a.rows_per_sample : 1));
    const int64_t n_samples = has_scale ? (a.M + a.rows_per_sample - 1) / a.rows_per_sample : 0;
    const __amdgpu_buffer_rsrc_t rs_scale = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(scale_ptr), 0, (int)(n_samples * 4), 0x00020000);
    const int64_t lda_b = a.lda * 2, ldr_b = a.ldr * 2, ldd_b = a.ld_ds * 2, ldo_b = a.ldo * 2, ldo2_b = a.ldo2 * 2;
    const uint32_t c16 = (uint32_t)ch * 16u;
    const uint32_t o_a = (uint32_t)rsub * (uint32_t)lda_b + c16;
    const uint32_t o_res = (uint32_t)rsub * (uint32_t)ldr_b + c16, o_ds = (uint32_t)rsub * (uint32_t)ldd_b + c16;
    const uint32_t o_out = (uint32_t)rsub * (uint32_t)ldo_b + c16, o_out2 = (uint32_t)rsub * (uint32_t)ldo2_b + c16;
    const uint32_t o_stat = (uint32_t)rsub * 4u;
    f32x2 gam[4], bet[4];                                  // LN_BWD: this thread's 8 columns of gamma / beta (beta: X = LayerNorm(s) needs it)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        gam[k] = rp_splat(1.f);
        bet[k] = rp_splat(0.f);
        if constexpr (kOp2) {
            if (a.gamma) gam[k] = *reinterpret_cast<const f32x2*>(a.gamma + ch * 8 + 2 * k);
            if (a.beta) bet[k] = *reinterpret_cast<const f32x2*>(a.beta + ch * 8 + 2 * k);
        }
    }

    // the dY tile (this thread's two 16-byte pieces: rows rsub and rsub + 16, chunk ch) and the row phase's operands, from HBM
    struct Ops { uint4 o1[kPass]; uint4 o2[kOp2 ? kPass : 1]; float mu[kOp2 ? kPass : 1], rs[kOp2 ? kPass : 1], sc[kPass]; };
    uint4 preA[kPass];
    auto fetch_a = [&](int64_t tile) {
        const __amdgpu_buffer_rsrc_t rsa = tile_rsrc(a.a, lda_b, N * 2, tile * kBM, a.M);
#pragma unroll
        for (int i = 0; i < kPass; ++i) preA[i] = rp_ld16(rsa, o_a + (uint32_t)i * 16u * (uint32_t)lda_b);
    };
    auto commit_a = [&]() {
#pragma unroll
        for (int i = 0; i < kPass; ++i) *reinterpret_cast<uint4*>(sA + wg_off(i * 16 + rsub, ch)) = preA[i];
    };
    auto fetch_ops = [&](int64_t tile, Ops& o) {
        const int64_t r0 = tile * kBM;
        const __amdgpu_buffer_rsrc_t rs1 = tile_rsrc(a.res, ldr_b, N * 2, r0, a.M);
        const __amdgpu_buffer_rsrc_t rs2 = tile_rsrc(a.ds_in, ldd_b, N * 2, r0, kOp2 ? a.M : 0);
        const __amdgpu_buffer_rsrc_t rsm = tile_rsrc(a.mean, 4, 4, r0, kOp2 ? a.M : 0), rsr = tile_rsrc(a.rstd, 4, 4, r0, kOp2 ? a.M : 0);
#pragma unroll
        for (int i = 0; i < kPass; ++i) {
            o.o1[i] = rp_ld16(rs1, o_res + (uint32_t)i * 16u * (uint32_t)ldr_b);
            if constexpr (kOp2) {
                o.o2[i] = rp_ld16(rs2, o_ds + (uint32_t)i * 16u * (uint32_t)ldd_b);
                o.mu[i] = rp_ld_f32(rsm, o_stat + (uint32_t)i * 64u);
                o.rs[i] = rp_ld_f32(rsr, o_stat + (uint32_t)i * 64u);
            }
            o.sc[i] = rp_ld_f32(rs_scale, per_sample.div((uint32_t)(r0 + i * 16 + rsub)) * 4u);      // (raw: 0 without a scale)
        }
    };

    // dW of this wave: rows n0 .. n0+31 (the Linear's outputs = dY's columns) x 256 input columns, as eight 32 x 32 accumulators
    f32x16 dW[8];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int q = 0; q < 16; ++q) dW[kt][q] = 0.f;

    Ops nxt;
    fetch_a(tile_of(0));
    fetch_ops(tile_of(0), nxt);
    commit_a();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // Per-lane LDS addresses of the transposed reads.  A 16-lane group (r16, hi) of a k-step reads the [4 rows][16 columns] block of
    // rows 16 st + 8 hi + 4 h2 + (0..3), columns c0 + 16 r16 + (0..15): lane p16 supplies the 8 bytes of row (p16 >> 2), columns
    // 4 (p16 & 3) .. +3 of the block and receives column p16 of it.  The low four row bits are (hi, h2, p16 >> 2): wg_f of them is
    // f0 | h2 (bit 0 of wg_f is row bit 2), independent of the k-step.  A column block c0 = 32 j (64 bytes = slots 4 j .. 4 j + 3):
    //     slot ^ f = ((j & 3) ^ fq) << 2 | ((2 r16 + ((p16 & 3) >> 1)) ^ (f & 3))      + 16 (j >> 2),   fq = (f >> 2) & 3
    // so a tile needs 2 (h2) x 4 (j & 3) address registers; k-step (+8192) and j >> 2 (+256) are immediate offsets.
    const int trow = 8 * hi + (p16 >> 2);
    const int f0 = wg_f(trow), fq = (f0 >> 2) & 3, cin = 2 * r16 + ((p16 & 3) >> 1);
    uint32_t trS[2][4], trA[2];
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
        const uint32_t base = (uint32_t)((trow + 4 * h2) * kRow + ((cin ^ ((f0 | h2) & 3)) << 4) + 8 * (p16 & 1));
#pragma unroll
        for (int j = 0; j < 4; ++j) trS[h2][j] = (uint32_t)(uintptr_t)sS + base + (uint32_t)((j ^ fq) << 6);
        trA[h2] = (uint32_t)(uintptr_t)sA + base + (uint32_t)(((wave & 3) ^ fq) << 6) + (uint32_t)((wave >> 2) << 8);
    }
    // k-loop operands: rows n0 + r (W^T) and r (dY tile) share row & 15, hence the swizzle
    const int fr = wg_f(r);
    const uint32_t wrow = (uint32_t)(uintptr_t)smem + (uint32_t)((n0 + r) * kRow), arow = (uint32_t)(uintptr_t)sA + (uint32_t)(r * kRow);

    for (int s = 0; s < n_tiles; ++s) {
        // ------------------------------------------------------------------------------------------------- P1
        const Ops cur = nxt;                                // operands of this tile's row phase (fetched a whole tile ago)
        fetch_a(tile_of(s + 1));                            // (past the last tile: empty buffers)
        fetch_ops(tile_of(s + 1), nxt);
        asm volatile("" ::: "memory");
        {
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                int ff = fr;
                asm volatile("" : "+v"(ff));               // (keeps the swizzled address arithmetic at its k-step: see edge_rows_kernel)
                const uint32_t t = (uint32_t)(((2 * ks + hi) ^ ff) << 4);
                const F wf = lds_frag<T>(wrow + t);
                const F xf = lds_frag<T>(arow + t);
                acc = mma32(wf, xf, acc);
            }
            // accumulators -> staging tile, rounded to the storage type (what the unfused chain stores)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                uint2 lo = pack4<T>(acc[8 * p], acc[8 * p + 1], acc[8 * p + 2], acc[8 * p + 3]);
                uint2 up = pack4<T>(acc[8 * p + 4], acc[8 * p + 5], acc[8 * p + 6], acc[8 * p + 7]);
                swap_halves(lo, up);
                *reinterpret_cast<uint4*>(sS + wg_off(r, (n0 >> 3) + 2 * p + hi)) = make_uint4(lo.x, lo.y, up.x, up.y);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ------------------------------------------------------------------------------------------------- P2
        {
            const int64_t m0 = tile_of(s) * kBM;
            const __amdgpu_buffer_rsrc_t rs_out = tile_rsrc(a.out, ldo_b, N * 2, m0, a.M);
            const __amdgpu_buffer_rsrc_t rs_out2 = tile_rsrc(a.out2, ldo2_b, N * 2, m0, kOp2 ? a.M : 0);
#pragma unroll
            for (int i = 0; i < kPass; ++i) {
                const int row = i * 16 + rsub;
                const int64_t m = m0 + row;
                const uint32_t po = (uint32_t)i * 16u;
                char* const slot = sS + wg_off(row, ch);
                f32x2 v[4], x[4];
                rp_unpack<T>(*reinterpret_cast<const uint4*>(slot), v);
                const float sc_i = has_scale ? cur.sc[i] : 1.f;
                if constexpr (EPI == EPI_GELU_BWD) {
                    f32x2 pv[4], o[4];
                    rp_unpack<T>(cur.o1[i], pv);
                    bool keep[8] = {true, true, true, true, true, true, true, true};
                    if (thresh) keep_vector<8>(step_seed(a.dropout_seed, seed_ctr), (m * N + ch * 8) >> 3, thresh, keep);
                    const float ik = inv_keep * sc_i;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        f32x2 e;
                        const f32x2 cdf = gelu_cdf2(pv[k], e);
                        const f32x2 dy = has_scale ? rp_round<T>(v[k] * sc_i) : v[k];
                        const f32x2 y = dy * (cdf + pv[k] * 0.3989422804014327f * e) * inv_keep;
                        o[k].x = keep[2 * k] ? y.x : 0.f;
                        o[k].y = keep[2 * k + 1] ? y.y : 0.f;
                        cs_x[k] += rp_round<T>(o[k]);
                        // the forward's activation, recomputed (edge_rows_kernel EPI_GELU: gl = keep ? v * cdf * ik : 0)
                        const f32x2 g = pv[k] * cdf * ik;
                        x[k].x = keep[2 * k] ? g.x : 0.f;
                        x[k].y = keep[2 * k + 1] ? g.y : 0.f;
                    }
                    rp_st16(rs_out, o_out + po * (uint32_t)ldo_b, rp_pack<T>(o));
                } else {
                    const float mu = cur.mu[i], rs = cur.rs[i];
                    f32x2 sv[4], ds[4], xh[4], gg[4];
                    rp_unpack<T>(cur.o1[i], sv);
                    rp_unpack<T>(cur.o2[i], ds);
                    f32x2 p1 = rp_splat(0.f), p2 = rp_splat(0.f);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        xh[k] = (sv[k] - mu) * rs;
                        gg[k] = v[k] * gam[k];
                        p1 += gg[k];
                        p2 += gg[k] * xh[k];
                        cs_g[k] += v[k] * xh[k];
                        cs_b[k] += v[k];
                        x[k] = xh[k] * gam[k] + bet[k];     // the forward's LayerNorm output (rows past M: rs = 0, but beta != 0 -> zeroed below)
                    }
                    const float c1 = rp_row_sum(p1.x + p1.y) * (1.f / N), c2 = rp_row_sum(p2.x + p2.y) * (1.f / N);
                    f32x2 d[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) d[k] = (gg[k] - c1 - xh[k] * c2) * rs + ds[k];
                    rp_st16(rs_out, o_out + po * (uint32_t)ldo_b, rp_pack<T>(d));
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        d[k] = rp_round<T>(rp_round<T>(d[k]) * sc_i);
                        cs_x[k] += d[k];
                    }
                    rp_st16(rs_out2, o_out2 + po * (uint32_t)ldo2_b, rp_pack<T>(d));
                    if (m >= a.M) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) x[k] = rp_splat(0.f);
                    }
                }
                *reinterpret_cast<uint4*>(slot) = rp_pack<T>(x);
                asm volatile("" ::: "memory");               // (one pass after the other: interleaved, their temporaries double)
            }
        }
        // dY^T of this wave's 32 columns, for P3: read at the END of the row phase (the dY tile is stable from the first barrier of the
        // tile until commit_a below, and nobody reads it after this: the next tile is committed behind P3 without another barrier)
        F aT[2];
#pragma unroll
        for (int st = 0; st < 2; ++st) aT[st] = frag_of<T>(tr_read(trA[0] + st * 16 * kRow), tr_read(trA[1] + st * 16 * kRow));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ------------------------------------------------------------------------------------------------- P3
#pragma unroll
        for (int st = 0; st < 2; ++st) {
#pragma unroll
            for (int kt = 0; kt < 8; ++kt) {
                const uint32_t off = (uint32_t)(st * 16 * kRow + (kt >> 2) * 256);
                const F bT = frag_of<T>(tr_read(trS[0][kt & 3] + off), tr_read(trS[1][kt & 3] + off));
                dW[kt] = mma32(aT[st], bT, dW[kt]);
            }
        }
        commit_a();                                          // the next dY tile (nobody reads the tile after P1)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // ---- dW plane of this workgroup: accumulator register q of lane (r, hi) = dW[n0 + acc_row(q, hi)][32 kt + r]
    {
        float* plane = a.dw_partial + (int64_t)blockIdx.x * N * K;
#pragma unroll
        for (int kt = 0; kt < 8; ++kt)
#pragma unroll
            for (int q = 0; q < 16; ++q) plane[(n0 + acc_row(q, hi)) * K + 32 * kt + r] = dW[kt][q];
    }

    if (a.colsum_partial) {
        // fold the 16 row groups, fixed order: [16][planes][256] floats in LDS (W^T is dead: every wave is past the last barrier)
        constexpr int kPl = EPI == EPI_LN_BWD ? 3 : 1;
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if constexpr (EPI == EPI_LN_BWD) {
                *reinterpret_cast<f32x2*>(red + (rsub * 3 + 0) * N + ch * 8 + 2 * k) = cs_g[k];
                *reinterpret_cast<f32x2*>(red + (rsub * 3 + 1) * N + ch * 8 + 2 * k) = cs_b[k];
            }
            *reinterpret_cast<f32x2*>(red + (rsub * kPl + kPl - 1) * N + ch * 8 + 2 * k) = cs_x[k];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        float* part = a.colsum_partial + (int64_t)blockIdx.x * kPl * N;
        for (int c = tid; c < kPl * N; c += kThreads) {
            float t = 0.f;
#pragma unroll
            for (int s_ = 0; s_ < 16; ++s_) t += red[s_ * kPl * N + c];
            part[c] = t;
        }
    }
}

template <typename T, int EPI>
static int launch(const tgt_edge_linear_args& a, int grid, hipStream_t st) {
    static bool attr_set[16] = {};
    if (!dyn_lds_once(attr_set, reinterpret_cast<const void*>(&edge_rows_wgrad_kernel<T, EPI>), kLds))
        return set_error(TGT_ERR_LAUNCH, "edge_rows_wgrad_kernel: cannot reserve %d bytes of LDS", kLds);
    hipLaunchKernelGGL((edge_rows_wgrad_kernel<T, EPI>), dim3((unsigned)grid), dim3(kThreads), kLds, st, a, seed_counter());
    return check_launch("edge_rows_wgrad_kernel");
}

}  // namespace wg

// tgt_edge_linear with dw_partial: K = N = 256, a backward epilogue, 16-bit
bool edge_wgrad_eligible(const tgt_edge_linear_args& a) {
    if (!a.dw_partial) return false;
    if (a.K != 256 || a.N != 256 || (a.epilogue != EPI_GELU_BWD && a.epilogue != EPI_LN_BWD)) return false;
    if (a.dtype != TGT_BF16 && a.dtype != TGT_F16) return false;
    if (a.epilogue == EPI_LN_BWD && (!a.gamma || !a.beta || !a.mean || !a.rstd || !a.res)) return false;
    return true;
}

int edge_wgrad_run(const tgt_edge_linear_args& a, int grid, hipStream_t st) {
    if (a.epilogue == EPI_GELU_BWD)
        return a.dtype == TGT_BF16 ? wg::launch<bf16_t, EPI_GELU_BWD>(a, grid, st) : wg::launch<f16_t, EPI_GELU_BWD>(a, grid, st);
    return a.dtype == TGT_BF16 ? wg::launch<bf16_t, EPI_LN_BWD>(a, grid, st) : wg::launch<f16_t, EPI_LN_BWD>(a, grid, st);
}

}  // namespace tgt
