// Row-tile GEMMs of the edge channel with the neighbouring passes fused in -- gfx950.
//
// The reference's edge channel is a chain of nn.LayerNorm -> nn.Linear -> (activation) -> nn.Linear ->
// residual `add_` on the (B*N*N, 256) edge rows (lib/tgt/layers/layers.py:37-38,:62-80,:155-160,:262-294;
// lib/tgt/layers/triplet.py:207-211,:229-230,:248-249).  With K = 256 every one of these Linears is
// HBM-bound (arithmetic intensity K*N/(K+N) <= 128 FLOP/B against a ridge of ~310), so what matters is how
// often the 134 MB edge tensor crosses HBM, not MFMA utilisation.  This kernel is one GEMM
//     out[M, N] = epilogue( A[M, K] . W[N, K]^T + bias ),   K in {64, 128, 256}
// whose epilogue absorbs the passes around the library GEMM it replaces:
//   EPI_BIAS  plain                                   (lin_EG, the third-arm E/G projection)
//   EPI_GELU  pre-activation + dropout(gelu(.))       (lin_W1 of the FFN)
//   EPI_RESID res + DropPath-scale[graph] * (.)       (lin_O_e, lin_W2: the result IS the new stream), optionally followed
//             by the LayerNorm of the NEW row = the entry of the next pre-norm sub-block
//   EPI_GELU_BWD   (.) * gelu'(pre) * keep / (1-p)    (data gradient through lin_W2 and the activation)
//   EPI_LN_BWD     LayerNorm backward of the result + the gradient arriving on the residual stream,
//                  dgamma / dbeta / bias-gradient column sums as per-workgroup partials
// so that the standalone LayerNorm / residual / GELU sweeps over the edge tensor disappear.
//
// Two kernels: the weight-resident SLICE kernel (narrow outputs, EPI_BIAS) and the ROW-PHASE kernel (N = 256, whole-row
// epilogues, two wave roles).  v_mfma_f32_32x32x16 with the WEIGHT rows as the A operand and the activation rows as the B
// operand: the result is transposed (lane = row, registers = columns), which leaves 4 consecutive columns per register quad:
// a v_permlane32_swap pairs two quads into one 16-byte store / load per lane.  (A third, general tile kernel -- K > 256 in
// chunks, LayerNorm prologue -- was parity-green but slower than the library everywhere and was removed in round 3.)
#include <cstdlib>
#include <type_traits>
#include "common.hpp"

namespace tgt {

enum { EPI_BIAS = 0, EPI_GELU = 1, EPI_RESID = 2, EPI_GELU_BWD = 3, EPI_LN_BWD = 4 };

struct EgGeo {                    // LDS geometry of the A tile for a given chunk width
    int rowbytes, rpw, mask;
    __device__ __forceinline__ EgGeo(int kc) {
        rowbytes = kc * 2;
        rpw = rowbytes >= 256 ? 1 : 256 / rowbytes;            // rows per 256-byte bank window
        const int slots = rowbytes / 16;
        mask = (slots < 16 ? slots : 16) - 1;
    }
    __device__ __forceinline__ int off(int row, int slot) const {
        return row * rowbytes + ((slot ^ ((row / rpw) & mask)) << 4);
    }
};

template <typename T>
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
    T t[4] = {from_f32<T>(a), from_f32<T>(b), from_f32<T>(c), from_f32<T>(d)};
    uint2 r;
    __builtin_memcpy(&r, t, 8);
    return r;
}
template <typename T>
__device__ __forceinline__ void unpack4(uint2 r, float* v) {
    T t[4];
    __builtin_memcpy(t, &r, 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = to_f32(t[i]);
}
// half-wave exchange: afterwards (a | b) of lanes < 32 is what (a of lane, a of lane+32) were, and (a | b)
// of lanes >= 32 what (b of lane-32, b of lane) were
__device__ __forceinline__ void swap_halves(uint2& a, uint2& b) {
    auto r0 = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
    auto r1 = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
    a.x = r0[0]; b.x = r0[1];
    a.y = r1[0]; b.y = r1[1];
}

// The 16 accumulator values of one 32x32 block of lane (r, hi) are columns  nbase + 8g + 4hi + j  (g = q>>2,
// j = q&3) of row m.  Quads g = 2p and 2p+1 are paired: after the exchange lanes < 32 hold columns
// nbase+16p .. +7 and lanes >= 32 columns nbase+16p+8 .. +15 of their row: one 16-byte access each.
template <typename T>
__device__ __forceinline__ void store_block(T* base, int64_t ld, int64_t m, int64_t M, int nbase, int N, int hi, const float* v) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        uint2 a = pack4<T>(v[8 * p], v[8 * p + 1], v[8 * p + 2], v[8 * p + 3]);
        uint2 b = pack4<T>(v[8 * p + 4], v[8 * p + 5], v[8 * p + 6], v[8 * p + 7]);
        swap_halves(a, b);
        const int col = nbase + 16 * p + 8 * hi;
        if (m < M && col < N) *reinterpret_cast<uint4*>(base + m * ld + col) = make_uint4(a.x, a.y, b.x, b.y);
    }
}
template <typename T>
__device__ __forceinline__ void load_block(const T* base, int64_t ld, int64_t m, int64_t M, int nbase, int N, int hi, float* v) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int col = nbase + 16 * p + 8 * hi;
        uint4 L = make_uint4(0, 0, 0, 0);
        if (m < M && col < N) L = *reinterpret_cast<const uint4*>(base + m * ld + col);
        uint2 a = make_uint2(L.x, L.y), b = make_uint2(L.z, L.w);
        swap_halves(a, b);
        unpack4<T>(a, v + 8 * p);
        unpack4<T>(b, v + 8 * p + 4);
    }
}

// keep flags of the 4 consecutive elements (row m, columns n .. n+3, n % 4 == 0) of an (M, N) tensor under
// the generator of elementwise.hip / common.hpp keep_vector<8>: words (n%8)/2 and (n%8)/2 + 1 of vector (m*N+n)/8
__device__ __forceinline__ void keep4(uint64_t seed, int64_t m, int N, int n, uint32_t thresh, bool* keep) {
    const int64_t vec = (m * N + n) >> 3;
    const uint32_t lo = (uint32_t)vec, hi = (uint32_t)((uint64_t)vec >> 32);
    const uint32_t base = mix32(lo ^ (uint32_t)seed) ^ mix32(hi + (uint32_t)(seed >> 32) + 0x9e3779b9u);
    const int w0 = (n & 7) >> 1;
#pragma unroll
    for (int w = 0; w < 2; ++w) {
        const uint32_t r = mix32(base + (uint32_t)(w0 + w + 1) * 0x9e3779b9u);
        keep[2 * w] = (r & 0xffffu) >= thresh;
        keep[2 * w + 1] = (r >> 16) >= thresh;
    }
}

// raw 16-byte pieces of one 32x32 block in the store_block / load_block addressing (issued early, decoded late)
__device__ __forceinline__ void load_raw(const void* base, int esz_ld_bytes_unused, int64_t off_elems, bool ok, uint4& L) {
    L = make_uint4(0, 0, 0, 0);
    if (ok) L = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(base) + off_elems);
}
template <typename T>
__device__ __forceinline__ void load_block_raw(const T* base, int64_t ld, int64_t m, int64_t M, int nbase, int N, int hi, uint4 (&L)[2]) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int col = nbase + 16 * p + 8 * hi;
        load_raw(base, 0, m * ld + col, m < M && col < N, L[p]);
    }
}
template <typename T>
__device__ __forceinline__ void decode_block(const uint4 (&L)[2], float* v) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        uint2 a = make_uint2(L[p].x, L[p].y), b = make_uint2(L[p].z, L[p].w);
        swap_halves(a, b);
        unpack4<T>(a, v + 8 * p);
        unpack4<T>(b, v + 8 * p + 4);
    }
}

static int eg_num_cus();
// ---------------------------------------------------------------------------------------------------------------
// Weight-resident slice kernel: the fast path for K <= 256.
//
// A workgroup (8 waves, one per CU) owns ONE column slice of the output (32*WN columns) for its whole life and
// keeps that slice of the weight in registers (K/16 operand fragments = 32 columns per wave, <= 64 VGPRs): after
// the first microsecond it never fetches a weight again.  It walks 128-row tiles of A; tile i+1 streams HBM -> LDS by
// LDS-DMA into the other buffer while the matrix cores work on tile i, and the k-loop is straight-line LDS reads +
// MFMAs (no vector-memory instruction, so nothing in it can wait on the DMA: vmcnt is an in-order counter).  The
// epilogue's stores stay in flight under the next tile.  WN = waves along the columns (8: 256 columns x 4 row blocks
// per wave; 4: 128 columns x 2 row blocks; 2: 64 columns x 1 row block).  Wide outputs (the 1536-channel Q/K/V
// projection) are cut into 256-column slices; the workgroups of one row group sit on one XCD (block b runs on XCD
// b % 8), so an A tile comes from HBM once and from that XCD's L2 for the other slices.
// Epilogues as above, plus EPI_RESID with gamma != NULL: LayerNorm of the NEW stream row as a second output
// (y = LN(out), mean, rstd) -- the fused `residual add + LayerNorm` entry of the next sub-block, so the consumer
// needs no LayerNorm prologue.
// ---------------------------------------------------------------------------------------------------------------
// RB = 32-row blocks per tile (rows per tile kBM = 32*RB): 4, or 2 for the register-hungry epilogues
template <typename T, int KS, int WN, int EPI, int RB>
__global__ void __launch_bounds__(512, 2) edge_slice_kernel(const tgt_edge_linear_args a, int groups) {
    using F = frag_t<T>;
    constexpr int WM = 8 / WN, MB = RB / WM, kBM = 32 * RB, K = KS * 16, kRowBytes = K * 2, kBufBytes = kBM * kRowBytes, NT = WN * 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hi = lane >> 5;
    const int wn = wave % WN, wm = wave / WN;             // column block / row group of this wave
    const int N = a.N;
    float* red = reinterpret_cast<float*>(smem + 2 * kBufBytes);      // [WN][128 rows][2]
    const T* A = reinterpret_cast<const T*>(a.a);
    const T* W = reinterpret_cast<const T*>(a.w);
    const EgGeo g(K);
    // block -> (row group, column slice): slices of one row group are consecutive blocks of ONE XCD
    const int n_slices = (N + NT - 1) / NT;
    const int b = blockIdx.x, xcd = b & 7, j = b >> 3;
    const int grp = (j / n_slices) * 8 + xcd, slice = j % n_slices;
    if (grp >= groups) return;
    const int n0 = slice * NT + wn * 32;
    const bool active = n0 < N;
    const int64_t row_tiles = (a.M + kBM - 1) / kBM;
    const int ablate = a._pad0;
    const int rbase = wm * MB * 32;                        // first row (inside the tile) of this wave

    auto stage = [&](int64_t tile, int buf) {
        char* xs = smem + buf * kBufBytes;
        constexpr int spr = K >> 3, total = kBM * spr;     // a multiple of 64 (whole wave instructions)
#pragma unroll
        for (int p0 = 0; p0 < total; p0 += 512) {
            const int pc = p0 + tid;
            const int row = pc / spr, ps = pc % spr;
            int64_t m = tile * kBM + row;
            m = m < a.M ? m : a.M - 1;
            const T* src = A + m * a.lda + ((ps ^ ((row / g.rpw) & g.mask)) << 3);
            if ((total % 512 == 0 || p0 + wave * 64 < total) && !(ablate & 4))
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(xs + (p0 + wave * 64) * 16), 16, 0, 0);
        }
    };

    // the weight slice of this wave (32 columns x K), resident
    F wr[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int n = n0 + r;
        wr[ks] = n < N ? load_frag<T>(W + (int64_t)n * a.ldw + ks * 16 + 8 * hi) : zero_frag<T>();
    }
    uint2 braw[4];
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        const int n = n0 + 8 * gq + 4 * hi;
        braw[gq] = make_uint2(0, 0);
        if (a.bias && n < N) braw[gq] = *reinterpret_cast<const uint2*>(reinterpret_cast<const T*>(a.bias) + n);
    }
    const uint32_t thresh = a.dropout_p <= 0.f ? 0u : (uint32_t)fminf(65535.f, fmaxf(1.f, rintf(a.dropout_p * 65536.f)));
    const float inv_keep = a.dropout_p <= 0.f ? 1.f : 1.f / (1.f - a.dropout_p);

    int64_t tile = grp;
    if (tile >= row_tiles) return;
    stage(tile, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    for (int it = 0; tile < row_tiles; ++it, tile += groups) {
        const char* xs = smem + (it & 1) * kBufBytes;
        const int64_t m0 = tile * kBM + rbase;
        if (tile + groups < row_tiles) stage(tile + groups, (it + 1) & 1);

        // the epilogue's (M, N) operand (residual / pre-activation / LayerNorm input) comes from HBM: issue it now, use it
        // after the k-loop (issued BEFORE nothing it must wait for: the DMA above is older, but has the same k-loop to land)
        uint4 opr[MB][2];
        if constexpr (EPI == EPI_RESID || EPI == EPI_GELU_BWD || EPI == EPI_LN_BWD) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
                load_block_raw<T>(reinterpret_cast<const T*>(a.res), a.ldr, m0 + mb * 32 + r, a.M, n0, N, hi, opr[mb]);
        }

        f32x16 acc[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[mb][q] = 0.f;
        if (active && !(ablate & 8)) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                F xf[MB];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) xf[mb] = load_frag<T>(reinterpret_cast<const T*>(xs + g.off(rbase + mb * 32 + r, 2 * ks + hi)));
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) acc[mb] = mma32(wr[ks], xf[mb], acc[mb]);
            }
        }

        // the next tile's DMA has had the whole k-loop to land and nothing younger is outstanding (the previous tile's
        // stores are older): draining here costs nothing, and this tile's stores stay in flight under the next k-loop
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

        float bv[16];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) unpack4<T>(braw[gq], bv + 4 * gq);
        T* out = reinterpret_cast<T*>(a.out);
        if (ablate & 2) {
            float t = 0.f;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) t += acc[mb][0] + acc[mb][15];
            if (t == 123.456f) reinterpret_cast<float*>(a.out)[tid] = t;
        } else if constexpr (EPI == EPI_BIAS) {
            if (active) {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const int64_t m = m0 + mb * 32 + r;
                    const float al = a.out_scale ? a.out_scale[(m < a.M ? m : a.M - 1) / a.rows_per_sample] : 1.f;
                    float v[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) v[q] = (acc[mb][q] + bv[q]) * al;
                    store_block<T>(out, a.ldo, m, a.M, n0, N, hi, v);
                }
            }
        } else if constexpr (EPI == EPI_GELU) {
            T* pre = reinterpret_cast<T*>(a.out2);
            if (active) {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const int64_t m = m0 + mb * 32 + r;
                    float v[16], gl[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) v[q] = acc[mb][q] + bv[q];
                    store_block<T>(pre, a.ldo2, m, a.M, n0, N, hi, v);
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        bool keep[4] = {true, true, true, true};
                        if (thresh) keep4(a.dropout_seed, m, N, n0 + 8 * gq + 4 * hi, thresh, keep);
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            const float x = to_f32(from_f32<T>(v[4 * gq + jj]));      // gelu of the value as stored
                            float e;
                            const float cdf = gelu_cdf(x, e);
                            gl[4 * gq + jj] = keep[jj] ? x * cdf * inv_keep : 0.f;
                        }
                    }
                    store_block<T>(out, a.ldo, m, a.M, n0, N, hi, gl);
                }
            }
        } else if constexpr (EPI == EPI_RESID) {
            const bool ln = a.gamma != nullptr;               // LayerNorm of the new stream row as a second output
            float v[MB][16];
            float s1[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int64_t m = m0 + mb * 32 + r;
                const float sc = a.row_scale ? a.row_scale[(m < a.M ? m : a.M - 1) / a.rows_per_sample] : 1.f;
                float rv[16], p1 = 0.f;
                decode_block<T>(opr[mb], rv);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    // LayerNorm sees the stream value as stored (rounded to its storage type)
                    const float t = to_f32(from_f32<T>(rv[q] + (acc[mb][q] + bv[q]) * sc));
                    v[mb][q] = t;
                    p1 += (n0 + acc_row(q, hi) < N) ? t : 0.f;
                }
                if (active) store_block<T>(out, a.ldo, m, a.M, n0, N, hi, v[mb]);
                s1[mb] = p1 + xhalf(p1);
            }
            if (ln) {                                           // uniform over the grid
                const float invC = 1.f / (float)N;
                if (hi == 0) {
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) red[(wn * kBM + rbase + mb * 32 + r) * 2] = s1[mb];
                }
                { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
                float mean[MB], rstd[MB];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const int row = rbase + mb * 32 + r;
                    float t = 0.f;
#pragma unroll
                    for (int w_ = 0; w_ < WN; ++w_) t += red[(w_ * kBM + row) * 2];
                    mean[mb] = t * invC;
                    float p2 = 0.f;
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const float d = (n0 + acc_row(q, hi) < N) ? v[mb][q] - mean[mb] : 0.f;
                        p2 += d * d;
                    }
                    p2 += xhalf(p2);
                    if (hi == 0) red[(wn * kBM + row) * 2 + 1] = p2;
                }
                { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
                T* Y = reinterpret_cast<T*>(a.y);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const int row = rbase + mb * 32 + r;
                    const int64_t m = m0 + mb * 32 + r;
                    float t = 0.f;
#pragma unroll
                    for (int w_ = 0; w_ < WN; ++w_) t += red[(w_ * kBM + row) * 2 + 1];
                    rstd[mb] = rsqrtf(t * invC + a.eps);
                    if (wn == 0 && hi == 0 && m < a.M) {
                        if (a.mean) a.mean[m] = mean[mb];
                        if (a.rstd) a.rstd[m] = rstd[mb];
                    }
                    float yv[16];
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int n = n0 + 8 * gq + 4 * hi;
                        float4 gm = make_float4(0.f, 0.f, 0.f, 0.f), bt = gm;
                        if (n < N) {
                            gm = *reinterpret_cast<const float4*>(a.gamma + n);
                            bt = *reinterpret_cast<const float4*>(a.beta + n);
                        }
                        yv[4 * gq + 0] = (v[mb][4 * gq + 0] - mean[mb]) * rstd[mb] * gm.x + bt.x;
                        yv[4 * gq + 1] = (v[mb][4 * gq + 1] - mean[mb]) * rstd[mb] * gm.y + bt.y;
                        yv[4 * gq + 2] = (v[mb][4 * gq + 2] - mean[mb]) * rstd[mb] * gm.z + bt.z;
                        yv[4 * gq + 3] = (v[mb][4 * gq + 3] - mean[mb]) * rstd[mb] * gm.w + bt.w;
                    }
                    if (active) store_block<T>(Y, a.ldy, m, a.M, n0, N, hi, yv);
                }
            }
        } else if constexpr (EPI == EPI_GELU_BWD) {
            if (active) {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const int64_t m = m0 + mb * 32 + r;
                    const float al = a.out_scale ? a.out_scale[(m < a.M ? m : a.M - 1) / a.rows_per_sample] : 1.f;
                    float pv[16], v[16];
                    decode_block<T>(opr[mb], pv);
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        bool keep[4] = {true, true, true, true};
                        if (thresh) keep4(a.dropout_seed, m, N, n0 + 8 * gq + 4 * hi, thresh, keep);
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            const float x = pv[4 * gq + jj];
                            float e;
                            const float cdf = gelu_cdf(x, e);
                            const float dy = to_f32(from_f32<T>(acc[mb][4 * gq + jj] * al));
                            v[4 * gq + jj] = keep[jj] ? dy * (cdf + x * 0.3989422804014327f * e) * inv_keep : 0.f;
                        }
                    }
                    store_block<T>(out, a.ldo, m, a.M, n0, N, hi, v);
                }
            }
        } else {       // EPI_LN_BWD: acc = dy at the output of LayerNorm(res; gamma), whole rows in this workgroup (one slice)
            const T* dsin = reinterpret_cast<const T*>(a.ds_in);
            float g16[16];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int n = n0 + 8 * gq + 4 * hi;
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if (n < N) t = *reinterpret_cast<const float4*>(a.gamma + n);
                g16[4 * gq] = t.x; g16[4 * gq + 1] = t.y; g16[4 * gq + 2] = t.z; g16[4 * gq + 3] = t.w;
            }
            float cs_a[16], cs_b[16], mu[MB], rs[MB], s1[MB], s2[MB];
#pragma unroll
            for (int q = 0; q < 16; ++q) cs_a[q] = cs_b[q] = 0.f;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int64_t m = m0 + mb * 32 + r;
                const bool ok = m < a.M;
                mu[mb] = ok ? a.mean[m] : 0.f;
                rs[mb] = ok ? a.rstd[m] : 0.f;
                float sv[16], p1 = 0.f, p2 = 0.f;
                decode_block<T>(opr[mb], sv);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const bool cok = ok && (n0 + acc_row(q, hi) < N);
                    const float dy = cok ? to_f32(from_f32<T>(acc[mb][q])) : 0.f;      // dy as the unfused chain stores it
                    const float x = cok ? (sv[q] - mu[mb]) * rs[mb] : 0.f;
                    const float gg = dy * g16[q];
                    acc[mb][q] = gg;
                    p1 += gg;
                    p2 += gg * x;
                    cs_a[q] += dy * x;
                    cs_b[q] += dy;
                }
                s1[mb] = p1 + xhalf(p1);
                s2[mb] = p2 + xhalf(p2);
            }
            if (hi == 0) {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    red[(wn * kBM + rbase + mb * 32 + r) * 2] = s1[mb];
                    red[(wn * kBM + rbase + mb * 32 + r) * 2 + 1] = s2[mb];
                }
            }
            // fold a per-lane 16-column partial over the 32 lanes of each half-wave (16 + 8+4+2+1 exchanges): lanes
            // r < 16 end with the total of register index q = r; the WM row groups of the workgroup write separate rows
            auto fold_store = [&](float (&v)[16], float* dst) {
#pragma unroll
                for (int q = 0; q < 16; ++q) v[q] += __shfl_xor(v[q], 16, 64);
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) {
                    const int width = 8 >> s_;
                    const bool upper = (r & width) != 0;
#pragma unroll
                    for (int c = 0; c < width; ++c) {
                        const float mine = upper ? v[c + width] : v[c];
                        const float send = upper ? v[c] : v[c + width];
                        v[c] = mine + __shfl_xor(send, width, 64);
                    }
                }
                const int q = r & 15;
                const int n = n0 + (q & 3) + 8 * (q >> 2) + 4 * hi;
                if (r < 16 && n < N) dst[n] = v[0];
            };
            float* part = a.colsum_partial ? a.colsum_partial + (tile * WM + wm) * 3 * N : nullptr;
            if (part) {
                fold_store(cs_a, part);
                fold_store(cs_b, part + N);
            }
            { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
            const float invC = 1.f / (float)N;
            T* dres = reinterpret_cast<T*>(a.out);
            T* dx = reinterpret_cast<T*>(a.out2);
#pragma unroll
            for (int q = 0; q < 16; ++q) cs_a[q] = 0.f;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int row = rbase + mb * 32 + r;
                float c1 = 0.f, c2 = 0.f;
#pragma unroll
                for (int w_ = 0; w_ < WN; ++w_) {
                    c1 += red[(w_ * kBM + row) * 2];
                    c2 += red[(w_ * kBM + row) * 2 + 1];
                }
                c1 *= invC;
                c2 *= invC;
                const int64_t m = m0 + mb * 32 + r;
                const float sc = a.row_scale ? a.row_scale[(m < a.M ? m : a.M - 1) / a.rows_per_sample] : 1.f;
                float dv[16], ds[16], sv[16];
                if (dsin) load_block<T>(dsin, a.ld_ds, m, a.M, n0, N, hi, ds);
                decode_block<T>(opr[mb], sv);                  // the raw stream rows are still in registers
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float x = (m < a.M && n0 + acc_row(q, hi) < N) ? (sv[q] - mu[mb]) * rs[mb] : 0.f;
                    float d = rs[mb] * (acc[mb][q] - c1 - x * c2);
                    if (dsin) d += ds[q];
                    dv[q] = d;
                }
                if (active) store_block<T>(dres, a.ldo, m, a.M, n0, N, hi, dv);
                if (dx || part) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const float t = to_f32(from_f32<T>(to_f32(from_f32<T>(dv[q])) * sc));
                        dv[q] = t;
                        cs_a[q] += (m < a.M && n0 + acc_row(q, hi) < N) ? t : 0.f;
                    }
                    if (dx && active) store_block<T>(dx, a.ldo2, m, a.M, n0, N, hi, dv);
                }
            }
            if (part) fold_store(cs_a, part + 2 * N);
        }
        // my pieces of the next tile have landed; after the barrier they have for every wave, and every wave is done
        // reading this tile's buffer (the DMA issued at the top of the next iteration but one overwrites it).
        // Raw barrier: __syncthreads() would also drain the stores just issued.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Row-phase kernel: the 256-column Linears of the edge channel whose epilogue works on WHOLE ROWS
// (residual add [+ LayerNorm of the new row], GELU + dropout, their backward forms, LayerNorm backward).
//
// Measured on the slice kernel above (tools/edge_gemm_bench.py with TGT_EG_ABLATE): its epilogue -- every lane storing
// 16-byte pieces of ITS OWN row, 32 rows per wave instruction -- is what the time goes to: W2+res+LN 0.157 ms, 0.060
// without the stores; the epilogue alone (no A loads, no MFMA) 0.124 ms for 3 tensor passes that stream in 0.08.
// Here the accumulators leave through LDS instead: bias add, round to the storage type (what nn.Linear emits under
// autocast), half-wave exchange, two ds_write_b128 per row block into a row-major staging tile (XOR-swizzled 16-byte
// slots; it reuses the A buffer the k-loop just finished with when K >= 256).  Then a ROW PHASE with the mapping of the
// LayerNorm kernels: thread = (row i*16 + tid/32, 16-byte chunk tid%32), so every global access of a wave is two whole
// 512-byte rows, row reductions are 5 xor-shuffles inside a 32-lane half, the per-column constants (gamma, beta) sit in
// 8 registers for the whole kernel, and the column sums of the LayerNorm backward (dgamma, dbeta, bias gradient) are
// per-thread accumulators across ALL tiles of the persistent workgroup, folded once at the end.
// Weight slice (32 columns x K per wave) resident in registers, A tiles by LDS-DMA into two buffers, as above.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void rp_unpack8(const uint4& raw, float* v) {
    T t[8];
    __builtin_memcpy(t, &raw, 16);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = to_f32(t[i]);
}
template <typename T>
__device__ __forceinline__ uint4 rp_pack8(const float* v) {
    T t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = from_f32<T>(v[i]);
    uint4 raw;
    __builtin_memcpy(&raw, t, 16);
    return raw;
}
// sum over the 32 lanes of a half-wave (one row)
__device__ __forceinline__ float rp_row_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Wave roles.  vmcnt is ONE in-order counter per wave, so a wave that both prefetches and stores can only wait for its
// prefetch together with every store it issued before (measured on two earlier forms of this kernel: load+MFMA time and
// row-phase time simply added up, 0.044 + 0.085 ms for W1+GELU; hipcc additionally answers an outstanding LDS-DMA with
// `s_waitcnt vmcnt(0)` in front of the next LDS read it cannot prove disjoint).  So the two kinds of traffic live in
// different waves of the workgroup:
//   waves 0-7   GEMM role: fetch the next 32-row A tile into registers (plain 16-byte loads; these waves never store, so the
//               compiler's wait before the closing ds_writes counts exactly those loads), k-loop on the current tile, weight
//               slice (32 columns x K per wave) resident in registers, accumulators out through the staging tile;
//   waves 8-15  row role: one pipeline stage behind, the row phase of the previous tile -- operand rows prefetched from
//               global memory a stage ahead into registers, whole-row stores that nobody in this role waits for until
//               the NEXT stage's operands are needed (a full stage later).
// 16 waves = 4 per SIMD (128 registers each): every SIMD holds two waves of each role, so the matrix pipe, the VALU work of
// the row phase and both kinds of memory traffic overlap.  (With 4 + 4 waves the row role ran one wave per SIMD and was
// latency-bound: 0.096 ms for the GELU row phase alone.)  One s_barrier per stage (32 rows) couples the roles; staging
// tiles are double-buffered.
template <typename T, int KS, int EPI>
__global__ void __launch_bounds__(1024, 4) edge_rows_kernel(const tgt_edge_linear_args a) {
    using F = frag_t<T>;
    constexpr int K = KS * 16, N = 256, kBM = 32, kABytes = kBM * K * 2, kSBytes = kBM * N * 2, kPass = 2;
    constexpr bool kOperand = EPI == EPI_RESID || EPI == EPI_GELU_BWD || EPI == EPI_LN_BWD;
    constexpr bool kOp2 = EPI == EPI_LN_BWD;
    constexpr int kOffStage = 2 * kABytes, kOffGB = kOffStage + 2 * kSBytes;      // LDS: A tiles [2] | staging tiles [2] | gamma, beta
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hi = lane >> 5;
    const EgGeo g(K), gs(N);
    const int64_t row_tiles = (a.M + kBM - 1) / kBM;
    const int ablate = a._pad0;
    float* gb = reinterpret_cast<float*>(smem + kOffGB);
    if (blockIdx.x >= row_tiles) return;
    const int n_tiles = (int)((row_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x);       // tiles of this workgroup
    auto tile_of = [&](int s) { return (int64_t)blockIdx.x + (int64_t)s * gridDim.x; };

    if (wave < 8) {
        // ------------------------------------------------------------------------------------------------ GEMM role
        const T* A = reinterpret_cast<const T*>(a.a);
        const T* W = reinterpret_cast<const T*>(a.w);
        const int t4 = tid;                                // 0..511
        const int n0 = wave * 32;
        constexpr int spr = K >> 3, total = kBM * spr, kNA = (total + 511) / 512;
        F wr[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wr[ks] = load_frag<T>(W + (int64_t)(n0 + r) * a.ldw + ks * 16 + 8 * hi);
        uint2 braw[4];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            braw[gq] = make_uint2(0, 0);
            if (a.bias && !(EPI == EPI_RESID && (a.flags & TGT_EDGE_BIAS_SCALED)))      // (BIAS_SCALED: the row phase adds scale * bias)
                braw[gq] = *reinterpret_cast<const uint2*>(reinterpret_cast<const T*>(a.bias) + n0 + 8 * gq + 4 * hi);
        }
        uint4 pre[kNA];
        auto fetch = [&](int64_t tile) {                   // 16-byte pieces (row, slot): consecutive threads = consecutive slots of a row
#pragma unroll
            for (int q = 0; q < kNA; ++q) {
                const int pc = q * 512 + t4;
                const int row = pc / spr, ps = pc % spr;
                int64_t m = tile * kBM + row;
                m = m < a.M ? m : a.M - 1;                 // rows past M re-read row M-1 (never stored)
                if ((total % 512 == 0 || pc < total) && !(ablate & 4)) pre[q] = *reinterpret_cast<const uint4*>(A + m * a.lda + ps * 8);
            }
        };
        auto commit = [&](int buf) {
#pragma unroll
            for (int q = 0; q < kNA; ++q) {
                const int pc = q * 512 + t4;
                const int row = pc / spr, ps = pc % spr;
                if (total % 512 == 0 || pc < total) *reinterpret_cast<uint4*>(smem + buf * kABytes + g.off(row, ps)) = pre[q];
            }
        };
        fetch(tile_of(0));
        commit(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int s = 0; s <= n_tiles; ++s) {
            if (s < n_tiles) {
                const char* xs = smem + (s & 1) * kABytes;
                char* sg = smem + kOffStage + (s & 1) * kSBytes;
                if (s + 1 < n_tiles) fetch(tile_of(s + 1));
                asm volatile("" ::: "memory");            // the prefetch is issued HERE, not sunk towards its use
                f32x16 acc;
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[q] = 0.f;
                if (!(ablate & 8)) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const F xf = load_frag<T>(reinterpret_cast<const T*>(xs + g.off(r, 2 * ks + hi)));
                        acc = mma32(wr[ks], xf, acc);
                    }
                }
                // accumulators -> staging tile: + bias, rounded to the storage type (what nn.Linear emits under autocast);
                // lanes (r, hi) of a row exchange halves so that each holds 8 consecutive columns = one 16-byte slot
                {
                    float bv[16];
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) unpack4<T>(braw[gq], bv + 4 * gq);
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        uint2 lo = pack4<T>(acc[8 * p] + bv[8 * p], acc[8 * p + 1] + bv[8 * p + 1], acc[8 * p + 2] + bv[8 * p + 2],
                                            acc[8 * p + 3] + bv[8 * p + 3]);
                        uint2 up = pack4<T>(acc[8 * p + 4] + bv[8 * p + 4], acc[8 * p + 5] + bv[8 * p + 5], acc[8 * p + 6] + bv[8 * p + 6],
                                            acc[8 * p + 7] + bv[8 * p + 7]);
                        swap_halves(lo, up);
                        *reinterpret_cast<uint4*>(sg + gs.off(r, (n0 >> 3) + 2 * p + hi)) = make_uint4(lo.x, lo.y, up.x, up.y);
                    }
                }
                if (s + 1 < n_tiles) commit((s + 1) & 1);  // (the A buffer of tile s-1: its k-loop ended before the last barrier)
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        return;
    }

    // ------------------------------------------------------------------------------------------------------ row role
    const int t4 = tid - 512;
    const int ch = t4 & 31, rsub = t4 >> 5;               // 16-byte chunk (8 columns) and row inside a group of 16
    if (t4 < N) {
        gb[t4] = a.gamma ? a.gamma[t4] : 1.f;
        gb[N + t4] = (a.gamma && a.beta) ? a.beta[t4] : 0.f;
    }
    const bool ln_out = EPI == EPI_RESID && a.gamma != nullptr;
    const uint32_t thresh = a.dropout_p <= 0.f ? 0u : (uint32_t)fminf(65535.f, fmaxf(1.f, rintf(a.dropout_p * 65536.f)));
    const float inv_keep = a.dropout_p <= 0.f ? 1.f : 1.f / (1.f - a.dropout_p);
    constexpr int kCs = EPI == EPI_LN_BWD ? 8 : 1;
    float cs_g[kCs], cs_b[kCs], cs_x[kCs];               // EPI_LN_BWD: column sums over every row this workgroup processes
#pragma unroll
    for (int j = 0; j < kCs; ++j) cs_g[j] = cs_b[j] = cs_x[j] = 0.f;

    struct Ops { uint4 o1[kOperand ? kPass : 1]; uint4 o2[kOp2 ? kPass : 1]; float mu[kOp2 ? kPass : 1], rs[kOp2 ? kPass : 1], sc[kPass]; };
    auto fetch_ops = [&](int64_t tile, Ops& o) {          // the row phase's operands of `tile`: whole rows, straight from global memory
#pragma unroll
        for (int i = 0; i < kPass; ++i) {
            int64_t m = tile * kBM + i * 16 + rsub;
            m = m < a.M ? m : a.M - 1;
            if (ablate & 4) continue;
            if constexpr (kOperand) o.o1[i] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(a.res) + m * a.ldr + ch * 8);
            if constexpr (kOp2) {
                o.o2[i] = a.ds_in ? *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(a.ds_in) + m * a.ld_ds + ch * 8) : make_uint4(0, 0, 0, 0);
                o.mu[i] = a.mean[m];
                o.rs[i] = a.rstd[m];
            }
            o.sc[i] = 1.f;
            if (EPI == EPI_GELU_BWD ? a.out_scale != nullptr : a.row_scale != nullptr)
                o.sc[i] = (EPI == EPI_GELU_BWD ? a.out_scale : a.row_scale)[m / a.rows_per_sample];
        }
    };
    Ops nxt;
    fetch_ops(tile_of(0), nxt);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float gam[8], bet[8];                                 // this thread's 8 columns, for the whole kernel
    float bsc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};    // EPI_RESID + BIAS_SCALED: the bias, added as row_scale * bias
    if constexpr (EPI == EPI_RESID) {
        if ((a.flags & TGT_EDGE_BIAS_SCALED) && a.bias) rp_unpack8<T>(*reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(a.bias) + ch * 8), bsc);
    }
    {
        const float4 g0 = *reinterpret_cast<const float4*>(gb + ch * 8), g1 = *reinterpret_cast<const float4*>(gb + ch * 8 + 4);
        gam[0] = g0.x; gam[1] = g0.y; gam[2] = g0.z; gam[3] = g0.w; gam[4] = g1.x; gam[5] = g1.y; gam[6] = g1.z; gam[7] = g1.w;
        const float4 b0 = *reinterpret_cast<const float4*>(gb + N + ch * 8), b1 = *reinterpret_cast<const float4*>(gb + N + ch * 8 + 4);
        bet[0] = b0.x; bet[1] = b0.y; bet[2] = b0.z; bet[3] = b0.w; bet[4] = b1.x; bet[5] = b1.y; bet[6] = b1.z; bet[7] = b1.w;
    }
    for (int s = 0; s <= n_tiles; ++s) {
        if (s >= 1 && !(ablate & 2)) {
            const Ops cur = nxt;                           // operands of tile s-1 (fetched a stage ago)
            if (s < n_tiles) fetch_ops(tile_of(s), nxt);
            asm volatile("" ::: "memory");
            const char* sg = smem + kOffStage + ((s - 1) & 1) * kSBytes;
            const int64_t m0 = tile_of(s - 1) * kBM;
#pragma unroll
            for (int i = 0; i < kPass; ++i) {
                const int row = i * 16 + rsub;
                const int64_t m = m0 + row;
                const bool ok = m < a.M;
                const int64_t mc = ok ? m : a.M - 1;
                float v[8];
                rp_unpack8<T>(*reinterpret_cast<const uint4*>(sg + gs.off(row, ch)), v);
                if constexpr (EPI == EPI_GELU) {
                    T* pre = reinterpret_cast<T*>(a.out2);
                    T* out = reinterpret_cast<T*>(a.out);
                    if (ok) *reinterpret_cast<uint4*>(pre + m * a.ldo2 + ch * 8) = rp_pack8<T>(v);
                    bool keep[8] = {true, true, true, true, true, true, true, true};
                    if (thresh) keep_vector<8>(a.dropout_seed, (mc * N + ch * 8) >> 3, thresh, keep);
                    float gl[8];
                    const float ik = inv_keep * cur.sc[i];       // (row_scale: the DropPath factor of the branch, folded into the activation)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float e;
                        const float cdf = gelu_cdf(v[j], e);
                        gl[j] = keep[j] ? v[j] * cdf * ik : 0.f;
                    }
                    if (ok) *reinterpret_cast<uint4*>(out + m * a.ldo + ch * 8) = rp_pack8<T>(gl);
                } else if constexpr (EPI == EPI_GELU_BWD) {
                    const float al = cur.sc[i];
                    float pv[8], o[8];
                    rp_unpack8<T>(cur.o1[i], pv);
                    bool keep[8] = {true, true, true, true, true, true, true, true};
                    if (thresh) keep_vector<8>(a.dropout_seed, (mc * N + ch * 8) >> 3, thresh, keep);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float e;
                        const float cdf = gelu_cdf(pv[j], e);
                        const float dy = a.out_scale ? to_f32(from_f32<T>(v[j] * al)) : v[j];
                        o[j] = keep[j] ? dy * (cdf + pv[j] * 0.3989422804014327f * e) * inv_keep : 0.f;
                    }
                    if (ok) *reinterpret_cast<uint4*>(reinterpret_cast<T*>(a.out) + m * a.ldo + ch * 8) = rp_pack8<T>(o);
                } else if constexpr (EPI == EPI_RESID) {
                    const float sc = cur.sc[i];
                    float rv[8], t[8];
                    rp_unpack8<T>(cur.o1[i], rv);
                    float p1 = 0.f;
                    const bool pres = (a.flags & TGT_EDGE_BIAS_SCALED) != 0;     // x arrived pre-scaled: res + x W^T + scale * bias
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        t[j] = to_f32(from_f32<T>(pres ? rv[j] + v[j] + sc * bsc[j] : rv[j] + v[j] * sc));      // the stream value as stored: LayerNorm sees that
                        p1 += t[j];
                    }
                    if (ok) *reinterpret_cast<uint4*>(reinterpret_cast<T*>(a.out) + m * a.ldo + ch * 8) = rp_pack8<T>(t);
                    if (ln_out) {
                        const float mean = rp_row_sum(p1) * (1.f / N);
                        float p2 = 0.f;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            t[j] -= mean;
                            p2 += t[j] * t[j];
                        }
                        const float rstd = rsqrtf(rp_row_sum(p2) * (1.f / N) + a.eps);
#pragma unroll
                        for (int j = 0; j < 8; ++j) t[j] = t[j] * rstd * gam[j] + bet[j];
                        if (ok) *reinterpret_cast<uint4*>(reinterpret_cast<T*>(a.y) + m * a.ldy + ch * 8) = rp_pack8<T>(t);
                        if (ch == 0 && ok) {
                            if (a.mean) a.mean[m] = mean;
                            if (a.rstd) a.rstd[m] = rstd;
                        }
                    }
                } else if constexpr (EPI == EPI_LN_BWD) {
                    // v = dy at the output of LayerNorm(res; gamma); d_res = rstd (g - mean(g) - xhat mean(g xhat)) + ds_in, g = dy gamma
                    const float mu = cur.mu[i], rs = cur.rs[i], sc = cur.sc[i];
                    float sv[8], ds[8], xh[8], gg[8];
                    rp_unpack8<T>(cur.o1[i], sv);
                    rp_unpack8<T>(cur.o2[i], ds);
                    float p1 = 0.f, p2 = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        xh[j] = ok ? (sv[j] - mu) * rs : 0.f;
                        const float dy = ok ? v[j] : 0.f;
                        gg[j] = dy * gam[j];
                        p1 += gg[j];
                        p2 += gg[j] * xh[j];
                        cs_g[j] += dy * xh[j];
                        cs_b[j] += dy;
                    }
                    const float c1 = rp_row_sum(p1) * (1.f / N), c2 = rp_row_sum(p2) * (1.f / N);
                    float d[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) d[j] = rs * (gg[j] - c1 - xh[j] * c2) + ds[j];
                    if (ok) *reinterpret_cast<uint4*>(reinterpret_cast<T*>(a.out) + m * a.ldo + ch * 8) = rp_pack8<T>(d);
                    if (a.out2 || a.colsum_partial) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            // the x-branch gradient as it is stored (rounded), so that its column sums equal a separate pass's
                            d[j] = to_f32(from_f32<T>(to_f32(from_f32<T>(d[j])) * sc));
                            cs_x[j] += ok ? d[j] : 0.f;
                        }
                        if (a.out2 && ok) *reinterpret_cast<uint4*>(reinterpret_cast<T*>(a.out2) + m * a.ldo2 + ch * 8) = rp_pack8<T>(d);
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    if constexpr (EPI == EPI_LN_BWD) {
        if (a.colsum_partial) {
            // fold the 16 row groups, fixed order: [16][3][256] floats in LDS (the A / staging tiles are dead: every wave is past
            // the last barrier; only the row role takes part from here on -- the GEMM waves have exited, and an exited wave
            // counts as arrived at a barrier)
            float* red = reinterpret_cast<float*>(smem);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                red[(rsub * 3 + 0) * N + ch * 8 + j] = cs_g[j];
                red[(rsub * 3 + 1) * N + ch * 8 + j] = cs_b[j];
                red[(rsub * 3 + 2) * N + ch * 8 + j] = cs_x[j];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            float* part = a.colsum_partial + (int64_t)blockIdx.x * 3 * N;
            for (int c = t4; c < 3 * N; c += 512) {
                float t = 0.f;
#pragma unroll
                for (int s_ = 0; s_ < 16; ++s_) t += red[s_ * 3 * N + c];
                part[c] = t;
            }
        }
    }
}

// test hook (tgt_edge_linear_set_grid_cap): at most this many persistent workgroups / row groups per launch, so that a small
// problem walks several tiles per workgroup -- the stage hand-over the BASELINE-size launches (32 tiles each) depend on
static int g_grid_cap = 0;
void edge_linear_set_grid_cap(int cap) { g_grid_cap = cap > 0 ? cap : 0; }

static int er_grid(int64_t M) {
    const int64_t row_tiles = (M + 31) / 32;
    int64_t g = row_tiles < eg_num_cus() ? row_tiles : eg_num_cus();
    if (g_grid_cap && g > g_grid_cap) g = g_grid_cap;
    return (int)g;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: remember it per (kernel instantiation, device)
static bool dyn_lds_once(bool (&done)[16], const void* fn, int lds) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    if (dev >= 16 || !done[dev]) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return false;
        if (dev < 16) done[dev] = true;
    }
    return true;
}

template <typename T, int KS, int EPI>
static int er_launch(const tgt_edge_linear_args& a, hipStream_t st) {
    constexpr int K = KS * 16;
    constexpr int lds0 = 2 * 32 * K * 2 + 2 * 32 * 256 * 2 + 2 * 256 * 4;
    constexpr int lds = lds0 < 49152 ? 49152 : lds0;                        // the final column-sum fold of LN_BWD needs 48 KB
    static bool attr_set[16] = {};
    if (!dyn_lds_once(attr_set, reinterpret_cast<const void*>(&edge_rows_kernel<T, KS, EPI>), lds))
        return set_error(TGT_ERR_LAUNCH, "edge_rows_kernel: cannot reserve %d bytes of LDS", lds);
    hipLaunchKernelGGL((edge_rows_kernel<T, KS, EPI>), dim3((unsigned)er_grid(a.M)), dim3(1024), lds, st, a);
    return check_launch("edge_rows_kernel");
}

template <typename T, int KS>
static int er_dispatch(const tgt_edge_linear_args& a, hipStream_t st) {
    switch (a.epilogue) {
        case EPI_GELU: return er_launch<T, KS, EPI_GELU>(a, st);
        case EPI_RESID: return er_launch<T, KS, EPI_RESID>(a, st);
        case EPI_GELU_BWD: return er_launch<T, KS, EPI_GELU_BWD>(a, st);
        case EPI_LN_BWD: return er_launch<T, KS, EPI_LN_BWD>(a, st);
        default: return set_error(TGT_ERR_INVALID, "edge linear (row-phase kernel): bad epilogue %d", a.epilogue);
    }
}

// the row-phase kernel takes the 256-column Linears with a whole-row epilogue: K in {64, 128, 256}, contiguous-enough rows
static bool er_eligible(const tgt_edge_linear_args& a) {
    if (a.N != 256 || (a.K != 64 && a.K != 128 && a.K != 256) || a.epilogue == EPI_BIAS) return false;
    if (a.gamma && a.epilogue != EPI_RESID && a.epilogue != EPI_LN_BWD) return false;
    if (a.gamma && a.epilogue == EPI_RESID && (!a.beta || !a.y)) return false;
    return true;
}

template <typename T>
static int er_run(const tgt_edge_linear_args& a, hipStream_t st) {
    switch (a.K) {
        case 64: return er_dispatch<T, 4>(a, st);
        case 128: return er_dispatch<T, 8>(a, st);
        default: return er_dispatch<T, 16>(a, st);
    }
}

template <typename T, int KS, int WN, int EPI, int RB>
static int es_launch(const tgt_edge_linear_args& a, hipStream_t st) {
    constexpr int kBM = 32 * RB;
    constexpr int lds = 2 * kBM * KS * 32 + WN * kBM * 2 * 4;
    static bool attr_set[16] = {};
    if (!dyn_lds_once(attr_set, reinterpret_cast<const void*>(&edge_slice_kernel<T, KS, WN, EPI, RB>), lds))
        return set_error(TGT_ERR_LAUNCH, "edge_slice_kernel: cannot reserve %d bytes of LDS", lds);
    const int n_slices = (a.N + 32 * WN - 1) / (32 * WN);
    const int64_t row_tiles = (a.M + kBM - 1) / kBM;
    // one workgroup per CU; per XCD (blocks b % 8) a whole number of row groups x all their slices
    int per_xcd = (eg_num_cus() / 8) / n_slices;
    if (per_xcd < 1) per_xcd = 1;
    int64_t groups = (int64_t)per_xcd * 8;
    if (groups > row_tiles) groups = row_tiles;
    if (g_grid_cap && groups > g_grid_cap) groups = g_grid_cap;
    const int64_t blocks = ((groups + 7) / 8) * n_slices * 8;
    hipLaunchKernelGGL((edge_slice_kernel<T, KS, WN, EPI, RB>), dim3((unsigned)blocks), dim3(512), lds, st, a, (int)groups);
    return check_launch("edge_slice_kernel");
}

template <typename T, int KS, int WN>
static int es_dispatch(const tgt_edge_linear_args& a, hipStream_t st) {
    switch (a.epilogue) {
        case EPI_BIAS: return es_launch<T, KS, WN, EPI_BIAS, 4>(a, st);
        case EPI_GELU: return es_launch<T, KS, WN, EPI_GELU, WN == 8 ? 2 : 4>(a, st);
        case EPI_RESID: return es_launch<T, KS, WN, EPI_RESID, WN == 8 ? 2 : 4>(a, st);
        case EPI_GELU_BWD: return es_launch<T, KS, WN, EPI_GELU_BWD, WN == 8 ? 2 : 4>(a, st);
        case EPI_LN_BWD: return es_launch<T, KS, WN, EPI_LN_BWD, WN == 8 ? 2 : 4>(a, st);
        default: return set_error(TGT_ERR_INVALID, "edge linear (slice kernel): bad epilogue %d", a.epilogue);
    }
}

// the slice kernel takes K in {64, 128, 256} without a LayerNorm prologue; the row-wise epilogues (LayerNorm of
// the new stream row, LN_BWD) need the whole row in one slice (N <= 256)
static bool es_eligible(const tgt_edge_linear_args& a) {
    if (a.K != 64 && a.K != 128 && a.K != 256) return false;
    if (a.epilogue == EPI_LN_BWD) return a.N <= 256;
    if (a.gamma && a.epilogue != EPI_RESID) return false;
    if (a.gamma && a.epilogue == EPI_RESID && (a.N > 256 || !a.beta || !a.y)) return false;
    return true;
}

template <typename T>
static int es_run(const tgt_edge_linear_args& a, hipStream_t st) {
    const int wn = a.N <= 64 ? 2 : (a.N <= 128 ? 4 : 8);
    switch (a.K) {
        case 64: return wn == 2 ? es_dispatch<T, 4, 2>(a, st) : (wn == 4 ? es_dispatch<T, 4, 4>(a, st) : es_dispatch<T, 4, 8>(a, st));
        case 128: return wn == 2 ? es_dispatch<T, 8, 2>(a, st) : (wn == 4 ? es_dispatch<T, 8, 4>(a, st) : es_dispatch<T, 8, 8>(a, st));
        default: return wn == 2 ? es_dispatch<T, 16, 2>(a, st) : (wn == 4 ? es_dispatch<T, 16, 4>(a, st) : es_dispatch<T, 16, 8>(a, st));
    }
}

static int eg_num_cus() {                 // of the CURRENT device (a process may drive several)
    static int n[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (!n[dev]) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) n[dev] = prop.multiProcessorCount;
        if (n[dev] <= 0) n[dev] = 256;
    }
    return n[dev];
}

// rows of colsum_partial the caller provides (ZERO-FILLED: which kernel runs, and so which rows are written, also depends on N):
// one per (128-row tile, row group of waves WM = 8 / WN) of the slice kernel -- an upper bound for the row-phase kernel,
// which writes one row per workgroup
int edge_linear_parts(int64_t M, int N) {
    const int wn = N <= 64 ? 2 : (N <= 128 ? 4 : 8);
    return wn == 8 ? (int)((M + 31) / 32) : (int)((M + 127) / 128) * (8 / wn);        // (row-phase LN_BWD: 32-row tiles)
}

int edge_linear_supported(const tgt_edge_linear_args* a) {
    if (!a) return 0;
    if (a->dtype != TGT_BF16 && a->dtype != TGT_F16) return 0;
    const int K = a->K, N = a->N;
    if ((K != 64 && K != 128 && K != 256) || N < 8 || N % 8) return 0;
    if (a->gamma && a->epilogue != EPI_RESID && a->epilogue != EPI_LN_BWD) return 0;          // (no LayerNorm prologue)
    if (a->gamma && a->epilogue == EPI_RESID && N > 256) return 0;
    if (a->epilogue == EPI_LN_BWD && (N > 256 || !a->gamma || !a->mean || !a->rstd || !a->res)) return 0;
    // row_scale on the bias only (the input arrived pre-scaled): the row-phase kernel's residual epilogue
    if ((a->flags & TGT_EDGE_BIAS_SCALED) && (a->epilogue != EPI_RESID || !er_eligible(*a))) return 0;
    // row_scale on the activation (tgt_gelu_dropout_scaled_fwd's per-sample factor): the row-phase kernel's GELU epilogue only
    if (a->epilogue == EPI_GELU && a->row_scale && !er_eligible(*a)) return 0;
    return 1;
}

int edge_linear_run(const tgt_edge_linear_args* a, hipStream_t st) {
    if (!a || !a->a || !a->w || !a->out) return set_error(TGT_ERR_INVALID, "edge linear: null argument");
    if (a->M < 0 || a->K <= 0 || a->N <= 0) return set_error(TGT_ERR_INVALID, "edge linear: bad sizes");
    if (!edge_linear_supported(a))
        return set_error(TGT_ERR_UNSUPPORTED, "edge linear: unsupported shape/dtype (K=%d N=%d dtype=%d epilogue=%d): needs a 16-bit "
                         "dtype, N %% 8 == 0, K in {64,128,256}; row-wise epilogues N <= 256",
                         a->K, a->N, a->dtype, a->epilogue);
    if (a->M == 0) return TGT_OK;
    static const int ablate = getenv("TGT_EG_ABLATE") ? atoi(getenv("TGT_EG_ABLATE")) : 0;   // kernel_bench probes: 1 no W stream, 2 no stores, 4 no A loads, 8 no MFMA
    tgt_edge_linear_args aa = *a;
    aa._pad0 = ablate;
    a = &aa;
    const uintptr_t al = (uintptr_t)a->a | (uintptr_t)a->w | (uintptr_t)a->out | (uintptr_t)a->out2 | (uintptr_t)a->res |
                         (uintptr_t)a->y | (uintptr_t)a->ds_in;
    if (al % 16 || (a->lda * 2) % 16 || (a->ldw * 2) % 16 || (a->ldo * 2) % 16 || (a->ldo2 * 2) % 16 || (a->ldr * 2) % 16 ||
        (a->ldy * 2) % 16 || (a->ld_ds * 2) % 16)
        return set_error(TGT_ERR_INVALID, "edge linear: tensors and row strides must be 16-byte aligned");
    if ((a->epilogue == EPI_GELU && !a->out2) || ((a->epilogue == EPI_RESID || a->epilogue == EPI_GELU_BWD) && !a->res))
        return set_error(TGT_ERR_INVALID, "edge linear: epilogue operand missing");
    if ((a->row_scale || a->out_scale) && a->rows_per_sample <= 0)
        return set_error(TGT_ERR_INVALID, "edge linear: rows_per_sample missing");
    if (a->gamma && a->epilogue != EPI_LN_BWD && !a->beta) return set_error(TGT_ERR_INVALID, "edge linear: beta missing");
    if (a->dropout_p < 0.f || a->dropout_p >= 1.f) return set_error(TGT_ERR_INVALID, "edge linear: dropout_p outside [0,1)");
    if (er_eligible(*a)) return a->dtype == TGT_BF16 ? er_run<bf16_t>(*a, st) : er_run<f16_t>(*a, st);
    if (es_eligible(*a)) return a->dtype == TGT_BF16 ? es_run<bf16_t>(*a, st) : es_run<f16_t>(*a, st);
    return set_error(TGT_ERR_UNSUPPORTED, "edge linear: no kernel for K=%d N=%d epilogue=%d", a->K, a->N, a->epilogue);
}

}  // namespace tgt
