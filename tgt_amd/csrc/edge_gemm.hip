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
#include "edge_common.hpp"

namespace tgt {

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
__global__ void __launch_bounds__(512, 2) edge_slice_kernel(const tgt_edge_linear_args a, int groups, const uint64_t* seed_ctr) {
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
            if (total % 512 == 0 || p0 + wave * 64 < total)
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
        if (active) {
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
        if constexpr (EPI == EPI_BIAS) {
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
                        if (thresh) keep4(step_seed(a.dropout_seed, seed_ctr), m, N, n0 + 8 * gq + 4 * hi, thresh, keep);
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
                        if (thresh) keep4(step_seed(a.dropout_seed, seed_ctr), m, N, n0 + 8 * gq + 4 * hi, thresh, keep);
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
template <typename T, int KS, int EPI>
__global__ void __launch_bounds__(1024, 4) edge_rows_kernel(const tgt_edge_linear_args a, const uint64_t* seed_ctr) {
    using F = frag_t<T>;
    constexpr int K = KS * 16, N = 256, kBM = 32, kABytes = kBM * K * 2, kSBytes = kBM * N * 2, kPass = 2;
    constexpr bool kOperand = EPI == EPI_RESID || EPI == EPI_GELU_BWD || EPI == EPI_LN_BWD;
    constexpr bool kOp2 = EPI == EPI_LN_BWD;
    constexpr int kOffStage = 2 * kABytes, kOffGB = kOffStage + 2 * kSBytes;      // LDS: A tiles [2] | staging tiles [2] | gamma, beta
    constexpr uint32_t kNone = 0xffffffffu;                                       // a byte offset outside every buffer
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hi = lane >> 5;
    const EgGeo g(K), gs(N);
    const int64_t row_tiles = (a.M + kBM - 1) / kBM;
    float* gb = reinterpret_cast<float*>(smem + kOffGB);
    if (blockIdx.x >= row_tiles) return;
    const int n_tiles = (int)((row_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x);       // tiles of this workgroup
    auto tile_of = [&](int s) { return (int64_t)blockIdx.x + (int64_t)s * gridDim.x; };    // (s >= n_tiles: past the last row -> empty buffers)
    // (static wave priority for either role -- what gave the projection-fused triplet forward 8 % -- measured neutral here:
    // profiles/r05o_ab_prio.txt)

    if (wave < 8) {
        // ------------------------------------------------------------------------------------------------ GEMM role
        const T* W = reinterpret_cast<const T*>(a.w);
        const int n0 = wave * 32;
        constexpr int spr = K >> 3, total = kBM * spr, kNA = (total + 511) / 512;
        F wr[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wr[ks] = load_frag<T>(W + (int64_t)(n0 + r) * a.ldw + ks * 16 + 8 * hi);
        uint2 braw[4];                                     // the bias of this lane's 16 columns, kept PACKED (unpacked per tile into the accumulator)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            braw[gq] = make_uint2(0, 0);
            if (a.bias && !(EPI == EPI_RESID && (a.flags & TGT_EDGE_BIAS_SCALED)))      // (BIAS_SCALED: the row phase adds scale * bias)
                braw[gq] = *reinterpret_cast<const uint2*>(reinterpret_cast<const T*>(a.bias) + n0 + 8 * gq + 4 * hi);
        }
        // this thread's 16-byte pieces (row, slot) of an A tile: byte offset inside the tile's buffer and LDS address
        const int64_t lda_b = a.lda * 2;
        uint32_t aoff[kNA];
        int loff[kNA];
#pragma unroll
        for (int q = 0; q < kNA; ++q) {
            const int pc = q * 512 + tid, row = pc / spr, ps = pc % spr;
            const bool mine = total % 512 == 0 || pc < total;
            aoff[q] = mine ? (uint32_t)row * (uint32_t)lda_b + (uint32_t)ps * 16u : kNone;
            loff[q] = g.off(row % kBM, ps);
        }
        uint4 pre[kNA];
        auto fetch = [&](int64_t tile) {
            const __amdgpu_buffer_rsrc_t rs = tile_rsrc(a.a, lda_b, K * 2, tile * kBM, a.M);
#pragma unroll
            for (int q = 0; q < kNA; ++q) pre[q] = rp_ld16(rs, aoff[q]);
        };
        auto commit = [&](int buf) {
#pragma unroll
            for (int q = 0; q < kNA; ++q)
                if (total % 512 == 0 || aoff[q] != kNone) *reinterpret_cast<uint4*>(smem + buf * kABytes + loff[q]) = pre[q];
        };
        fetch(tile_of(0));
        commit(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int s = 0; s <= n_tiles; ++s) {
            if (s < n_tiles) {
                const char* xs = smem + (s & 1) * kABytes;
                char* sg = smem + kOffStage + (s & 1) * kSBytes;
                fetch(tile_of(s + 1));
                asm volatile("" ::: "memory");            // the prefetch is issued HERE, not sunk towards its use
                f32x16 acc;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {           // accumulator <- bias
                    float bv[4];
                    unpack4<T>(braw[gq], bv);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[4 * gq + j] = bv[j];
                }
                // (the 16-byte slot of k-step ks is XOR-swizzled by the row: left to itself hipcc hoists all KS slot addresses out of the
                // tile loop -- 17 loop-invariant registers at K = 256, which at the 128-register cap it paid for by spilling two weight
                // fragments and RELOADING them from scratch inside this loop, behind a vmcnt(0) that also waited for the prefetch just
                // issued.  The empty asm pins the two-instruction address computation to its k-step.)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    int rr = r;
                    asm volatile("" : "+v"(rr));
                    const F xf = load_frag<T>(reinterpret_cast<const T*>(xs + g.off(rr, 2 * ks + hi)));
                    acc = mma32(wr[ks], xf, acc);
                }
                // accumulators -> staging tile, rounded to the storage type (what nn.Linear emits under autocast);
                // lanes (r, hi) of a row exchange halves so that each holds 8 consecutive columns = one 16-byte slot
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    uint2 lo = pack4<T>(acc[8 * p], acc[8 * p + 1], acc[8 * p + 2], acc[8 * p + 3]);
                    uint2 up = pack4<T>(acc[8 * p + 4], acc[8 * p + 5], acc[8 * p + 6], acc[8 * p + 7]);
                    swap_halves(lo, up);
                    *reinterpret_cast<uint4*>(sg + gs.off(r, (n0 >> 3) + 2 * p + hi)) = make_uint4(lo.x, lo.y, up.x, up.y);
                }
                commit((s + 1) & 1);                       // (the A buffer of tile s-1: its k-loop ended before the last barrier)
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
    const uint32_t thresh = a.dropout_p <= 0.f ? 0u : (uint32_t)fminf(65535.f, fmaxf(1.f, rintf(a.dropout_p * 65536.f)));
    const float inv_keep = a.dropout_p <= 0.f ? 1.f : 1.f / (1.f - a.dropout_p);
    constexpr int kCs = EPI == EPI_LN_BWD ? 4 : 1, kCsX = (EPI == EPI_LN_BWD || EPI == EPI_GELU_BWD) ? 4 : 1;
    f32x2 cs_g[kCs], cs_b[kCs], cs_x[kCsX];              // column sums over every row this workgroup processes (LN_BWD: three
                                                          // planes; GELU_BWD: of the result = the bias gradient of the Linear in front)
#pragma unroll
    for (int j = 0; j < kCs; ++j) cs_g[j] = cs_b[j] = rp_splat(0.f);
#pragma unroll
    for (int j = 0; j < kCsX; ++j) cs_x[j] = rp_splat(0.f);

    // per-graph factor (DropPath): one float per rows_per_sample rows, through a buffer of its own (absent: empty, and the 1.f below)
    const float* scale_ptr = EPI == EPI_GELU_BWD ? a.out_scale : a.row_scale;
    const bool has_scale = scale_ptr != nullptr;
    const FastDiv per_sample((uint32_t)(has_scale ? a.rows_per_sample : 1));      // (host-checked: M < 2^31 when a scale is given)
    const int64_t n_samples = has_scale ? (a.M + a.rows_per_sample - 1) / a.rows_per_sample : 0;
    const __amdgpu_buffer_rsrc_t rs_scale = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(scale_ptr), 0, (int)(n_samples * 4), 0x00020000);
    // this thread's constant byte offsets inside a tile (pass 0; pass 1 is 16 rows further)
    const int64_t ldr_b = a.ldr * 2, ldd_b = a.ld_ds * 2, ldo_b = a.ldo * 2, ldo2_b = a.ldo2 * 2, ldy_b = a.ldy * 2;
    const uint32_t c16 = (uint32_t)ch * 16u;
    const uint32_t o_res = (uint32_t)rsub * (uint32_t)ldr_b + c16, o_ds = (uint32_t)rsub * (uint32_t)ldd_b + c16;
    const uint32_t o_out = (uint32_t)rsub * (uint32_t)ldo_b + c16, o_out2 = (uint32_t)rsub * (uint32_t)ldo2_b + c16;
    const uint32_t o_y = (uint32_t)rsub * (uint32_t)ldy_b + c16;
    const uint32_t o_stat = (uint32_t)rsub * 4u;

    struct Ops { uint4 o1[kOperand ? kPass : 1]; uint4 o2[kOp2 ? kPass : 1]; float mu[kOp2 ? kPass : 1], rs[kOp2 ? kPass : 1], sc[kPass]; };
    auto fetch_ops = [&](int64_t tile, Ops& o) {          // the row phase's operands of `tile`: whole rows, straight from global memory
        const int64_t r0 = tile * kBM;
        const __amdgpu_buffer_rsrc_t rs1 = tile_rsrc(a.res, ldr_b, N * 2, r0, kOperand ? a.M : 0);
        const __amdgpu_buffer_rsrc_t rs2 = tile_rsrc(a.ds_in, ldd_b, N * 2, r0, kOp2 ? a.M : 0);
        const __amdgpu_buffer_rsrc_t rsm = tile_rsrc(a.mean, 4, 4, r0, kOp2 ? a.M : 0), rsr = tile_rsrc(a.rstd, 4, 4, r0, kOp2 ? a.M : 0);
#pragma unroll
        for (int i = 0; i < kPass; ++i) {
            if constexpr (kOperand) o.o1[i] = rp_ld16(rs1, o_res + (uint32_t)i * 16u * (uint32_t)ldr_b);
            if constexpr (kOp2) {
                o.o2[i] = rp_ld16(rs2, o_ds + (uint32_t)i * 16u * (uint32_t)ldd_b);
                o.mu[i] = rp_ld_f32(rsm, o_stat + (uint32_t)i * 64u);
                o.rs[i] = rp_ld_f32(rsr, o_stat + (uint32_t)i * 64u);
            }
            o.sc[i] = rp_ld_f32(rs_scale, per_sample.div((uint32_t)(r0 + i * 16 + rsub)) * 4u);      // (raw: 0 without a scale)
        }
    };
    Ops nxt;
    fetch_ops(tile_of(0), nxt);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    f32x2 gam[4], bet[4];                                 // this thread's 8 columns (4 pairs), for the whole kernel
    f32x2 bsc[4];                                         // EPI_RESID + BIAS_SCALED: the bias, added as row_scale * bias (else 0)
    const bool pres = EPI == EPI_RESID && (a.flags & TGT_EDGE_BIAS_SCALED) != 0;     // x arrived pre-scaled: res + x W^T + scale * bias
#pragma unroll
    for (int k = 0; k < 4; ++k) bsc[k] = rp_splat(0.f);
    if constexpr (EPI == EPI_RESID) {
        if (pres && a.bias) rp_unpack<T>(*reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(a.bias) + ch * 8), bsc);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        gam[k] = *reinterpret_cast<const f32x2*>(gb + ch * 8 + 2 * k);
        bet[k] = *reinterpret_cast<const f32x2*>(gb + N + ch * 8 + 2 * k);
    }
    // vmcnt is in order: the operands of stage s are fetched at the TOP of stage s-1, i.e. they are OLDER than that stage's stores,
    // and the wait in front of their first use leaves exactly those stores in flight -- a full stage to drain under the next one.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // stage 0: the GEMM role fills the first staging tile
    for (int s = 1; s <= n_tiles; ++s) {
        const Ops cur = nxt;                               // operands of tile s-1 (fetched a stage ago)
        fetch_ops(tile_of(s), nxt);
        asm volatile("" ::: "memory");
        const char* sg = smem + kOffStage + ((s - 1) & 1) * kSBytes;
        const int64_t m0 = tile_of(s - 1) * kBM;
        const __amdgpu_buffer_rsrc_t rs_out = tile_rsrc(a.out, ldo_b, N * 2, m0, a.M);
        const __amdgpu_buffer_rsrc_t rs_out2 = tile_rsrc(a.out2, ldo2_b, N * 2, m0, (EPI == EPI_GELU || EPI == EPI_LN_BWD) ? a.M : 0);
        const __amdgpu_buffer_rsrc_t rs_y = tile_rsrc(a.y, ldy_b, N * 2, m0, (EPI == EPI_RESID && a.gamma) ? a.M : 0);
        const __amdgpu_buffer_rsrc_t rs_mean = tile_rsrc(a.mean, 4, 4, m0, (EPI == EPI_RESID && a.gamma) ? a.M : 0);
        const __amdgpu_buffer_rsrc_t rs_rstd = tile_rsrc(a.rstd, 4, 4, m0, (EPI == EPI_RESID && a.gamma) ? a.M : 0);
#pragma unroll
        for (int i = 0; i < kPass; ++i) {
            const int row = i * 16 + rsub;
            const int64_t m = m0 + row;
            const uint32_t po = (uint32_t)i * 16u;          // rows of this pass below the thread's pass-0 row
            f32x2 v[4];
            rp_unpack<T>(*reinterpret_cast<const uint4*>(sg + gs.off(row, ch)), v);
            const float sc_i = has_scale ? cur.sc[i] : 1.f;
            if constexpr (EPI == EPI_GELU) {
                rp_st16(rs_out2, o_out2 + po * (uint32_t)ldo2_b, rp_pack<T>(v));                  // the pre-activation
                bool keep[8] = {true, true, true, true, true, true, true, true};
                if (thresh) keep_vector<8>(step_seed(a.dropout_seed, seed_ctr), (m * N + ch * 8) >> 3, thresh, keep);
                f32x2 gl[4];
                const float ik = inv_keep * sc_i;       // (row_scale: the DropPath factor of the branch, folded into the activation)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    f32x2 e;
                    const f32x2 cdf = gelu_cdf2(v[k], e);
                    const f32x2 y = v[k] * cdf * ik;
                    gl[k].x = keep[2 * k] ? y.x : 0.f;
                    gl[k].y = keep[2 * k + 1] ? y.y : 0.f;
                }
                rp_st16(rs_out, o_out + po * (uint32_t)ldo_b, rp_pack<T>(gl));
            } else if constexpr (EPI == EPI_GELU_BWD) {
                const float al = sc_i;
                f32x2 pv[4], o[4];
                rp_unpack<T>(cur.o1[i], pv);
                bool keep[8] = {true, true, true, true, true, true, true, true};
                if (thresh) keep_vector<8>(step_seed(a.dropout_seed, seed_ctr), (m * N + ch * 8) >> 3, thresh, keep);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    f32x2 e;
                    const f32x2 cdf = gelu_cdf2(pv[k], e);
                    const f32x2 dy = has_scale ? rp_round<T>(v[k] * al) : v[k];
                    const f32x2 y = dy * (cdf + pv[k] * 0.3989422804014327f * e) * inv_keep;
                    o[k].x = keep[2 * k] ? y.x : 0.f;
                    o[k].y = keep[2 * k + 1] ? y.y : 0.f;
                    cs_x[k] += rp_round<T>(o[k]);          // (of the values AS STORED: equal to a separate pass over the result)
                }
                rp_st16(rs_out, o_out + po * (uint32_t)ldo_b, rp_pack<T>(o));
            } else if constexpr (EPI == EPI_RESID) {
                // t = res + s1 * z + s2 * bias:  (s1, s2) = (scale, 0) plain [z carries the bias], (1, scale) when z arrived pre-scaled
                const float s1 = pres ? 1.f : sc_i, s2 = pres ? sc_i : 0.f;
                f32x2 rv[4], t[4];
                rp_unpack<T>(cur.o1[i], rv);
                f32x2 p1 = rp_splat(0.f);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    t[k] = rp_round<T>(bsc[k] * s2 + (v[k] * s1 + rv[k]));      // the stream value as stored: LayerNorm sees that
                    p1 += t[k];
                }
                rp_st16(rs_out, o_out + po * (uint32_t)ldo_b, rp_pack<T>(t));
                // LayerNorm of the new row (y, mean, rstd: empty buffers when not asked for -- the arithmetic is the same few instructions)
                const float mean = rp_row_sum(p1.x + p1.y) * (1.f / N);
                f32x2 p2 = rp_splat(0.f);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    t[k] -= mean;
                    p2 += t[k] * t[k];
                }
                const float rstd = rsqrtf(rp_row_sum(p2.x + p2.y) * (1.f / N) + a.eps);
#pragma unroll
                for (int k = 0; k < 4; ++k) t[k] = t[k] * rstd * gam[k] + bet[k];
                rp_st16(rs_y, o_y + po * (uint32_t)ldy_b, rp_pack<T>(t));
                const uint32_t so = ch == 0 ? o_stat + po * 4u : kNone;      // (one lane of the row stores its statistics)
                rp_st_f32(rs_mean, so, mean);
                rp_st_f32(rs_rstd, so, rstd);
            } else if constexpr (EPI == EPI_LN_BWD) {
                // v = dy at the output of LayerNorm(res; gamma); d_res = rstd (g - mean(g) - xhat mean(g xhat)) + ds_in, g = dy gamma
                // (rows at or past M: A, res, mean, rstd, ds all read as 0 and there is no bias, so they add nothing to the column sums)
                const float mu = cur.mu[i], rs = cur.rs[i], sc = sc_i;
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
                }
                const float c1 = rp_row_sum(p1.x + p1.y) * (1.f / N), c2 = rp_row_sum(p2.x + p2.y) * (1.f / N);
                f32x2 d[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) d[k] = (gg[k] - c1 - xh[k] * c2) * rs + ds[k];
                rp_st16(rs_out, o_out + po * (uint32_t)ldo_b, rp_pack<T>(d));
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    // the x-branch gradient as it is stored (rounded), so that its column sums equal a separate pass's
                    d[k] = rp_round<T>(rp_round<T>(d[k]) * sc);
                    cs_x[k] += d[k];
                }
                rp_st16(rs_out2, o_out2 + po * (uint32_t)ldo2_b, rp_pack<T>(d));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    if constexpr (EPI == EPI_LN_BWD || EPI == EPI_GELU_BWD) {
        if (a.colsum_partial) {
            // fold the 16 row groups, fixed order: [16][planes][256] floats in LDS (the A / staging tiles are dead: every wave is past
            // the last barrier; only the row role takes part from here on -- the GEMM waves have exited, and an exited wave
            // counts as arrived at a barrier).  Planes: LN_BWD dgamma | dbeta | sum of the scaled x-gradient; GELU_BWD the result.
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
            for (int c = t4; c < kPl * N; c += 512) {
                float t = 0.f;
#pragma unroll
                for (int s_ = 0; s_ < 16; ++s_) t += red[s_ * kPl * N + c];
                part[c] = t;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// K = 512, N = 256: lin_O of the triplet modules (reference triplet.py:248-249) + DropPath + residual add + the LayerNorm that
// opens the edge FFN (layers.py:284-290), ONE launch instead of library GEMM -> z -> add + LayerNorm pass (the 134 MB z is
// neither written nor read back).
//
// The 256 x 512 weight is 256 KB: resident in registers only when ALL 16 waves of the workgroup hold a piece (32 columns x 256 k
// = 64 registers each) -- there is no room for a separate row role.  So the roles of the K <= 256 kernel become PHASES of every
// wave: k-loop on its (column block, K half), fp32 partial tile to LDS, barrier, row phase (thread = row x 16-byte chunk: the two
// K halves are added, rounded to the storage type, residual + LayerNorm as in the K <= 256 kernel), barrier.  The matrix pipe and
// the row arithmetic no longer overlap, but with straight-line phases and exact wait counts (buffer-resource addressing, see
// above) the memory traffic does: the next tile's A rows and residual rows are in flight from the top of the stage, this tile's
// stores drain under the next stage -- and the two phases together (~5500 cycles per 32-row tile) stay under the tile's HBM time
// (80 KB per tile and CU: ~9800 cycles at 5 TB/s).
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(1024, 4) edge_rows512_kernel(const tgt_edge_linear_args a) {
    using F = frag_t<T>;
    constexpr int K = 512, KS = 16, N = 256, kBM = 32;
    constexpr int kABytes = kBM * K * 2, kPBytes = kBM * N * 4;
    constexpr int kOffP = 2 * kABytes, kOffGB = kOffP + 2 * kPBytes;         // LDS: A tiles [2] | fp32 partial tiles [2 K halves] | gamma, beta, bias
    constexpr uint32_t kNone = 0xffffffffu;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hi = lane >> 5;
    const int kh = wave >> 3, n0 = (wave & 7) * 32;       // K half and column block of this wave's GEMM phase
    const int row = tid >> 5, ch = tid & 31;              // row and 16-byte chunk (8 columns) of this thread's row phase
    const EgGeo g(K);
    const int64_t row_tiles = (a.M + kBM - 1) / kBM;
    float* gb = reinterpret_cast<float*>(smem + kOffGB);
    if (blockIdx.x >= row_tiles) return;
    const int n_tiles = (int)((row_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x);
    auto tile_of = [&](int s) { return (int64_t)blockIdx.x + (int64_t)s * gridDim.x; };
    const bool pres = (a.flags & TGT_EDGE_BIAS_SCALED) != 0;                  // a arrived pre-scaled: res + a W^T + scale * bias
    if (tid < N) {
        gb[tid] = a.gamma ? a.gamma[tid] : 1.f;
        gb[N + tid] = (a.gamma && a.beta) ? a.beta[tid] : 0.f;
        gb[2 * N + tid] = a.bias ? to_f32(reinterpret_cast<const T*>(a.bias)[tid]) : 0.f;
    }
    const T* W = reinterpret_cast<const T*>(a.w);
    F wr[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wr[ks] = load_frag<T>(W + (int64_t)(n0 + r) * a.ldw + kh * 256 + ks * 16 + 8 * hi);
    // this thread's two 16-byte pieces of an A tile (32 rows x 64 slots)
    const int64_t lda_b = a.lda * 2;
    uint32_t aoff[2];
    int loff[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int pc = q * 1024 + tid, prow = pc >> 6, ps = pc & 63;
        aoff[q] = (uint32_t)prow * (uint32_t)lda_b + (uint32_t)ps * 16u;
        loff[q] = g.off(prow, ps);
    }
    const float* scale_ptr = a.row_scale;
    const bool has_scale = scale_ptr != nullptr;
    const FastDiv per_sample((uint32_t)(has_scale ? a.rows_per_sample : 1));
    const int64_t n_samples = has_scale ? (a.M + a.rows_per_sample - 1) / a.rows_per_sample : 0;
    const __amdgpu_buffer_rsrc_t rs_scale = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(scale_ptr), 0, (int)(n_samples * 4), 0x00020000);
    const int64_t ldr_b = a.ldr * 2, ldo_b = a.ldo * 2, ldy_b = a.ldy * 2;
    const uint32_t c16 = (uint32_t)ch * 16u;
    const uint32_t o_res = (uint32_t)row * (uint32_t)ldr_b + c16, o_out = (uint32_t)row * (uint32_t)ldo_b + c16;
    const uint32_t o_y = (uint32_t)row * (uint32_t)ldy_b + c16, o_stat = ch == 0 ? (uint32_t)row * 4u : kNone;
    // the partial tile as 16-byte slots (4 floats), XOR-swizzled by the row: conflict-free for the accumulator writes
    auto poff = [&](int prow, int slot) { return prow * 1024 + ((slot ^ (prow & 15)) << 4); };

    // Register budget: 64 (weights) + 16 (accumulator) + 8 (A prefetch) at the 128-register cap leaves room for ONE set of row
    // operands: the A rows of tile s+1 are requested at the top of stage s (a whole stage ahead), the residual rows of tile s+1
    // right after stage s's stores (one GEMM phase ahead).
    uint4 pre[2];
    uint4 op_res;
    float op_sc;
    auto fetch_a = [&](int64_t tile) {
        const __amdgpu_buffer_rsrc_t rsa = tile_rsrc(a.a, lda_b, K * 2, tile * kBM, a.M);
        pre[0] = rp_ld16(rsa, aoff[0]);
        pre[1] = rp_ld16(rsa, aoff[1]);
    };
    auto fetch_ops = [&](int64_t tile) {
        const int64_t r0 = tile * kBM;
        const __amdgpu_buffer_rsrc_t rs1 = tile_rsrc(a.res, ldr_b, N * 2, r0, a.M);
        op_res = rp_ld16(rs1, o_res);
        op_sc = rp_ld_f32(rs_scale, per_sample.div((uint32_t)(r0 + row)) * 4u);      // (raw: 0 without a scale; selected where it is used)
    };
    auto commit = [&](int buf) {
        *reinterpret_cast<uint4*>(smem + buf * kABytes + loff[0]) = pre[0];
        *reinterpret_cast<uint4*>(smem + buf * kABytes + loff[1]) = pre[1];
    };
    fetch_a(tile_of(0));
    fetch_ops(tile_of(0));
    commit(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    for (int s = 0; s < n_tiles; ++s) {
        fetch_a(tile_of(s + 1));                           // (past the last tile: empty buffers)
        asm volatile("" ::: "memory");
        // ------------------------------------------------------------------------------------------- GEMM phase
        const char* xs = smem + (s & 1) * kABytes;
        f32x16 acc;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {                   // K half 0 starts from the bias (unless the row phase adds scale * bias)
            const float4 bv = *reinterpret_cast<const float4*>(gb + 2 * N + n0 + 8 * gq + 4 * hi);
            const bool use = kh == 0 && !pres;
            acc[4 * gq] = use ? bv.x : 0.f; acc[4 * gq + 1] = use ? bv.y : 0.f; acc[4 * gq + 2] = use ? bv.z : 0.f; acc[4 * gq + 3] = use ? bv.w : 0.f;
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            int rr = r;
            asm volatile("" : "+v"(rr));                   // (keeps the swizzled address arithmetic at its k-step: see edge_rows_kernel)
            const F xf = load_frag<T>(reinterpret_cast<const T*>(xs + g.off(rr, kh * 32 + 2 * ks + hi)));
            acc = mma32(wr[ks], xf, acc);
        }
        {
            char* pt = smem + kOffP + kh * kPBytes;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                *reinterpret_cast<float4*>(pt + poff(r, (n0 >> 2) + 2 * gq + hi)) = make_float4(acc[4 * gq], acc[4 * gq + 1], acc[4 * gq + 2], acc[4 * gq + 3]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // -------------------------------------------------------------------------------------------- row phase
        {
            const int64_t m0 = tile_of(s) * kBM;
            const __amdgpu_buffer_rsrc_t rs_out = tile_rsrc(a.out, ldo_b, N * 2, m0, a.M);
            const __amdgpu_buffer_rsrc_t rs_y = tile_rsrc(a.y, ldy_b, N * 2, m0, a.gamma ? a.M : 0);
            const __amdgpu_buffer_rsrc_t rs_mean = tile_rsrc(a.mean, 4, 4, m0, a.gamma ? a.M : 0);
            const __amdgpu_buffer_rsrc_t rs_rstd = tile_rsrc(a.rstd, 4, 4, m0, a.gamma ? a.M : 0);
            const char* p0 = smem + kOffP;
            f32x2 v[4];
#pragma unroll
            for (int h = 0; h < 2; ++h) {                  // columns 8 ch + 4 h .. + 3: slot 2 ch + h of both K halves
                const float4 x0 = *reinterpret_cast<const float4*>(p0 + poff(row, 2 * ch + h));
                const float4 x1 = *reinterpret_cast<const float4*>(p0 + kPBytes + poff(row, 2 * ch + h));
                v[2 * h].x = x0.x + x1.x; v[2 * h].y = x0.y + x1.y;
                v[2 * h + 1].x = x0.z + x1.z; v[2 * h + 1].y = x0.w + x1.w;
            }
            const float sc = has_scale ? op_sc : 1.f;
            const float s1 = pres ? 1.f : sc, s2 = pres ? sc : 0.f;
            f32x2 rv[4], t[4];
            rp_unpack<T>(op_res, rv);
            f32x2 p1 = rp_splat(0.f);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x2 z = rp_round<T>(v[k]);          // the Linear's output as stored (what nn.Linear emits under autocast)
                const f32x2 bk = *reinterpret_cast<const f32x2*>(gb + 2 * N + ch * 8 + 2 * k);
                t[k] = rp_round<T>(bk * s2 + (z * s1 + rv[k]));
                p1 += t[k];
            }
            rp_st16(rs_out, o_out, rp_pack<T>(t));
            const float mean = rp_row_sum(p1.x + p1.y) * (1.f / N);
            f32x2 p2 = rp_splat(0.f);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                t[k] -= mean;
                p2 += t[k] * t[k];
            }
            const float rstd = rsqrtf(rp_row_sum(p2.x + p2.y) * (1.f / N) + a.eps);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x2 gk = *reinterpret_cast<const f32x2*>(gb + ch * 8 + 2 * k), be = *reinterpret_cast<const f32x2*>(gb + N + ch * 8 + 2 * k);
                t[k] = t[k] * rstd * gk + be;
            }
            rp_st16(rs_y, o_y, rp_pack<T>(t));
            rp_st_f32(rs_mean, o_stat, mean);
            rp_st_f32(rs_rstd, o_stat, rstd);
        }
        fetch_ops(tile_of(s + 1));
        asm volatile("" ::: "memory");
        commit((s + 1) & 1);                               // (the A buffer of tile s-1; its k-loop ended two barriers ago)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// K = 256, N = 512, plain (bias) epilogue: the DATA GRADIENT of lin_O (d_z (M,256) x W_O (256,512) -> dVa (M,512), reference
// triplet.py:248-249 under autograd) -- and any 256 -> 512 Linear on the edge rows.  The library runs it at 3.5 TB/s on its
// 3 E of traffic (0.115 ms); this is the mirror image of the K = 512 kernel above: all 16 waves hold 32 output columns x 256 k
// (64 registers of weights), one k-loop per 32-row tile, the accumulators go through ONE storage-type tile in LDS
// (32 rows x 1 KB, 16-byte slots XOR-swizzled by the row) and leave as whole 1 KB rows, 16 bytes per thread and store.
// Straight-line stages on tile-based buffer resources, exact wait counts (see edge_rows_kernel).
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(1024, 4) edge_wide512_kernel(const tgt_edge_linear_args a) {
    using F = frag_t<T>;
    constexpr int K = 256, KS = 16, N = 512, kBM = 32;
    constexpr int kABytes = kBM * K * 2, kOBytes = kBM * N * 2;
    constexpr int kOffO = 2 * kABytes, kOffB = kOffO + kOBytes;               // LDS: A tiles [2] | output tile | bias (fp32)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hi = lane >> 5;
    const int n0 = wave * 32;                             // column block of this wave's k-loop
    const int row = tid >> 5, ch = tid & 31;              // row and 16-byte chunk of this thread's loads / stores
    const EgGeo g(K);
    const int64_t row_tiles = (a.M + kBM - 1) / kBM;
    float* bs = reinterpret_cast<float*>(smem + kOffB);
    if (blockIdx.x >= row_tiles) return;
    const int n_tiles = (int)((row_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x);
    auto tile_of = [&](int s) { return (int64_t)blockIdx.x + (int64_t)s * gridDim.x; };
    if (tid < N) bs[tid] = a.bias ? to_f32(reinterpret_cast<const T*>(a.bias)[tid]) : 0.f;
    const T* W = reinterpret_cast<const T*>(a.w);
    F wr[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wr[ks] = load_frag<T>(W + (int64_t)(n0 + r) * a.ldw + ks * 16 + 8 * hi);
    const int64_t lda_b = a.lda * 2, ldo_b = a.ldo * 2;
    const uint32_t aoff = (uint32_t)row * (uint32_t)lda_b + (uint32_t)ch * 16u;
    const int loff = g.off(row, ch);
    const uint32_t o_out = (uint32_t)row * (uint32_t)ldo_b + (uint32_t)ch * 16u;
    auto ooff = [&](int prow, int slot) { return prow * 1024 + ((slot ^ (prow & 31)) << 4); };
    uint4 pre;
    auto fetch_a = [&](int64_t tile) { pre = rp_ld16(tile_rsrc(a.a, lda_b, K * 2, tile * kBM, a.M), aoff); };
    auto commit = [&](int buf) { *reinterpret_cast<uint4*>(smem + buf * kABytes + loff) = pre; };
    fetch_a(tile_of(0));
    commit(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    char* ot = smem + kOffO;
    for (int s = 0; s < n_tiles; ++s) {
        fetch_a(tile_of(s + 1));                           // (past the last tile: an empty buffer)
        asm volatile("" ::: "memory");
        const char* xs = smem + (s & 1) * kABytes;
        f32x16 acc;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const float4 bv = *reinterpret_cast<const float4*>(bs + n0 + 8 * gq + 4 * hi);
            acc[4 * gq] = bv.x; acc[4 * gq + 1] = bv.y; acc[4 * gq + 2] = bv.z; acc[4 * gq + 3] = bv.w;
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            int rr = r;
            asm volatile("" : "+v"(rr));                   // (keeps the swizzled address arithmetic at its k-step: see edge_rows_kernel)
            const F xf = load_frag<T>(reinterpret_cast<const T*>(xs + g.off(rr, 2 * ks + hi)));
            acc = mma32(wr[ks], xf, acc);
        }
        // accumulator element 4 gq + e = (tile row r, column n0 + 8 gq + 4 hi + e): 8 bytes of slot n0/8 + gq
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            f32x2 lo = {acc[4 * gq], acc[4 * gq + 1]}, hi2 = {acc[4 * gq + 2], acc[4 * gq + 3]};
            *reinterpret_cast<uint2*>(ot + ooff(r, (n0 >> 3) + gq) + 8 * hi) = make_uint2(rp_pack2<T>(lo), rp_pack2<T>(hi2));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        {
            const __amdgpu_buffer_rsrc_t rs_out = tile_rsrc(a.out, ldo_b, N * 2, tile_of(s) * kBM, a.M);
            const uint4 v0 = *reinterpret_cast<const uint4*>(ot + ooff(row, ch));
            const uint4 v1 = *reinterpret_cast<const uint4*>(ot + ooff(row, ch + 32));
            rp_st16(rs_out, o_out, v0);
            rp_st16(rs_out, o_out + 512u, v1);
        }
        commit((s + 1) & 1);                               // (the A buffer of tile s-1; its k-loop ended two barriers ago)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

bool edge_wgrad_eligible(const tgt_edge_linear_args& a);                       // edge_wgrad.hip
int edge_wgrad_run(const tgt_edge_linear_args& a, int grid, hipStream_t st);

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
// the grid of a row-phase launch: the caller's row count of colsum_partial when it states one (one row per persistent workgroup:
// exactly the provided rows are written, whatever device / grid cap the sizing call saw), else the default
static int er_grid(const tgt_edge_linear_args& a) {
    const int64_t row_tiles = (a.M + 31) / 32;
    if ((a.colsum_partial || a.dw_partial) && a.colsum_rows > 0) return (int)(a.colsum_rows < row_tiles ? a.colsum_rows : row_tiles);
    return er_grid(a.M);
}


template <typename T, int KS, int EPI>
static int er_launch(const tgt_edge_linear_args& a, hipStream_t st) {
    constexpr int K = KS * 16;
    constexpr int lds0 = 2 * 32 * K * 2 + 2 * 32 * 256 * 2 + 2 * 256 * 4;
    constexpr int lds = lds0 < 49152 ? 49152 : lds0;                        // the final column-sum fold of LN_BWD needs 48 KB
    static bool attr_set[16] = {};
    if (!dyn_lds_once(attr_set, reinterpret_cast<const void*>(&edge_rows_kernel<T, KS, EPI>), lds))
        return set_error(TGT_ERR_LAUNCH, "edge_rows_kernel: cannot reserve %d bytes of LDS", lds);
    hipLaunchKernelGGL((edge_rows_kernel<T, KS, EPI>), dim3((unsigned)er_grid(a)), dim3(1024), lds, st, a, seed_counter());
    return check_launch("edge_rows_kernel");
}

template <typename T>
static int er512_launch(const tgt_edge_linear_args& a, hipStream_t st) {
    constexpr int lds = 2 * 32 * 512 * 2 + 2 * 32 * 256 * 4 + 3 * 256 * 4;
    static bool attr_set[16] = {};
    if (!dyn_lds_once(attr_set, reinterpret_cast<const void*>(&edge_rows512_kernel<T>), lds))
        return set_error(TGT_ERR_LAUNCH, "edge_rows512_kernel: cannot reserve %d bytes of LDS", lds);
    hipLaunchKernelGGL((edge_rows512_kernel<T>), dim3((unsigned)er_grid(a)), dim3(1024), lds, st, a);
    return check_launch("edge_rows512_kernel");
}

template <typename T>
static int ew512_launch(const tgt_edge_linear_args& a, hipStream_t st) {
    constexpr int lds = 2 * 32 * 256 * 2 + 32 * 512 * 2 + 512 * 4;
    static bool attr_set[16] = {};
    if (!dyn_lds_once(attr_set, reinterpret_cast<const void*>(&edge_wide512_kernel<T>), lds))
        return set_error(TGT_ERR_LAUNCH, "edge_wide512_kernel: cannot reserve %d bytes of LDS", lds);
    hipLaunchKernelGGL((edge_wide512_kernel<T>), dim3((unsigned)er_grid(a)), dim3(1024), lds, st, a);
    return check_launch("edge_wide512_kernel");
}
// K = 256 -> N = 512 with the plain epilogue and nothing else attached
static bool ew512_eligible(const tgt_edge_linear_args& a) {
    return a.K == 256 && a.N == 512 && a.epilogue == EPI_BIAS && !a.gamma && !a.row_scale && !a.out_scale;
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
    hipLaunchKernelGGL((edge_slice_kernel<T, KS, WN, EPI, RB>), dim3((unsigned)blocks), dim3(512), lds, st, a, (int)groups, seed_counter());
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

// rows of colsum_partial the caller provides: N = 256 runs on the row-phase kernel, which writes ONE row per persistent workgroup
// (every workgroup of the launch writes its row: no zero fill needed); narrower outputs run on the slice kernel: one row per
// (128-row tile, row group of waves WM = 8 / WN), ZERO-FILLED by the caller
int edge_linear_parts(int64_t M, int N) {
    if (N == 256) return er_grid(M);
    const int wn = N <= 64 ? 2 : (N <= 128 ? 4 : 8);
    return (int)((M + 127) / 128) * (8 / wn);
}

int edge_linear_supported(const tgt_edge_linear_args* a) {
    if (!a) return 0;
    if (a->dtype != TGT_BF16 && a->dtype != TGT_F16) return 0;
    const int K = a->K, N = a->N;
    if (K == 512) return (N == 256 && a->epilogue == EPI_RESID && (!a->gamma || (a->beta && a->y))) ? 1 : 0;      // lin_O + residual [+ LayerNorm]
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
                         "dtype, N %% 8 == 0, K in {64,128,256} (K = 512: residual epilogue, N = 256); row-wise epilogues N <= 256",
                         a->K, a->N, a->dtype, a->epilogue);
    if (a->M == 0) return TGT_OK;
    const uintptr_t al = (uintptr_t)a->a | (uintptr_t)a->w | (uintptr_t)a->out | (uintptr_t)a->out2 | (uintptr_t)a->res |
                         (uintptr_t)a->y | (uintptr_t)a->ds_in;
    if (al % 16 || (a->lda * 2) % 16 || (a->ldw * 2) % 16 || (a->ldo * 2) % 16 || (a->ldo2 * 2) % 16 || (a->ldr * 2) % 16 ||
        (a->ldy * 2) % 16 || (a->ld_ds * 2) % 16)
        return set_error(TGT_ERR_INVALID, "edge linear: tensors and row strides must be 16-byte aligned");
    if ((a->epilogue == EPI_GELU && !a->out2) || ((a->epilogue == EPI_RESID || a->epilogue == EPI_GELU_BWD) && !a->res))
        return set_error(TGT_ERR_INVALID, "edge linear: epilogue operand missing");
    if ((a->epilogue == EPI_GELU_BWD || a->epilogue == EPI_LN_BWD) && a->bias)
        return set_error(TGT_ERR_INVALID, "edge linear: the backward epilogues are data-gradient GEMMs, they take no bias");
    if ((a->row_scale || a->out_scale) && (a->rows_per_sample <= 0 || a->rows_per_sample > 0x7fffffffLL || a->M > 0x7fffffffLL))
        return set_error(TGT_ERR_INVALID, "edge linear: rows_per_sample missing (or more than 2^31 rows with a per-sample scale)");
    if (a->gamma && a->epilogue != EPI_LN_BWD && !a->beta) return set_error(TGT_ERR_INVALID, "edge linear: beta missing");
    if (a->dropout_p < 0.f || a->dropout_p >= 1.f) return set_error(TGT_ERR_INVALID, "edge linear: dropout_p outside [0,1)");
    if ((a->colsum_partial || a->dw_partial) && a->N == 256 && (a->colsum_rows < 0 || a->colsum_rows > (a->M + 31) / 32))
        return set_error(TGT_ERR_INVALID, "edge linear: colsum_rows = %d, but a launch over %lld rows writes at most %lld rows of colsum_partial "
                         "(size the buffer with tgt_edge_linear_parts)", a->colsum_rows, (long long)a->M, (long long)((a->M + 31) / 32));
    if (a->dw_partial) {
        if (!edge_wgrad_eligible(*a))
            return set_error(TGT_ERR_UNSUPPORTED, "edge linear: dw_partial (fused weight gradient) needs K = N = 256, a 16-bit dtype and "
                             "TGT_EPI_GELU_BWD, or TGT_EPI_LN_BWD with gamma AND beta (K=%d N=%d epilogue=%d)", a->K, a->N, a->epilogue);
        if (((uintptr_t)a->dw_partial) % 16) return set_error(TGT_ERR_INVALID, "edge linear: dw_partial must be 16-byte aligned");
        return edge_wgrad_run(*a, er_grid(*a), st);
    }
    if (a->K == 512) return a->dtype == TGT_BF16 ? er512_launch<bf16_t>(*a, st) : er512_launch<f16_t>(*a, st);
    if (ew512_eligible(*a)) return a->dtype == TGT_BF16 ? ew512_launch<bf16_t>(*a, st) : ew512_launch<f16_t>(*a, st);
    if (er_eligible(*a)) return a->dtype == TGT_BF16 ? er_run<bf16_t>(*a, st) : er_run<f16_t>(*a, st);
    if (es_eligible(*a)) return a->dtype == TGT_BF16 ? es_run<bf16_t>(*a, st) : es_run<f16_t>(*a, st);
    return set_error(TGT_ERR_UNSUPPORTED, "edge linear: no kernel for K=%d N=%d epilogue=%d", a->K, a->N, a->epilogue);
}

}  // namespace tgt
