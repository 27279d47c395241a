// Row-tile GEMMs of the edge channel with the neighbouring passes fused in -- gfx950.
//
// The reference's edge channel is a chain of nn.LayerNorm -> nn.Linear -> (activation) -> nn.Linear ->
// residual `add_` on the (B*N*N, 256) edge rows (lib/tgt/layers/layers.py:37-38,:62-80,:155-160,:262-294;
// lib/tgt/layers/triplet.py:207-211,:229-230,:248-249).  With K = 256 every one of these Linears is
// HBM-bound (arithmetic intensity K*N/(K+N) <= 128 FLOP/B against a ridge of ~310), so what matters is how
// often the 134 MB edge tensor crosses HBM, not MFMA utilisation.  This kernel is one GEMM
//     out[M, N] = epilogue( prologue(A[M, K]) . W[N, K]^T + bias )
// whose prologue / epilogue absorb the passes around the library GEMM it replaces:
//   prologue : LayerNorm over K in LDS (mean / rstd saved; the normalised rows optionally written out, the
//              weight gradient still needs them)
//   epilogue : EPI_BIAS  plain                                   (lin_EG, the fused triplet projection, dgrad)
//              EPI_GELU  pre-activation + dropout(gelu(.))       (lin_W1 of the FFN)
//              EPI_RESID res + DropPath-scale[graph] * (.)       (lin_O_e, lin_O, lin_W2: the result IS the new stream)
//              EPI_GELU_BWD   (.) * gelu'(pre) * keep / (1-p)    (data gradient through lin_W2 and the activation)
//              EPI_LN_BWD     LayerNorm backward of the result + the gradient arriving on the residual stream,
//                             dgamma / dbeta / bias-gradient column sums as per-tile partials
// so that the standalone LayerNorm / residual / GELU sweeps over the edge tensor disappear.
//
// Mapping.  Workgroup = 128 rows x up to 256 output columns, 4 waves; two workgroups per CU (64 KB of LDS,
// <= 256 VGPRs) overlap each other's load / MFMA / store phases -- no software pipeline across tiles.
// The A tile (128 rows x <= 256 k) sits in LDS, 16-byte slots XOR-swizzled by the row so that the 16 lanes
// of a ds_read_b128 group hit 16 different bank groups; K > 256 is walked in 256-wide chunks with the
// accumulators kept (only when the output has a single column tile).  Every wave owns ALL 128 rows x a
// 64-column (NB = 2) or 32-column (NB = 1) slice: a weight element is fetched once per workgroup, straight
// from L2 into registers (the weight is <= 0.8 MB and shared by all 2048 workgroups), never through LDS,
// so the k-loop has no barrier.  v_mfma_f32_32x32x16 with the WEIGHT rows as the A operand and the
// activation rows as the B operand: the result is transposed (lane = row, registers = columns), which makes
// row reductions (LayerNorm backward) in-lane, and leaves 4 consecutive columns per register quad: a
// v_permlane32_swap pairs two quads into one 16-byte store / load per lane.
#include "common.hpp"

namespace tgt {

enum { EPI_BIAS = 0, EPI_GELU = 1, EPI_RESID = 2, EPI_GELU_BWD = 3, EPI_LN_BWD = 4 };

constexpr int kKC = 256;          // k-chunk held in LDS

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

// MB = 32-row blocks per wave (rows per workgroup kBM = 32*MB): 4, or 2 for the register-hungry LN_BWD epilogue
template <typename T, int MB, int NB, int EPI, bool LN>
__global__ void __launch_bounds__(256, 2) edge_linear_kernel(const tgt_edge_linear_args a) {
    using F = frag_t<T>;
    constexpr int kBM = 32 * MB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hi = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * kBM;
    const int K = a.K, N = a.N;
    const int kc_max = K < kKC ? K : kKC;
    const EgGeo geo(kc_max);
    char* xs = smem;
    float* st_mean = reinterpret_cast<float*>(smem + kBM * geo.rowbytes);
    float* st_rstd = st_mean + kBM;
    float* red = st_rstd + kBM;                         // EPI_LN_BWD: [4 waves][128 rows][2]
    const T* A = reinterpret_cast<const T*>(a.a);
    const T* W = reinterpret_cast<const T*>(a.w);
    constexpr int NT = 4 * NB * 32;
    const int n_tiles = (N + NT - 1) / NT, chunks = (K + kKC - 1) / kKC;

    auto stage = [&](int kc0, int kcl) {
        const int spr = kcl >> 3, total = kBM * spr;       // 16-byte pieces: a multiple of 64 (kBM >= 64, spr >= 2 ... 256-thread strides)
        for (int p0 = tid; p0 < total; p0 += 256 * 4) {
            uint4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int p = p0 + 256 * i;
                const int row = p / spr, slot = p - row * spr;
                v[i] = make_uint4(0, 0, 0, 0);
                if (p < total && m0 + row < a.M) v[i] = *reinterpret_cast<const uint4*>(A + (m0 + row) * a.lda + kc0 + slot * 8);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int p = p0 + 256 * i;
                const int row = p / spr, slot = p - row * spr;
                if (p < total) *reinterpret_cast<uint4*>(xs + geo.off(row, slot)) = v[i];
            }
        }
    };

    f32x16 acc[MB][NB];
    auto init_acc = [&](int n0) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            float bv[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int n = n0 + nb * 32 + acc_row(q, hi);
                bv[q] = (a.bias && n < N) ? to_f32(reinterpret_cast<const T*>(a.bias)[n]) : 0.f;
            }
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[mb][nb][q] = bv[q];
        }
    };
    auto wfrag = [&](int n, int k) -> F {
        return n < N ? load_frag<T>(W + (int64_t)n * a.ldw + k) : zero_frag<T>();
    };
    auto kloop = [&](int n0, int kc0, int kcl) {
        if (n0 >= N) return;                             // whole slice past the last column (wave-uniform)
        const int nks = kcl >> 4;
        F wc[NB], wn[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) wc[nb] = wfrag(n0 + nb * 32 + r, kc0 + 8 * hi);
        for (int ks = 0; ks < nks; ++ks) {
            if (ks + 1 < nks) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) wn[nb] = wfrag(n0 + nb * 32 + r, kc0 + (ks + 1) * 16 + 8 * hi);
            }
            F xf[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) xf[mb] = load_frag<T>(reinterpret_cast<const T*>(xs + geo.off(mb * 32 + r, 2 * ks + hi)));
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = mma32(wc[nb], xf[mb], acc[mb][nb]);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) wc[nb] = wn[nb];
        }
    };

    const uint32_t thresh = a.dropout_p <= 0.f ? 0u : (uint32_t)fminf(65535.f, fmaxf(1.f, rintf(a.dropout_p * 65536.f)));
    const float inv_keep = a.dropout_p <= 0.f ? 1.f : 1.f / (1.f - a.dropout_p);

    auto epilogue = [&](int n0) {
        if (n0 >= N && EPI != EPI_LN_BWD) return;
        T* out = reinterpret_cast<T*>(a.out);
        if constexpr (EPI == EPI_BIAS) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int64_t m = m0 + mb * 32 + r;
                const float al = a.out_scale ? a.out_scale[(m < a.M ? m : a.M - 1) / a.rows_per_sample] : 1.f;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    float v[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) v[q] = acc[mb][nb][q] * al;
                    store_block<T>(out, a.ldo, m, a.M, n0 + nb * 32, N, hi, v);
                }
            }
        } else if constexpr (EPI == EPI_GELU) {
            T* pre = reinterpret_cast<T*>(a.out2);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int64_t m = m0 + mb * 32 + r;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    float v[16], g[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) v[q] = acc[mb][nb][q];
                    store_block<T>(pre, a.ldo2, m, a.M, n0 + nb * 32, N, hi, v);
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        bool keep[4] = {true, true, true, true};
                        if (thresh) keep4(a.dropout_seed, m, N, n0 + nb * 32 + 8 * gq + 4 * hi, thresh, keep);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float x = to_f32(from_f32<T>(v[4 * gq + j]));      // gelu of the value as stored
                            float e;
                            const float cdf = gelu_cdf(x, e);
                            g[4 * gq + j] = keep[j] ? x * cdf * inv_keep : 0.f;
                        }
                    }
                    store_block<T>(out, a.ldo, m, a.M, n0 + nb * 32, N, hi, g);
                }
            }
        } else if constexpr (EPI == EPI_RESID) {
            const T* res = reinterpret_cast<const T*>(a.res);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int64_t m = m0 + mb * 32 + r;
                const float sc = a.row_scale ? a.row_scale[(m < a.M ? m : a.M - 1) / a.rows_per_sample] : 1.f;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    float rv[16], v[16];
                    load_block<T>(res, a.ldr, m, a.M, n0 + nb * 32, N, hi, rv);
#pragma unroll
                    for (int q = 0; q < 16; ++q) v[q] = rv[q] + acc[mb][nb][q] * sc;
                    store_block<T>(out, a.ldo, m, a.M, n0 + nb * 32, N, hi, v);
                }
            }
        } else if constexpr (EPI == EPI_GELU_BWD) {
            const T* pre = reinterpret_cast<const T*>(a.res);         // the forward's pre-activation
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int64_t m = m0 + mb * 32 + r;
                const float al = a.out_scale ? a.out_scale[(m < a.M ? m : a.M - 1) / a.rows_per_sample] : 1.f;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    float pv[16], v[16];
                    load_block<T>(pre, a.ldr, m, a.M, n0 + nb * 32, N, hi, pv);
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        bool keep[4] = {true, true, true, true};
                        if (thresh) keep4(a.dropout_seed, m, N, n0 + nb * 32 + 8 * gq + 4 * hi, thresh, keep);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float x = pv[4 * gq + j];
                            float e;
                            const float cdf = gelu_cdf(x, e);
                            // the incoming gradient is rounded to the storage type first, as the unfused chain stores it
                            const float dy = to_f32(from_f32<T>(acc[mb][nb][4 * gq + j] * al));
                            v[4 * gq + j] = keep[j] ? dy * (cdf + x * 0.3989422804014327f * e) * inv_keep : 0.f;
                        }
                    }
                    store_block<T>(out, a.ldo, m, a.M, n0 + nb * 32, N, hi, v);
                }
            }
        } else {     // EPI_LN_BWD: acc = dy (gradient at the LayerNorm output); N = the normalised width, one column tile
            const T* S = reinterpret_cast<const T*>(a.res);           // the LayerNorm input (residual stream)
            const T* dsin = reinterpret_cast<const T*>(a.ds_in);      // gradient arriving on the residual stream (may be NULL)
            auto gamma4 = [&](int nb, int gq, float* g4) {             // gamma of the quad's 4 consecutive columns (L1-resident)
                const int n = n0 + nb * 32 + 8 * gq + 4 * hi;
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if (n < N) t = *reinterpret_cast<const float4*>(a.gamma + n);
                g4[0] = t.x; g4[1] = t.y; g4[2] = t.z; g4[3] = t.w;
            };
            float cs_a[NB][16], cs_b[NB][16];                         // per-lane column partials over the row blocks
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int q = 0; q < 16; ++q) cs_a[nb][q] = cs_b[nb][q] = 0.f;
            float rs[MB], mu[MB], s1[MB], s2[MB];
            // pass 1: row sums of g = dy*gamma and g*xhat; column sums of dy*xhat (dgamma) and dy (dbeta)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int64_t m = m0 + mb * 32 + r;
                const bool ok = m < a.M;
                mu[mb] = ok ? a.mean[m] : 0.f;
                rs[mb] = ok ? a.rstd[m] : 0.f;
                float p1 = 0.f, p2 = 0.f;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    float sv[16];
                    load_block<T>(S, a.ldr, m, a.M, n0 + nb * 32, N, hi, sv);
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        float g4[4];
                        gamma4(nb, gq, g4);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int q = 4 * gq + j;
                            const bool cok = ok && (n0 + nb * 32 + acc_row(q, hi) < N);
                            const float dy = cok ? to_f32(from_f32<T>(acc[mb][nb][q])) : 0.f;     // dy as the unfused chain stores it
                            acc[mb][nb][q] = dy;
                            const float x = cok ? (sv[q] - mu[mb]) * rs[mb] : 0.f;
                            const float g = dy * g4[j];
                            p1 += g;
                            p2 += g * x;
                            cs_a[nb][q] += dy * x;
                            cs_b[nb][q] += dy;
                        }
                    }
                }
                s1[mb] = p1 + xhalf(p1);
                s2[mb] = p2 + xhalf(p2);
            }
            if (hi == 0) {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    red[(wave * kBM + mb * 32 + r) * 2] = s1[mb];
                    red[(wave * kBM + mb * 32 + r) * 2 + 1] = s2[mb];
                }
            }
            // fold a per-lane 16-column partial over the 32 lanes of each half-wave: 16 + 8+4+2+1 exchanges; lanes
            // r < 16 end up with the total of register index q = r (column nbase + (q&3) + 8(q>>2) + 4hi)
            auto fold_store = [&](float (&v)[16], int nb, float* dst) {
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
                const int n = n0 + nb * 32 + (q & 3) + 8 * (q >> 2) + 4 * hi;
                if (r < 16 && n < N) dst[n] = v[0];
            };
            float* part = a.colsum_partial ? a.colsum_partial + (int64_t)blockIdx.x * 3 * N : nullptr;
            if (part) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    fold_store(cs_a[nb], nb, part);
                    fold_store(cs_b[nb], nb, part + N);
                }
            }
            __syncthreads();
            const float invC = 1.f / (float)N;
            T* dres = reinterpret_cast<T*>(a.out);
            T* dx = reinterpret_cast<T*>(a.out2);                     // d_res * row_scale (may be NULL)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int q = 0; q < 16; ++q) cs_a[nb][q] = 0.f;       // now: column sums of the x-branch gradient
            // pass 2: dx = rstd * (g - mean(g) - xhat * mean(g*xhat)) + ds_in
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int row = mb * 32 + r;
                float c1 = 0.f, c2 = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    c1 += red[(w * kBM + row) * 2];
                    c2 += red[(w * kBM + row) * 2 + 1];
                }
                c1 *= invC;
                c2 *= invC;
                const int64_t m = m0 + row;
                const float sc = a.row_scale ? a.row_scale[(m < a.M ? m : a.M - 1) / a.rows_per_sample] : 1.f;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    float sv[16], dv[16], ds[16];
                    load_block<T>(S, a.ldr, m, a.M, n0 + nb * 32, N, hi, sv);          // second touch: L2
                    if (dsin) load_block<T>(dsin, a.ld_ds, m, a.M, n0 + nb * 32, N, hi, ds);
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        float g4[4];
                        gamma4(nb, gq, g4);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int q = 4 * gq + j;
                            const float x = (sv[q] - mu[mb]) * rs[mb];
                            float d = rs[mb] * (acc[mb][nb][q] * g4[j] - c1 - x * c2);
                            if (dsin) d += ds[q];
                            dv[q] = d;
                        }
                    }
                    store_block<T>(dres, a.ldo, m, a.M, n0 + nb * 32, N, hi, dv);
                    if (dx || part) {
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            // the x-branch gradient as it is stored (rounded), so that its column sums equal a separate pass's
                            const float t = to_f32(from_f32<T>(to_f32(from_f32<T>(dv[q])) * sc));
                            dv[q] = t;
                            cs_a[nb][q] += (m < a.M && n0 + nb * 32 + acc_row(q, hi) < N) ? t : 0.f;
                        }
                        if (dx) store_block<T>(dx, a.ldo2, m, a.M, n0 + nb * 32, N, hi, dv);
                    }
                }
            }
            if (part) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) fold_store(cs_a[nb], nb, part + 2 * N);
            }
        }
    };

    // ------------------------------------------------------------------ body
    if (chunks == 1) {
        stage(0, K);
        __syncthreads();
        if constexpr (LN) {
            // statistics: 2 threads per row, 16-byte slots; thread `half` starts 8 slots later (other bank groups)
            static_assert(kBM == 128, "the LayerNorm prologue maps 2 threads to each of 128 rows");
            const int row = tid >> 1, half = tid & 1;
            const int spr = K >> 3, per = spr >> 1;
            float s = 0.f;
            for (int j = 0; j < per; ++j) {
                const int slot = half * per + ((j + 8 * half) % per);
                F f = load_frag<T>(reinterpret_cast<const T*>(xs + geo.off(row, slot)));
#pragma unroll
                for (int t = 0; t < 8; ++t) s += to_f32(f[t]);
            }
            s += __shfl_xor(s, 1, 64);
            const float mean = s / (float)K;
            float qv = 0.f;
            for (int j = 0; j < per; ++j) {
                const int slot = half * per + ((j + 8 * half) % per);
                F f = load_frag<T>(reinterpret_cast<const T*>(xs + geo.off(row, slot)));
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const float d = to_f32(f[t]) - mean;
                    qv += d * d;
                }
            }
            qv += __shfl_xor(qv, 1, 64);
            const float rstd = rsqrtf(qv / (float)K + a.eps);
            if (half == 0) {
                st_mean[row] = mean;
                st_rstd[row] = rstd;
                if (m0 + row < a.M) {
                    if (a.mean) a.mean[m0 + row] = mean;
                    if (a.rstd) a.rstd[m0 + row] = rstd;
                }
            }
            __syncthreads();
            // normalise in place: thread -> fixed 16-byte column slot, rows tid/spr + (256/spr)*i
            const int spr2 = K >> 3;
            const int slot = tid % spr2, rstep = 256 / spr2;
            float gam[8], bet[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                gam[t] = a.gamma[slot * 8 + t];
                bet[t] = a.beta[slot * 8 + t];
            }
            T* Y = reinterpret_cast<T*>(a.y);
            for (int row2 = tid / spr2; row2 < kBM; row2 += rstep) {
                T* p = reinterpret_cast<T*>(xs + geo.off(row2, slot));
                F f = load_frag<T>(p);
                const float mu = st_mean[row2], rsd = st_rstd[row2];
#pragma unroll
                for (int t = 0; t < 8; ++t) f[t] = from_f32<T>((to_f32(f[t]) - mu) * rsd * gam[t] + bet[t]);
                uint4 raw;
                __builtin_memcpy(&raw, &f, 16);
                *reinterpret_cast<uint4*>(p) = raw;
                if (Y && m0 + row2 < a.M) *reinterpret_cast<uint4*>(Y + (m0 + row2) * a.ldy + slot * 8) = raw;
            }
            __syncthreads();
        }
        for (int nt = 0; nt < n_tiles; ++nt) {
            const int n0 = nt * NT + wave * NB * 32;
            init_acc(n0);
            kloop(n0, 0, K);
            epilogue(n0);
        }
    } else {
        const int n0 = wave * NB * 32;
        init_acc(n0);
        for (int c = 0; c < chunks; ++c) {
            const int kc0 = c * kKC, kcl = (K - kc0) < kKC ? (K - kc0) : kKC;
            if (c) __syncthreads();                     // every wave is done reading the previous chunk
            stage(kc0, kcl);
            __syncthreads();
            kloop(n0, kc0, kcl);
        }
        epilogue(n0);
    }
}

template <typename T, int MB, int NB, int EPI, bool LN>
static int eg_launch(const tgt_edge_linear_args& a, hipStream_t st) {
    constexpr int kBM = 32 * MB;
    const int kc = a.K < kKC ? a.K : kKC;
    const int lds = kBM * kc * 2 + 2 * kBM * 4 + (EPI == EPI_LN_BWD ? 4 * kBM * 2 * 4 : 0);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_linear_kernel<T, MB, NB, EPI, LN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kBM * kKC * 2 + 2 * kBM * 4 + 4 * kBM * 2 * 4);
        attr_set = true;
    }
    const int64_t grid = (a.M + kBM - 1) / kBM;
    hipLaunchKernelGGL((edge_linear_kernel<T, MB, NB, EPI, LN>), dim3((unsigned)grid), dim3(256), lds, st, a);
    return check_launch("edge_linear_kernel");
}

template <typename T, int NB>
static int eg_dispatch(const tgt_edge_linear_args& a, hipStream_t st) {
    const bool ln = a.gamma != nullptr && a.epilogue != EPI_LN_BWD;
    switch (a.epilogue) {
        case EPI_BIAS: return ln ? eg_launch<T, 4, NB, EPI_BIAS, true>(a, st) : eg_launch<T, 4, NB, EPI_BIAS, false>(a, st);
        case EPI_GELU: return ln ? eg_launch<T, 4, NB, EPI_GELU, true>(a, st) : eg_launch<T, 4, NB, EPI_GELU, false>(a, st);
        case EPI_RESID: return eg_launch<T, 4, NB, EPI_RESID, false>(a, st);
        case EPI_GELU_BWD: return eg_launch<T, 4, NB, EPI_GELU_BWD, false>(a, st);
        case EPI_LN_BWD: return eg_launch<T, 2, NB, EPI_LN_BWD, false>(a, st);
        default: return set_error(TGT_ERR_INVALID, "edge linear: bad epilogue %d", a.epilogue);
    }
}

int edge_linear_parts(int64_t M, int epilogue) { return (int)((M + (epilogue == EPI_LN_BWD ? 64 : 128) - 1) / (epilogue == EPI_LN_BWD ? 64 : 128)); }

int edge_linear_supported(const tgt_edge_linear_args* a) {
    if (!a) return 0;
    if (a->dtype != TGT_BF16 && a->dtype != TGT_F16) return 0;
    const int K = a->K, N = a->N;
    if (K < 16 || N < 8 || N % 8) return 0;
    if (K >= kKC ? (K % 16 != 0) : (K != 16 && K != 32 && K != 64 && K != 128)) return 0;
    const int nt = (N + 255) / 256, chunks = (K + kKC - 1) / kKC;
    if (nt > 1 && chunks > 1) return 0;
    if (a->gamma && a->epilogue != EPI_LN_BWD && chunks > 1) return 0;
    if (a->epilogue == EPI_LN_BWD && (N > 256 || !a->gamma || !a->mean || !a->rstd || !a->res)) return 0;
    return 1;
}

int edge_linear_run(const tgt_edge_linear_args* a, hipStream_t st) {
    if (!a || !a->a || !a->w || !a->out) return set_error(TGT_ERR_INVALID, "edge linear: null argument");
    if (a->M < 0 || a->K <= 0 || a->N <= 0) return set_error(TGT_ERR_INVALID, "edge linear: bad sizes");
    if (!edge_linear_supported(a))
        return set_error(TGT_ERR_UNSUPPORTED, "edge linear: unsupported shape/dtype (K=%d N=%d dtype=%d epilogue=%d): needs a 16-bit "
                         "dtype, N %% 8 == 0, K in {16,32,64,128} or a multiple of 16 >= 256, and not both K > 256 and N > 256",
                         a->K, a->N, a->dtype, a->epilogue);
    if (a->M == 0) return TGT_OK;
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
    // narrow outputs: one 32-column block per wave keeps all four waves busy
    const bool narrow = a->N <= 128;
    if (a->dtype == TGT_BF16) return narrow ? eg_dispatch<bf16_t, 1>(*a, st) : eg_dispatch<bf16_t, 2>(*a, st);
    return narrow ? eg_dispatch<f16_t, 1>(*a, st) : eg_dispatch<f16_t, 2>(*a, st);
}

}  // namespace tgt
