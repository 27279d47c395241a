// TriangularUpdate core (reference lib/tgt/layers/triplet.py:156-172) for gfx950.
//
//   X_in [a,k,h] = sigmoid(Xg_in [a,k,h] + M[a,k]) * Xl_in [a,k,h]      X in {E, V}  ("siglin", :130-132)
//   X_out[k,a,h] = sigmoid(Xg_out[k,a,h] + M[k,a]) * Xl_out[k,a,h]
//   O_in [i,j,h] = sum_k E_in [i,k,h] V_in [j,k,h]          (einsum 'bikh,bjkh->bijh', :166)
//   O_out[i,j,h] = sum_k E_out[k,i,h] V_out[k,j,h]          (einsum 'bkih,bkjh->bijh', :167)
// e4 / v4: (B,N,N,4H) rows = [in_gate | in_lin | out_gate | out_lin] (the lin_E / lin_V outputs);
// out: (B,N,N,2H) = [O_in | O_out].  Scalar values per head (no D axis): O(N^3 H) FMAs on
// O(N^2 H) data, no matrix-core shape to speak of; the head axis is contiguous, so lane <-> head
// gives coalesced rows and the k-loop runs in registers.  Backward: one kernel, lane = (pair, head),
// produces all four gradients of a pair with two N-long loops.
#include "common.hpp"

namespace tgt {

struct TriUpdArgs {
    const void* e4; const void* v4; const float* mask; void* out;
    const void* d_out; void* d_e4; void* d_v4;
    int B, N, H, dtype;
};

template <typename T> __device__ __forceinline__ float ldf(const T* p, int64_t i) { return to_f32(p[i]); }

// lanes per pair = smallest pow2 >= min(H, 64)
__host__ __device__ inline int tu_lpp(int H) { int l = 1; while (l < H && l < 64) l <<= 1; return l; }

template <typename T>
__global__ void __launch_bounds__(256) tri_upd_fwd_kernel(const TriUpdArgs a) {
    const int lpp = tu_lpp(a.H), ppw = 64 / lpp, hb = (a.H + 63) / 64;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t unit = wave * ppw + lane / lpp, total = (int64_t)a.B * a.N * a.N * hb;
    const int h = (int)(unit % hb) * 64 + lane % lpp;
    if (unit >= total || h >= a.H) return;
    const int64_t pair = unit / hb;
    const int N = a.N, H = a.H, j = (int)(pair % N), i = (int)((pair / N) % N);
    const int64_t b = pair / ((int64_t)N * N), base = b * N * N;
    const T* e4 = reinterpret_cast<const T*>(a.e4);
    const T* v4 = reinterpret_cast<const T*>(a.v4);
    float oin = 0.f, oout = 0.f;
    for (int k = 0; k < N; ++k) {
        const int64_t ik = base + (int64_t)i * N + k, jk = base + (int64_t)j * N + k;
        const int64_t ki = base + (int64_t)k * N + i, kj = base + (int64_t)k * N + j;
        const float ein = fast_sigmoid(ldf(e4, ik * 4 * H + h) + a.mask[ik]) * ldf(e4, ik * 4 * H + H + h);
        const float vin = fast_sigmoid(ldf(v4, jk * 4 * H + h) + a.mask[jk]) * ldf(v4, jk * 4 * H + H + h);
        const float eout = fast_sigmoid(ldf(e4, ki * 4 * H + 2 * H + h) + a.mask[ki]) * ldf(e4, ki * 4 * H + 3 * H + h);
        const float vout = fast_sigmoid(ldf(v4, kj * 4 * H + 2 * H + h) + a.mask[kj]) * ldf(v4, kj * 4 * H + 3 * H + h);
        oin += ein * vin;
        oout += eout * vout;
    }
    T* out = reinterpret_cast<T*>(a.out);
    out[pair * 2 * H + h] = from_f32<T>(oin);
    out[pair * 2 * H + H + h] = from_f32<T>(oout);
}

// backward: lane = (pair (x,y), head).  The pair is element [x,y] of e4 AND of v4:
//   as E_in [i=x,k=y]:  dE = sum_j dO_in [x,j] V_in [j,y]        as V_in [j=x,k=y]: dV = sum_i dO_in [i,x] E_in [i,y]
//   as E_out[k=x,i=y]:  dE = sum_j dO_out[y,j] V_out[x,j]        as V_out[k=x,j=y]: dV = sum_i dO_out[i,y] E_out[x,i]
template <typename T>
__global__ void __launch_bounds__(256) tri_upd_bwd_kernel(const TriUpdArgs a) {
    const int lpp = tu_lpp(a.H), ppw = 64 / lpp, hb = (a.H + 63) / 64;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t unit = wave * ppw + lane / lpp, total = (int64_t)a.B * a.N * a.N * hb;
    const int h = (int)(unit % hb) * 64 + lane % lpp;
    if (unit >= total || h >= a.H) return;
    const int64_t pair = unit / hb;
    const int N = a.N, H = a.H, y = (int)(pair % N), x = (int)((pair / N) % N);
    const int64_t b = pair / ((int64_t)N * N), base = b * N * N;
    const T* e4 = reinterpret_cast<const T*>(a.e4);
    const T* v4 = reinterpret_cast<const T*>(a.v4);
    const T* dO = reinterpret_cast<const T*>(a.d_out);
    auto siglin = [&](const T* t, int64_t p, int off) {
        return fast_sigmoid(ldf(t, p * 4 * H + off + h) + a.mask[p]) * ldf(t, p * 4 * H + off + H + h);
    };
    float dEin = 0.f, dVin = 0.f, dEout = 0.f, dVout = 0.f;
    for (int t = 0; t < N; ++t) {
        const int64_t xt = base + (int64_t)x * N + t, tx = base + (int64_t)t * N + x;
        const int64_t ty = base + (int64_t)t * N + y, yt = base + (int64_t)y * N + t;
        dEin += ldf(dO, xt * 2 * H + h) * siglin(v4, ty, 0);            // j = t: dO_in[x,j] V_in[j,y]
        dVin += ldf(dO, tx * 2 * H + h) * siglin(e4, ty, 0);            // i = t: dO_in[i,x] E_in[i,y]
        dEout += ldf(dO, yt * 2 * H + H + h) * siglin(v4, xt, 2 * H);   // j = t: dO_out[y,j] V_out[x,j]
        dVout += ldf(dO, ty * 2 * H + H + h) * siglin(e4, xt, 2 * H);   // i = t: dO_out[i,y] E_out[x,i]
    }
    const float m = a.mask[pair];
    T* de = reinterpret_cast<T*>(a.d_e4);
    T* dv = reinterpret_cast<T*>(a.d_v4);
    auto put = [&](T* d, const T* src, int off, float g) {       // d(siglin): gate and linear parts
        const float s = fast_sigmoid(ldf(src, pair * 4 * H + off + h) + m), l = ldf(src, pair * 4 * H + off + H + h);
        d[pair * 4 * H + off + h] = from_f32<T>(g * l * s * (1.f - s));
        d[pair * 4 * H + off + H + h] = from_f32<T>(g * s);
    };
    put(de, e4, 0, dEin);
    put(dv, v4, 0, dVin);
    put(de, e4, 2 * H, dEout);
    put(dv, v4, 2 * H, dVout);
}

template <typename T>
static int tri_upd_launch(const TriUpdArgs& a, bool bwd, hipStream_t st) {
    const int lpp = tu_lpp(a.H), ppw = 64 / lpp, hb = (a.H + 63) / 64;
    const int64_t units = (int64_t)a.B * a.N * a.N * hb, waves = (units + ppw - 1) / ppw;
    const int grid = (int)((waves + 3) / 4);
    if (!bwd) hipLaunchKernelGGL((tri_upd_fwd_kernel<T>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((tri_upd_bwd_kernel<T>), dim3(grid), dim3(256), 0, st, a);
    return check_launch(bwd ? "tri_upd_bwd_kernel" : "tri_upd_fwd_kernel");
}

int triangular_update_run(const void* e4, const void* v4, const float* mask, void* out, const void* d_out, void* d_e4,
                          void* d_v4, int B, int N, int H, int dtype, bool bwd, hipStream_t st) {
    if (B < 0 || N < 0 || H <= 0) return set_error(TGT_ERR_INVALID, "triangular update: bad size");
    if (B == 0 || N == 0) return TGT_OK;
    if (!e4 || !v4 || !mask) return set_error(TGT_ERR_INVALID, "triangular update: null tensor");
    if (!bwd && !out) return set_error(TGT_ERR_INVALID, "triangular update: null out");
    if (bwd && (!d_out || !d_e4 || !d_v4)) return set_error(TGT_ERR_INVALID, "triangular update bwd: null gradient tensor");
    TriUpdArgs a = {e4, v4, mask, out, d_out, d_e4, d_v4, B, N, H, dtype};
    switch (dtype) {
        case TGT_F32: return tri_upd_launch<float>(a, bwd, st);
        case TGT_BF16: return tri_upd_launch<bf16_t>(a, bwd, st);
        case TGT_F16: return tri_upd_launch<f16_t>(a, bwd, st);
        default: return set_error(TGT_ERR_INVALID, "triangular update: bad dtype %d", dtype);
    }
}

}  // namespace tgt
