// Shared machinery of the triplet kernels (attention + aggregate) for gfx950:
// slab geometry, HBM<->register<->LDS staging, matrix-core operand fragments.
// See triplet_attention.hip for the design notes.
#pragma once
#include "common.hpp"

namespace tgt {

template <typename T, int D, int HG>
struct TriGeo {
    static constexpr int kThreads = HG * 64;
    static constexpr int kRowBytes = HG * D * (int)sizeof(T);   // this group's bytes of one row
    static constexpr int kSlots = kRowBytes / 16;               // 16-byte slots per row
    static constexpr int kSlabBytes = 32 * kRowBytes;
    static constexpr int kChunks = 32 * kSlots;
    static constexpr int kIters = (kChunks + kThreads - 1) / kThreads;
    static constexpr int kRowsPerBankRow = kRowBytes >= 256 ? 1 : 256 / kRowBytes;
    static constexpr int kSwzMask = (kSlots < 16 ? kSlots : 16) - 1;
    static constexpr int kDC = (D + 15) / 16;                   // 16-wide d chunks
    static_assert(kRowBytes % 16 == 0, "row piece must be a multiple of 16 bytes");
    static_assert(D % 8 == 0 && D <= 32, "D in {8,16,24,32}");

    // XOR swizzle of the 16-byte slot index so that ds_read_b128 of one slot
    // column over 16 different rows is bank-conflict free.
    __device__ static __forceinline__ int lds_off(int row, int slot) {
        const int f = (row / kRowsPerBankRow) & kSwzMask;
        return row * kRowBytes + ((slot ^ f) << 4);
    }
    // byte offset (inside a slab) of element `col` (in T units, < HG*D) of `row`
    __device__ static __forceinline__ int lds_elem(int row, int col) {
        const int bo = col * (int)sizeof(T);
        return lds_off(row, bo >> 4) + (bo & 15);
    }
};

// Slab staging.  ROWS = rows of the slab (32 per node tile); slab row `row`
// is global row `row0 + row` (rows >= N are zero-filled / not stored).
template <typename G, int ROWS>
struct SlabIO {
    static constexpr int kChunks = ROWS * G::kSlots;
    static constexpr int kIters = (kChunks + G::kThreads - 1) / G::kThreads;
};

template <typename G, int ROWS, int IT = SlabIO<G, ROWS>::kIters>
__device__ __forceinline__ void slab_commit(const uint4 (&pre)[IT], char* slab, int tid) {
#pragma unroll
    for (int it = 0; it < SlabIO<G, ROWS>::kIters; ++it) {
        const int c = it * G::kThreads + tid;
        const int row = c / G::kSlots, slot = c % G::kSlots;
        if (c < SlabIO<G, ROWS>::kChunks) *reinterpret_cast<uint4*>(slab + G::lds_off(row, slot)) = pre[it];
    }
}
// Column sums folded into the slab stores (the bias gradient of the projection that produced the slab's
// tensor): a thread always owns the same 16-byte column chunk (slot = tid % kSlots), so it keeps E private
// fp32 accumulators -- in LDS, the register file is full -- at cs[tid*E .. +E); the workgroup reduces them
// once at the end (slab_colsum_finish).  Fixed order -> deterministic.
// floats of one accumulator plane (one of dQ/dK/dV): E per thread
template <typename G, typename T> constexpr int slab_colsum_plane_floats() { return G::kThreads * (16 / (int)sizeof(T)); }
// sum the per-thread accumulators of plane `cs` over the threads that share a slot and write the
// kSlots*E column sums to out[0 .. kSlots*E)
template <typename G, typename T>
__device__ __forceinline__ void slab_colsum_finish(const float* cs, float* out, int tid) {
    constexpr int E = 16 / (int)sizeof(T), W = G::kSlots * E;
    for (int col = tid; col < W; col += G::kThreads) {
        const int slot = col / E, e = col % E;
        float v = 0.f;
        for (int t = slot; t < G::kThreads; t += G::kSlots) v += cs[t * E + e];
        out[col] = v;
    }
}

// ---------------------------------------------------------------------------
// Buffer-addressed slab IO (the backward kernel is VALU-bound: 64-bit vector address arithmetic
// per slab and j is work it cannot afford).  A slab lives in ONE graph of a (B,N,N,ld) tensor:
//   buffer resource = that graph (base, N*N*ld bytes; accesses past it read 0 / are dropped),
//   soffset (scalar)  = channel offset + j*j_stride + row0*row_stride   -- changes with j
//   voffset (vector)  = row*row_stride + slot*16                        -- fixed per thread
// so the walk over j issues buffer_load/store ... s[rsrc], s_off offen with no vector address math.
// ---------------------------------------------------------------------------
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
struct SlabBuf {
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t chan;         // byte offset of this group's first channel of the tensor in a row
    uint32_t row_stride;   // bytes between consecutive slab rows
    uint32_t j_stride;     // bytes between consecutive j
};
__device__ __forceinline__ __amdgpu_buffer_rsrc_t graph_rsrc(const void* tensor, int64_t graph_bytes, int b) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(tensor)) + (int64_t)b * graph_bytes, 0,
                                             (int)graph_bytes, 0x00020000);
}
template <typename G, int ROWS, int IT = SlabIO<G, ROWS>::kIters>
__device__ __forceinline__ void slab_issue(uint4 (&pre)[IT], const SlabBuf& s, int j, int row0, int N, int tid) {
    const uint32_t so = s.chan + (uint32_t)j * s.j_stride + (uint32_t)row0 * s.row_stride;
#pragma unroll
    for (int it = 0; it < SlabIO<G, ROWS>::kIters; ++it) {
        const int c = it * G::kThreads + tid;
        const int row = c / G::kSlots, slot = c % G::kSlots;
        u32x4_t v = {0, 0, 0, 0};
        if (c < SlabIO<G, ROWS>::kChunks && row0 + row < N)
            v = __builtin_amdgcn_raw_buffer_load_b128(s.rsrc, (int)((uint32_t)row * s.row_stride + (uint32_t)slot * 16u), (int)so, TGT_LD_AUX);
        pre[it] = make_uint4(v.x, v.y, v.z, v.w);
    }
}
__device__ __forceinline__ void buf_store16(const SlabBuf& s, uint4 v, uint32_t voff, uint32_t so) {
    u32x4_t d = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(d, s.rsrc, (int)voff, (int)so, TGT_ST_AUX);
}
template <typename G, int ROWS>
__device__ __forceinline__ void slab_store(const char* slab, const SlabBuf& s, int j, int row0, int N, int tid) {
    const uint32_t so = s.chan + (uint32_t)j * s.j_stride + (uint32_t)row0 * s.row_stride;
#pragma unroll
    for (int it = 0; it < SlabIO<G, ROWS>::kIters; ++it) {
        const int c = it * G::kThreads + tid;
        const int row = c / G::kSlots, slot = c % G::kSlots;
        if (c < SlabIO<G, ROWS>::kChunks && row0 + row < N)
            buf_store16(s, *reinterpret_cast<const uint4*>(slab + G::lds_off(row, slot)),
                        (uint32_t)row * s.row_stride + (uint32_t)slot * 16u, so);
    }
}
// zeros to the slab's rows (a DropPath-dropped graph: tgt_triplet_attention_args.graph_scale)
template <typename G, int ROWS>
__device__ __forceinline__ void slab_store_zero(const SlabBuf& s, int j, int row0, int N, int tid) {
    const uint32_t so = s.chan + (uint32_t)j * s.j_stride + (uint32_t)row0 * s.row_stride;
#pragma unroll
    for (int it = 0; it < SlabIO<G, ROWS>::kIters; ++it) {
        const int c = it * G::kThreads + tid;
        const int row = c / G::kSlots, slot = c % G::kSlots;
        if (c < SlabIO<G, ROWS>::kChunks && row0 + row < N)
            buf_store16(s, make_uint4(0, 0, 0, 0), (uint32_t)row * s.row_stride + (uint32_t)slot * 16u, so);
    }
}
template <typename G, int ROWS, typename T, int IT = SlabIO<G, ROWS>::kIters>
__device__ __forceinline__ void slab_store_add(const char* slab, const uint4 (&prior)[IT], const SlabBuf& s, int j,
                                               int row0, int N, int tid) {
    constexpr int E = 16 / (int)sizeof(T);
    const uint32_t so = s.chan + (uint32_t)j * s.j_stride + (uint32_t)row0 * s.row_stride;
#pragma unroll
    for (int it = 0; it < SlabIO<G, ROWS>::kIters; ++it) {
        const int c = it * G::kThreads + tid;
        const int row = c / G::kSlots, slot = c % G::kSlots;
        if (c < SlabIO<G, ROWS>::kChunks && row0 + row < N) {
            uint4 a = *reinterpret_cast<const uint4*>(slab + G::lds_off(row, slot)), b = prior[it], o;
            T xa[E], xb[E];
            __builtin_memcpy(xa, &a, 16);
            __builtin_memcpy(xb, &b, 16);
#pragma unroll
            for (int t = 0; t < E; ++t) xa[t] = from_f32<T>(to_f32(xa[t]) + to_f32(xb[t]));
            __builtin_memcpy(&o, xa, 16);
            buf_store16(s, o, (uint32_t)row * s.row_stride + (uint32_t)slot * 16u, so);
        }
    }
}
template <typename G, int ROWS, typename T, bool ADD, int IT = SlabIO<G, ROWS>::kIters>
__device__ __forceinline__ void slab_store_sum(const char* slab, const uint4 (&prior)[IT], const SlabBuf& s, int j, int row0,
                                               int N, int tid, float* cs) {
    constexpr int E = 16 / (int)sizeof(T);
    static_assert(G::kThreads % G::kSlots == 0, "a thread must own one column chunk");
    float4* mine = reinterpret_cast<float4*>(cs + tid * E);
    const uint32_t so = s.chan + (uint32_t)j * s.j_stride + (uint32_t)row0 * s.row_stride;
    float acc[E];
#pragma unroll
    for (int t = 0; t < E / 4; ++t) *reinterpret_cast<float4*>(acc + 4 * t) = mine[t];
#pragma unroll
    for (int it = 0; it < SlabIO<G, ROWS>::kIters; ++it) {
        const int c = it * G::kThreads + tid;
        const int row = c / G::kSlots, slot = c % G::kSlots;
        if (c < SlabIO<G, ROWS>::kChunks && row0 + row < N) {
            uint4 a = *reinterpret_cast<const uint4*>(slab + G::lds_off(row, slot));
            T xa[E];
            __builtin_memcpy(xa, &a, 16);
            if constexpr (ADD) {
                T xb[E];
                __builtin_memcpy(xb, &prior[it], 16);
#pragma unroll
                for (int t = 0; t < E; ++t) {
                    const float old = to_f32(xb[t]);
                    xa[t] = from_f32<T>(to_f32(xa[t]) + old);
                    acc[t] += to_f32(xa[t]) - old;
                }
                __builtin_memcpy(&a, xa, 16);
            } else {
#pragma unroll
                for (int t = 0; t < E; ++t) acc[t] += to_f32(xa[t]);
            }
            buf_store16(s, a, (uint32_t)row * s.row_stride + (uint32_t)slot * 16u, so);
        }
    }
#pragma unroll
    for (int t = 0; t < E / 4; ++t) mine[t] = *reinterpret_cast<const float4*>(acc + 4 * t);
}

// operand fragments of head `wave` for slab row r: d in [16*dc + 8*hi, +8)
template <typename T, int D, int HG>
__device__ __forceinline__ void read_frags(frag_t<T> (&f)[(D + 15) / 16], const char* slab,
                                           int wave, int r, int hi) {
    using G = TriGeo<T, D, HG>;
#pragma unroll
    for (int dc = 0; dc < G::kDC; ++dc) {
        const int d0 = 16 * dc + 8 * hi;
        if (d0 < D) {
            if constexpr (sizeof(T) == 2) {
                f[dc] = load_frag<T>(reinterpret_cast<const T*>(slab + G::lds_elem(r, wave * D + d0)));
            } else {
                frag_t<T> v;
                uint4 r0 = *reinterpret_cast<const uint4*>(slab + G::lds_elem(r, wave * D + d0));
                uint4 r1 = *reinterpret_cast<const uint4*>(slab + G::lds_elem(r, wave * D + d0 + 4));
                __builtin_memcpy(&v, &r0, 16);
                __builtin_memcpy(reinterpret_cast<char*>(&v) + 16, &r1, 16);
                f[dc] = v;
            }
        } else {
            f[dc] = zero_frag<T>();
        }
    }
}

// write a transposed result tile X^T[d][row] (lane = row r, registers = d) as
// T into columns [wave*D, wave*D + D) of slab row r.
template <typename T, int D, int HG>
__device__ __forceinline__ void write_rows(char* slab, const f32x16& acc, int wave, int r, int hi) {
    using G = TriGeo<T, D, HG>;
#pragma unroll
    for (int q = 0; q < D / 8; ++q) {
        const int d0 = 8 * q + 4 * hi;      // registers 4q..4q+3 hold d0..d0+3
        T tmp[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) tmp[t] = from_f32<T>(acc[4 * q + t]);
        char* dst = slab + G::lds_elem(r, wave * D + d0);
        if constexpr (sizeof(T) == 2) {
            uint2 v;
            __builtin_memcpy(&v, tmp, 8);
            *reinterpret_cast<uint2*>(dst) = v;
        } else {
            uint4 v;
            __builtin_memcpy(&v, tmp, 16);
            *reinterpret_cast<uint4*>(dst) = v;
        }
    }
}

// identity fragments.  ident_d[dc]: B[kk = d][n = d'] = (d == d') for the d-chunk dc.
template <typename T, int DC>
__device__ __forceinline__ void make_ident_d(frag_t<T> (&f)[DC], int r, int hi) {
#pragma unroll
    for (int dc = 0; dc < DC; ++dc)
#pragma unroll
        for (int t = 0; t < 8; ++t) f[dc][t] = from_f32<T>(r == 16 * dc + 8 * hi + t ? 1.f : 0.f);
}
// ident_k[c]: B[kk][n = k'] = 1 where the fragment element (hi,t) of chunk c
// stands for row k' in the accumulator order: k = 16c + (t&3) + 8(t>>2) + 4hi.
template <typename T>
__device__ __forceinline__ void make_ident_k(frag_t<T> (&f)[2], int r, int hi) {
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int t = 0; t < 8; ++t) f[c][t] = from_f32<T>(r == acc_row(8 * c + t, hi) ? 1.f : 0.f);
}


// ---------------------------------------------------------------------------
// Third-arm tile (bias E, gate logit G, additive mask M over node pairs) of one
// head in accumulator layout: lane column i = r, register q <-> k = acc_row(q,hi).
//   inward  (dir 0): element (i,k) is stored at pair (x,y) = (i,k)
//   outward (dir 1): element (i,k) is stored at pair (x,y) = (k,i)
// biasM = E + M (or -inf past N), gate = sigmoid(G + M) (1 if ungated).
// Entries past N: k >= N gets -inf (weight exactly 0).  Lanes i >= N are padding
// columns whose results are never stored; PAD_COLS_NEG_INF makes their weights
// exactly 0 (backward, where they feed reductions over i).
// ---------------------------------------------------------------------------
struct ThirdArm {
    const void* eg;
    int64_t ld;
    int e_off, g_off;
    const float* mask;     // (B,N,N) or nullptr (no mask in this direction)
    bool biased, gated;
};

template <typename T, bool PAD_COLS_NEG_INF>
__device__ __forceinline__ void load_third_arm(const ThirdArm& ta, int b, int dir, int h, int N, int r, int hi,
                                               float (&biasM)[16], float (&gate)[16], int i0 = 0, int k0 = 0) {
    const T* eg = reinterpret_cast<const T*>(ta.eg);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int k = k0 + acc_row(q, hi), i = i0 + r;
        const bool valid = i < N && k < N;
        const int x = dir == 0 ? i : k, y = dir == 0 ? k : i;
        const int64_t idx = ((int64_t)b * N + x) * N + y;
        const float m = (valid && ta.mask) ? ta.mask[idx] : 0.f;
        float e = 0.f, gl = 0.f;
        if (valid && ta.biased) e = to_f32(eg[idx * ta.ld + ta.e_off + h]);
        if (valid && ta.gated) gl = to_f32(eg[idx * ta.ld + ta.g_off + h]);
        biasM[q] = (k < N && (!PAD_COLS_NEG_INF || i < N)) ? e + m : -INFINITY;
        gate[q] = valid ? (ta.gated ? fast_sigmoid(gl + m) : 1.f) : 0.f;
    }
}

// scatter a per-head (i,k) tile of third-arm gradients back (as T)
template <typename T>
__device__ __forceinline__ void store_third_arm_grad(const ThirdArm& ta, void* d_eg, int b, int dir, int h, int N,
                                                     int r, int hi, const float (&dE)[16], const float (&dG)[16],
                                                     int i0 = 0, int k0 = 0) {
    if (!(ta.biased || ta.gated)) return;
    T* deg = reinterpret_cast<T*>(d_eg);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int k = k0 + acc_row(q, hi), i = i0 + r;
        if (i < N && k < N) {
            const int x = dir == 0 ? i : k, y = dir == 0 ? k : i;
            const int64_t idx = ((int64_t)b * N + x) * N + y;
            if (ta.biased) deg[idx * ta.ld + ta.e_off + h] = from_f32<T>(dE[q]);
            if (ta.gated) deg[idx * ta.ld + ta.g_off + h] = from_f32<T>(dG[q]);
        }
    }
}

// ---------------------------------------------------------------------------
// Staged third arm.  The per-lane gathers above touch one cache line per lane
// per value (48 instructions x 64 lines per wave).  Instead the WORKGROUP pulls
// the E/G columns of its HG heads and the mask for the pair region of one
// query-tile pass through LDS (8 pairs per wave instruction), and each wave
// then picks its head's tile in accumulator layout from LDS; the backward
// scatters dE/dG back the same way.  The staging area aliases the slab
// buffers (used strictly before / after the walk over j).
//   region of pass i0: inward  x in [i0,i0+32), y in [0,32NT)
//                      outward x in [0,32NT),   y in [i0,i0+32)
// row pitch + 4 bytes: lanes that differ in x hit different banks.
// ---------------------------------------------------------------------------
template <typename T, int HG, int NT>
struct ArmStage {
    static constexpr int kVals = 2 * HG;
    static constexpr int kPairBytes = kVals * (int)sizeof(T);
    static constexpr int kOffM = ((32 * NT * 32 * kPairBytes + 4 * 32 * NT + 15) / 16) * 16;
    static constexpr int kBytes = kOffM + 32 * NT * 32 * 4 + 4 * 32 * NT;
    __device__ static __forceinline__ int nx(int dir) { return dir == 0 ? 32 : 32 * NT; }
    __device__ static __forceinline__ int ny(int dir) { return dir == 0 ? 32 * NT : 32; }
    __device__ static __forceinline__ int pitch(int dir) { return ny(dir) * kPairBytes + 4; }
    __device__ static __forceinline__ int mpitch(int dir) { return ny(dir) * 4 + 4; }
};

template <typename T, int HG, int NT>
__device__ __forceinline__ void arm_stage_load(const ThirdArm& ta, int b, int dir, int g, int N, int i0, char* lds,
                                               int tid) {
    using A = ArmStage<T, HG, NT>;
    const int nx = A::nx(dir), ny = A::ny(dir), x0 = dir == 0 ? i0 : 0, y0 = dir == 0 ? 0 : i0;
    const int pitch = A::pitch(dir), mpitch = A::mpitch(dir);
    const T* eg = reinterpret_cast<const T*>(ta.eg);
    for (int idx = tid; idx < nx * ny * A::kVals; idx += HG * 64) {
        const int v = idx % A::kVals, p = idx / A::kVals, yy = p % ny, xx = p / ny;
        const int x = x0 + xx, y = y0 + yy;
        T val = from_f32<T>(0.f);
        if (x < N && y < N) {
            const bool is_e = v < HG;
            if (is_e ? ta.biased : ta.gated)
                val = eg[(((int64_t)b * N + x) * N + y) * ta.ld + (is_e ? ta.e_off + g * HG + v : ta.g_off + g * HG + v - HG)];
        }
        *reinterpret_cast<T*>(lds + xx * pitch + yy * A::kPairBytes + v * (int)sizeof(T)) = val;
    }
    for (int idx = tid; idx < nx * ny; idx += HG * 64) {
        const int yy = idx % ny, xx = idx / ny, x = x0 + xx, y = y0 + yy;
        float m = 0.f;
        if (x < N && y < N && ta.mask) m = ta.mask[((int64_t)b * N + x) * N + y];
        *reinterpret_cast<float*>(lds + A::kOffM + xx * mpitch + yy * 4) = m;
    }
}

// tile (query tile at i0, key tile kt) of head `hh` of the group, accumulator layout
template <typename T, int HG, int NT, bool PAD_COLS_NEG_INF>
__device__ __forceinline__ void arm_stage_read(const ThirdArm& ta, const char* lds, int dir, int hh, int N, int r, int hi,
                                               int i0, int kt, float (&biasM)[16], float (&gate)[16]) {
    using A = ArmStage<T, HG, NT>;
    const int pitch = A::pitch(dir), mpitch = A::mpitch(dir);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int kl = 32 * kt + acc_row(q, hi), k = kl, i = i0 + r;
        const bool valid = i < N && k < N;
        const int xx = dir == 0 ? r : kl, yy = dir == 0 ? kl : r;
        const char* pp = lds + xx * pitch + yy * A::kPairBytes;
        const float e = to_f32(*reinterpret_cast<const T*>(pp + hh * (int)sizeof(T)));
        const float gl = to_f32(*reinterpret_cast<const T*>(pp + (HG + hh) * (int)sizeof(T)));
        const float m = *reinterpret_cast<const float*>(lds + A::kOffM + xx * mpitch + yy * 4);
        biasM[q] = (k < N && (!PAD_COLS_NEG_INF || i < N)) ? e + m : -INFINITY;
        gate[q] = valid ? (ta.gated ? fast_sigmoid(gl + m) : 1.f) : 0.f;
    }
}

template <typename T, int HG, int NT>
__device__ __forceinline__ void arm_stage_put_grad(char* lds, int dir, int hh, int r, int hi, int kt,
                                                   const float (&dE)[16], const float (&dG)[16]) {
    using A = ArmStage<T, HG, NT>;
    const int pitch = A::pitch(dir);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int kl = 32 * kt + acc_row(q, hi);
        const int xx = dir == 0 ? r : kl, yy = dir == 0 ? kl : r;
        char* pp = lds + xx * pitch + yy * A::kPairBytes;
        *reinterpret_cast<T*>(pp + hh * (int)sizeof(T)) = from_f32<T>(dE[q]);
        *reinterpret_cast<T*>(pp + (HG + hh) * (int)sizeof(T)) = from_f32<T>(dG[q]);
    }
}

// Returns this thread's sum of the values it stored: idx % kVals is fixed per thread
// ((HG*64) % kVals == 0), i.e. a partial column sum of dE (v < HG) or dG of head g*HG + v % HG.
template <typename T, int HG, int NT>
__device__ __forceinline__ float arm_stage_store_grad(const ThirdArm& ta, void* d_eg, int b, int dir, int g, int N,
                                                      int i0, const char* lds, int tid) {
    using A = ArmStage<T, HG, NT>;
    static_assert((HG * 64) % A::kVals == 0, "a thread must own one E/G column");
    float part = 0.f;
    if (!(ta.biased || ta.gated)) return part;
    const int nx = A::nx(dir), ny = A::ny(dir), x0 = dir == 0 ? i0 : 0, y0 = dir == 0 ? 0 : i0;
    const int pitch = A::pitch(dir);
    T* deg = reinterpret_cast<T*>(d_eg);
#pragma nounroll
    for (int idx = tid; idx < nx * ny * A::kVals; idx += HG * 64) {
        const int v = idx % A::kVals, p = idx / A::kVals, yy = p % ny, xx = p / ny;
        const int x = x0 + xx, y = y0 + yy;
        const bool is_e = v < HG;
        if (x < N && y < N && (is_e ? ta.biased : ta.gated)) {
            const T val = *reinterpret_cast<const T*>(lds + xx * pitch + yy * A::kPairBytes + v * (int)sizeof(T));
            deg[(((int64_t)b * N + x) * N + y) * ta.ld + (is_e ? ta.e_off + g * HG + v : ta.g_off + g * HG + v - HG)] = val;
            part += to_f32(val);
        }
    }
    return part;
}

// ---------------------------------------------------------------------------
// Attention dropout (reference lib/tgt/layers/triplet.py:223-225, :242-244, :59-60: F.dropout on
// the gated weights).  Counter-based, so the backward recomputes the forward's pattern from the
// seed and nothing is stored.  Element (unit, i, k): `unit` numbers the (graph, direction, head
// [, shared node j]) the weights belong to; one hash word serves the pair (k even, k + 1):
//   word = mix32( mix32(seed_lo ^ mix32(unit)) + seed_hi + ((i*64 + k) >> 1) * 0x9e3779b9 )
//   keep(k even) = (word & 0xffff) >= thresh16,  keep(k odd) = (word >> 16) >= thresh16,
//   thresh16 = clamp(round(p * 65536), 1, 65535); kept weights are scaled by 1/(1-p).
// (tests/golden_util.py::triplet_dropout_keep restates it in numpy for the parity tests.)
// Returns the keep bits of the 16 accumulator elements of one lane: bit q <-> k = 32*kt + acc_row(q,hi).
// ---------------------------------------------------------------------------
struct TriDrop {
    uint32_t thresh16, seed_lo, seed_hi;
    float scale;
    bool on;
};
__device__ __forceinline__ TriDrop tri_drop(float p, uint64_t seed) {
    TriDrop d;
    d.on = p > 0.f;
    float t = rintf(p * 65536.f);
    t = t < 1.f ? 1.f : (t > 65535.f ? 65535.f : t);
    d.thresh16 = (uint32_t)t;
    d.seed_lo = (uint32_t)seed;
    d.seed_hi = (uint32_t)(seed >> 32);
    d.scale = d.on ? 1.f / (1.f - p) : 1.f;
    return d;
}
__device__ __forceinline__ uint32_t tri_drop_bits(const TriDrop& d, uint32_t unit, int i, int kt, int hi) {
    const uint32_t base = mix32(d.seed_lo ^ mix32(unit)) + d.seed_hi;
    uint32_t bits = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const int k = 32 * kt + acc_row(2 * w, hi);
        const uint32_t r = mix32(base + (uint32_t)((i * 64 + k) >> 1) * 0x9e3779b9u);
        bits |= ((r & 0xffffu) >= d.thresh16 ? 1u : 0u) << (2 * w);
        bits |= ((r >> 16) >= d.thresh16 ? 1u : 0u) << (2 * w + 1);
    }
    return bits;
}

// ---------------------------------------------------------------------------
// workgroup coordinates and slab sources of the triplet-attention kernels
// ---------------------------------------------------------------------------
struct TriCtx {
    int b, dir, g, h, N;
};

template <typename T, int D, int HG>
__device__ __forceinline__ TriCtx tri_ctx(const tgt_triplet_attention_args& a, int wave) {
    TriCtx c;
    const int ngroups = a.H / HG;
    int bid = blockIdx.x;
    c.g = bid % ngroups;
    bid /= ngroups;
    c.dir = bid & 1;
    c.b = bid >> 1;
    c.h = c.g * HG + wave;
    c.N = a.N;
    return c;
}

__device__ __forceinline__ ThirdArm tri_third_arm(const tgt_triplet_attention_args& a, int dir) {
    return ThirdArm{a.eg[dir], a.ld_eg[dir], a.e_off[dir], a.g_off[dir], a.mask,
                    (a.flags & TGT_TRI_BIASED) != 0, (a.flags & TGT_TRI_GATED) != 0};
}

}  // namespace tgt
