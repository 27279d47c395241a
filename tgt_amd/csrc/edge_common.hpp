// Shared device helpers of the edge-row GEMM kernels (edge_gemm.hip, edge_wgrad.hip): tile geometry in LDS, 16-bit <-> fp32
// pair arithmetic of the row phase, tile-based buffer resources.  Moved out of edge_gemm.hip unchanged (round 6).
#pragma once
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
        if (m < M && col < N) st16_stream<TGT_NT_SLICE != 0>(base + m * ld + col, make_uint4(a.x, a.y, b.x, b.y));
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
// sum over the 32 lanes of a half-wave (one row): DPP adds + one v_permlane16_swap (common.hpp), no LDS crossbar
__device__ __forceinline__ float rp_row_sum(float v) { return group_sum<32>(v); }

// The row phase computes on PAIRS: the two 16-bit values of a dword become one f32x2, every arithmetic step is one packed
// instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) and a pair goes back with ONE v_cvt_pk_bf16_f32 -- written on scalars,
// hipcc pairs element 0 of one dword with element 0 of the next and then needs two fix-up instructions per dword to re-pair.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ void rp_unpack(const uint4& raw, f32x2 (&v)[4]) {
    const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if constexpr (std::is_same<T, bf16_t>::value) {
            v[k].x = __builtin_bit_cast(float, w[k] << 16);
            v[k].y = __builtin_bit_cast(float, w[k] & 0xffff0000u);
        } else {
            v[k] = __builtin_convertvector(__builtin_bit_cast(f16x2_t, w[k]), f32x2);
        }
    }
}
template <typename T>
__device__ __forceinline__ unsigned rp_pack2(f32x2 v) {
    if constexpr (std::is_same<T, bf16_t>::value) return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
    else return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
}
template <typename T>
__device__ __forceinline__ uint4 rp_pack(const f32x2 (&v)[4]) {
    return make_uint4(rp_pack2<T>(v[0]), rp_pack2<T>(v[1]), rp_pack2<T>(v[2]), rp_pack2<T>(v[3]));
}
// the value a pair has once stored in T (what a later pass over the stored tensor would read)
template <typename T>
__device__ __forceinline__ f32x2 rp_round(f32x2 v) {
    const uint4 w = make_uint4(rp_pack2<T>(v), 0, 0, 0);
    f32x2 o[4];
    rp_unpack<T>(w, o);
    return o[0];
}
__device__ __forceinline__ f32x2 rp_splat(float x) { f32x2 r = {x, x}; return r; }
// gelu_cdf (common.hpp) on a pair: the polynomial in packed arithmetic, exp / rcp per element
__device__ __forceinline__ f32x2 gelu_cdf2(f32x2 v, f32x2& e) {
    f32x2 ax = {fabsf(v.x), fabsf(v.y)};
    ax = ax * 0.70710678118654752f;
    const f32x2 den = ax * 0.3275911f + 1.f;
    const f32x2 t = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
    const f32x2 q = -(ax * ax);
    e.x = __expf(q.x);
    e.y = __expf(q.y);
    const f32x2 poly = t * (t * (t * (t * (t * 1.061405429f + -1.453152027f) + 1.421413741f) + -0.284496736f) + 0.254829592f);
    const f32x2 h = 0.5f - 0.5f * poly * e;
    f32x2 r = {0.5f + copysignf(h.x, v.x), 0.5f + copysignf(h.y, v.y)};
    return r;
}

// Wave roles.  vmcnt is ONE in-order counter per wave, so a wave that both prefetches and stores can only wait for its
// prefetch together with every store it issued before (measured on two earlier forms of this kernel: load+MFMA time and
// row-phase time simply added up, 0.044 + 0.085 ms for W1+GELU; hipcc additionally answers an outstanding LDS-DMA with
// `s_waitcnt vmcnt(0)` in front of the next LDS read it cannot prove disjoint).  So the two kinds of traffic live in
// different waves of the workgroup:
//   waves 0-7   GEMM role: fetch the next 32-row A tile into registers (16-byte buffer loads; these waves never store, so the
//               compiler's wait before the closing ds_writes counts exactly those loads), k-loop on the current tile, weight
//               slice (32 columns x K per wave) resident in registers, accumulators out through the staging tile;
//   waves 8-15  row role: one pipeline stage behind, the row phase of the previous tile -- operand rows prefetched from
//               global memory a stage ahead into registers, whole-row stores that nobody in this role waits for until
//               the NEXT stage's operands are needed (a full stage later).
// 16 waves = 4 per SIMD (128 registers each): every SIMD holds two waves of each role, so the matrix pipe, the VALU work of
// the row phase and both kinds of memory traffic overlap.  (With 4 + 4 waves the row role ran one wave per SIMD and was
// latency-bound: 0.096 ms for the GELU row phase alone.)  One s_barrier per stage (32 rows) couples the roles; staging
// tiles are double-buffered.
//
// Round 3: every global access goes through a raw BUFFER RESOURCE re-based on the tile (rows [32 t, 32 t + 32) of the tensor,
// `tile_rsrc`): an absent tensor, a row at or past M, an inactive lane simply fall outside the resource -- loads return 0,
// stores are dropped by the address unit.  So the loop bodies of both roles are STRAIGHT-LINE code: no validity predicate, no
// 64-bit vector address arithmetic (one constant 32-bit offset per thread and tensor), and -- the point -- hipcc's wait-count
// insertion can count the in-flight operations exactly.  Read off the ISA of the round-2 form: (1) the GEMM role spilled its
// prefetched A tile to scratch at the 128-register cap, which needs the data and so made the "prefetch" a synchronous load:
// every stage began with an HBM round trip; (2) the row role's conditional stores (`if (row < M)`, optional outputs) let the
// compiler prove only `vmcnt(2..3)` where 4-8 stores were in flight, so every stage also waited for its own stores; (3) the
// per-row `m / rows_per_sample` was a 64-bit software division (~130 instructions, twice a stage) and the row reductions went
// through ds_bpermute.  (3) is FastDiv + DPP (common.hpp), (1) is the bias kept packed (16 registers less) + scalar addressing.
// ---------------------------------------------------------------------------------------------------------------
// rows [row0, row0 + 32) of an (M, ld) row-major tensor whose rows hold `row_used` bytes, as a raw buffer (see above)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const void* base, int64_t ld_bytes, int row_used, int64_t row0, int64_t M) {
    int64_t n = base ? M - row0 : 0;
    n = n < 0 ? 0 : (n > 32 ? 32 : n);
    const int64_t bytes = n > 0 ? (n - 1) * ld_bytes + row_used : 0;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(base)) + row0 * ld_bytes, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint4 rp_ld16(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, TGT_LD_AUX);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void rp_st16(__amdgpu_buffer_rsrc_t r, uint32_t off, const uint4& v) {
    const u32x4_t d = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(d, r, (int)off, 0, TGT_ST_AUX);
}
__device__ __forceinline__ float rp_ld_f32(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
}
__device__ __forceinline__ void rp_st_f32(__amdgpu_buffer_rsrc_t r, uint32_t off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)off, 0, 0);
}


}  // namespace tgt
