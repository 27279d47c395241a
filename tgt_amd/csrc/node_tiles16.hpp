// Shared pieces of the 16-wide-tile node attention kernels (node_attention16.hip: whole key rows per query block;
// node_attention_kb.hip: key-blocked, all heads of a pair's 128-byte rows in one workgroup): operand fragments of
// v_mfma_f32_16x16x16, the 4 x 8 half-word transposes of the staging threads, 16-byte buffer accesses, the launch order.
#pragma once
#include "common.hpp"
#include "triplet_common.hpp"

namespace tgt {
namespace na16 {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
template <typename T> struct F4;
template <> struct F4<bf16_t> { typedef s16x4 type; };
template <> struct F4<f16_t> { typedef h16x4 type; };
template <typename T> using frag4_t = typename F4<T>::type;
template <typename T> inline constexpr bool kIsBf16 = false;
template <> inline constexpr bool kIsBf16<bf16_t> = true;

// C[m][n] += sum_kk A[m][kk] B[kk][n], kk in [0,16): lane l = (x = l & 15, g = l >> 4) supplies A[m = x][kk = 4g + t] /
// B[kk = 4g + t][n = x], t = 0..3, and holds C[m = 4g + q][n = x], q = 0..3
__device__ __forceinline__ f32x4 mma16(s16x4 a, s16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mma16(h16x4 a, h16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); }

template <typename T>
__device__ __forceinline__ frag4_t<T> pack4(const f32x4& v) {
    T t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = from_f32<T>(v[i]);
    frag4_t<T> f;
    __builtin_memcpy(&f, t, 8);
    return f;
}
template <typename T>
__device__ __forceinline__ uint2 pack4u(const float (&v)[4]) {
    T t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = from_f32<T>(v[i]);
    uint2 u;
    __builtin_memcpy(&u, t, 8);
    return u;
}
template <typename T>
__device__ __forceinline__ void unpack4(const uint2& u, float (&v)[4]) {
    T t[4];
    __builtin_memcpy(t, &u, 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = to_f32(t[i]);
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
// half-wave / quarter-wave exchanges of the softmax statistics: lanes x, x + 16, x + 32, x + 48 hold one query
// (v_permlane16_swap / v_permlane32_swap: VALU, not the LDS crossbar -- common.hpp)
__device__ __forceinline__ float qsum(float v) {
    float a, b;
    lane_swap_pair<true>(v, a, b); v = a + b;
    lane_swap_pair<false>(v, a, b); return a + b;
}
__device__ __forceinline__ float qmax(float v) {
    float a, b;
    lane_swap_pair<true>(v, a, b); v = fmaxf(a, b);
    lane_swap_pair<false>(v, a, b); return fmaxf(a, b);
}

constexpr uint32_t kOob = 0x7ffffff0u;

__device__ __forceinline__ uint4 buf_ld16(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void buf_st16(__amdgpu_buffer_rsrc_t r, uint32_t off, const uint4& v) {
    const u32x4_t d = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(d, r, (int)off, 0, 0);
}
__device__ __forceinline__ uint32_t dw(const uint4& v, int p) { return p == 0 ? v.x : p == 1 ? v.y : p == 2 ? v.z : v.w; }
// v[i] = the 8 halves (heads 0..7) of row i  ->  o[j] = the 4 halves (rows 0..3) of head j
__device__ __forceinline__ void tr4x8(const uint4 (&v)[4], uint2 (&o)[8]) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        o[2 * p].x = __builtin_amdgcn_perm(dw(v[1], p), dw(v[0], p), 0x05040100u);
        o[2 * p].y = __builtin_amdgcn_perm(dw(v[3], p), dw(v[2], p), 0x05040100u);
        o[2 * p + 1].x = __builtin_amdgcn_perm(dw(v[1], p), dw(v[0], p), 0x07060302u);
        o[2 * p + 1].y = __builtin_amdgcn_perm(dw(v[3], p), dw(v[2], p), 0x07060302u);
    }
}
// a[j] = the 4 halves (rows 0..3) of head j  ->  the 8 halves of row i
__device__ __forceinline__ uint4 tr8x4_row(const uint2 (&a)[8], int i) {
    const uint32_t sel = (i & 1) ? 0x07060302u : 0x05040100u;
    uint32_t s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = (i >> 1) ? a[j].y : a[j].x;
    return make_uint4(__builtin_amdgcn_perm(s[1], s[0], sel), __builtin_amdgcn_perm(s[3], s[2], sel),
                      __builtin_amdgcn_perm(s[5], s[4], sel), __builtin_amdgcn_perm(s[7], s[6], sel));
}
// eight 8-byte LDS accesses of a staging thread: head j at p + j * head_pitch
__device__ __forceinline__ void lds_put8x8(char* p, int head_pitch, const uint2 (&o)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) *reinterpret_cast<uint2*>(p + j * head_pitch) = o[j];
}
__device__ __forceinline__ void lds_get8x8(const char* p, int head_pitch, uint2 (&o)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = *reinterpret_cast<const uint2*>(p + j * head_pitch);
}

// Unit order: XCD x (= block index & 7) owns the graphs b = x mod 8 and hands their units to its workgroups in order, so that the
// units of one graph (the head groups of a query block share its 128-byte E / G rows, the query blocks its K / V rows) run
// together on one XCD.
__device__ __forceinline__ bool unit_of_block(int B, int per_graph, int& b, int& sub) {
    const int x = blockIdx.x & 7, t = blockIdx.x >> 3;
    const int nt = ((B - x + 7) >> 3) * per_graph;
    b = (t / per_graph) * 8 + x;
    sub = t % per_graph;
    return t < nt;
}

}  // namespace na16
}  // namespace tgt
