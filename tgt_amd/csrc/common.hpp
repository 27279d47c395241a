// Shared device helpers for the gfx950 (CDNA4, wave64) TGT kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/tgt_hip.h"

namespace tgt {

typedef __bf16   bf16_t;
typedef _Float16 f16_t;

typedef __attribute__((ext_vector_type(8)))  __bf16   bf16x8;
typedef __attribute__((ext_vector_type(8)))  _Float16 f16x8;
typedef __attribute__((ext_vector_type(8)))  float    f32x8;
typedef __attribute__((ext_vector_type(4)))  float    f32x4;
typedef __attribute__((ext_vector_type(16))) float    f32x16;
// Cache policy of the big streaming RESULT stores (buffer_store aux field / __builtin_nontemporal_store): 2 = nt.  Every kernel here
// writes each output byte once and nobody re-reads it before the kernel ends; as ordinary write-back stores they leave the
// XCDs' L2s full of dirty lines that are written back at the dependent-kernel boundary and evict what the NEXT kernel wants to
// keep (weights, third-arm tiles).  Same-box A/B inside the training step: nt +2.6 %, sc0 sc1 (write-through) +0.1 %
// (profiles/r04m_ab_nt_stores.txt).  Extending it to the
// plain-pointer stores of the slice / LayerNorm / activation / node kernels (consumers that DO hit in L2 / MALL) lost 2.4 %.
#ifndef TGT_ST_AUX
#define TGT_ST_AUX 2
#endif
#ifndef TGT_LD_AUX
#define TGT_LD_AUX 2            // the same for the streaming operand loads (rows each workgroup reads once): +0.5 ... +1.2 % on top
#endif
typedef __attribute__((ext_vector_type(2)))  float    f32x2;     // packed fp32 pairs: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));      // raw buffer load / store data
// 16 bytes to / from global memory through a plain pointer, optionally with the streaming (nt) policy above
template <bool NT>
__device__ __forceinline__ void st16_stream(void* p, const uint4& v) {
    if constexpr (NT) {
        const u32x4_t d = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(d, reinterpret_cast<u32x4_t*>(p));
    } else {
        *reinterpret_cast<uint4*>(p) = v;
    }
}
template <bool NT>
__device__ __forceinline__ uint4 ld16_stream(const void* p) {
    if constexpr (NT) {
        const u32x4_t d = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
        return make_uint4(d.x, d.y, d.z, d.w);
    } else {
        return *reinterpret_cast<const uint4*>(p);
    }
}
// per-site switches of the experiments behind the defaults (profiles/r04o_ab_nt_sites.txt, r04q_*)
#ifndef TGT_NT_NODE
#define TGT_NT_NODE 0
#endif
#ifndef TGT_NT_SLICE
#define TGT_NT_SLICE 0
#endif
#ifndef TGT_NT_LN
#define TGT_NT_LN 0
#endif
#ifndef TGT_NT_LNLOAD
#define TGT_NT_LNLOAD 0
#endif

// ---------------------------------------------------------------------------
// 32x32 matrix-core tile:  C[m][n] += sum_kk A[m][kk] * B[kk][n],  kk in [0,16)
//
// Operand fragment = 8 elements per lane.  Lane l = (r = l & 31, hi = l >> 5)
// supplies A[m = r][kk = 8*hi + t] / B[kk = 8*hi + t][n = r], t = 0..7.
// 16-bit types: one v_mfma_f32_32x32x16_{bf16,f16}.  float: eight exact-f32
// v_mfma_f32_32x32x2_f32 (instruction t contracts kk = {t, 8+t}); same
// fragment shape, so the kernels are written once.
// Result layout (all types): lane (r,hi), register q in [0,16) holds
//   C[m = (q & 3) + 8*(q >> 2) + 4*hi][n = r].
// ---------------------------------------------------------------------------
template <typename T> struct Frag;
template <> struct Frag<bf16_t> { typedef bf16x8 type; };
template <> struct Frag<f16_t>  { typedef f16x8 type; };
template <> struct Frag<float>  { typedef f32x8 type; };
template <typename T> using frag_t = typename Frag<T>::type;

__device__ __forceinline__ f32x16 mma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mma32(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mma32(f32x8 a, f32x8 b, f32x16 c) {
#pragma unroll
    for (int t = 0; t < 8; ++t) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], c, 0, 0, 0);
    return c;
}

// row index held by accumulator register q of lane-half hi
__device__ __forceinline__ int acc_row(int q, int hi) { return (q & 3) + 8 * (q >> 2) + 4 * hi; }

template <typename T> __device__ __forceinline__ T from_f32(float x) { return static_cast<T>(x); }
template <typename T> __device__ __forceinline__ float to_f32(T x) { return static_cast<float>(x); }

// registers [8c, 8c+8) of an accumulator -> operand fragment of k-chunk c
template <typename T>
__device__ __forceinline__ typename Frag<T>::type pack_chunk(const f32x16& v, int c) {
    typename Frag<T>::type f;
#pragma unroll
    for (int t = 0; t < 8; ++t) f[t] = from_f32<T>(v[8 * c + t]);
    return f;
}
template <typename T>
__device__ __forceinline__ typename Frag<T>::type zero_frag() {
    typename Frag<T>::type f;
#pragma unroll
    for (int t = 0; t < 8; ++t) f[t] = from_f32<T>(0.f);
    return f;
}

// 8 contiguous elements from LDS / global (16-byte aligned for 16-bit types,
// 32-byte span = two 16-byte pieces for float).
template <typename T>
__device__ __forceinline__ typename Frag<T>::type load_frag(const T* p) {
    typename Frag<T>::type f;
    if constexpr (sizeof(T) == 2) {
        uint4 raw = *reinterpret_cast<const uint4*>(p);
        __builtin_memcpy(&f, &raw, 16);
    } else {
        uint4 r0 = *reinterpret_cast<const uint4*>(p);
        uint4 r1 = *reinterpret_cast<const uint4*>(p + 4);
        __builtin_memcpy(&f, &r0, 16);
        __builtin_memcpy(reinterpret_cast<char*>(&f) + 16, &r1, 16);
    }
    return f;
}

__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }
// v_rcp_f32 (1 ulp) -- __frcp_rn expands to the full IEEE division sequence (~10 VALU instructions)
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_sigmoid(float x) { return fast_rcp(1.f + __expf(-x)); }
__device__ __forceinline__ uint32_t mix32(uint32_t h) {      // "lowbias32" integer finalizer
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return h;
}
// exchange with the lane holding the other half of the same column (l ^ 32)
// v_permlane32_swap (gfx950): a VALU exchange of the two 32-lane halves -- __shfl_xor(v, 32) goes through
// the LDS crossbar (ds_bpermute), whose latency sits on the softmax dependency chain three times per j
__device__ __forceinline__ float xhalf(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);     // r[0] = [lo, lo], r[1] = [hi, hi]
    return __builtin_bit_cast(float, (threadIdx.x & 32) ? r[0] : r[1]);
}

// ---------------------------------------------------------------------------
// Cross-lane reductions without the LDS crossbar.  hipcc lowers __shfl_xor to ds_bpermute_b32 plus four index
// instructions per step (v_xor / v_cmp / v_cndmask / v_lshl) and an lgkmcnt wait: ~7 instructions and one LDS round trip
// (~100 cycles on the dependency chain) per step.  Here: DPP adds inside a 16-lane row (quad_perm, row_half_mirror,
// row_mirror: one VALU instruction each), v_permlane16_swap / v_permlane32_swap (gfx950) across rows and halves.
// ALL-REDUCE over aligned groups of W lanes: every lane of a group ends with the group's value.
//
// The swap builtins return {first operand after the swap, second operand after the swap}; called with ONE value for both
// operands hipcc (ROCm 7.2) treats the two results as equal outside a select -- it stores r[0] where r[1] was asked for
// (seen in the ISA of a probe: `v_permlane16_swap v2, v3` followed by `v_add_f32 v2, v2, v2`).  lane_swap_pair() therefore
// hands it two values the compiler cannot relate (an empty asm on a copy) and fences the results the same way.
// ---------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// (mine, partner's) in some order: the values of lanes l and l ^ 16 (ROWS16) or l ^ 32
template <bool ROWS16>
__device__ __forceinline__ void lane_swap_pair(float v, float& a, float& b) {
    unsigned x = __builtin_bit_cast(unsigned, v), y = x;
    asm volatile("" : "+v"(y));
    unsigned r0, r1;
    if constexpr (ROWS16) {
        const auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
        r0 = r[0]; r1 = r[1];
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
        r0 = r[0]; r1 = r[1];
    }
    asm volatile("" : "+v"(r0), "+v"(r1));
    a = __builtin_bit_cast(float, r0);
    b = __builtin_bit_cast(float, r1);
}
struct RedSum { static __device__ __forceinline__ float op(float a, float b) { return a + b; } };
struct RedMax { static __device__ __forceinline__ float op(float a, float b) { return fmaxf(a, b); } };
template <int W, typename OP>
__device__ __forceinline__ float group_reduce(float v) {
    static_assert(W == 1 || W == 2 || W == 4 || W == 8 || W == 16 || W == 32 || W == 64, "group width");
    if constexpr (W >= 2) v = OP::op(v, dpp_mov<0xB1>(v));       // quad_perm [1,0,3,2]: l ^ 1
    if constexpr (W >= 4) v = OP::op(v, dpp_mov<0x4E>(v));       // quad_perm [2,3,0,1]: l ^ 2
    if constexpr (W >= 8) v = OP::op(v, dpp_mov<0x141>(v));      // row_half_mirror: the other quad of the 8
    if constexpr (W >= 16) v = OP::op(v, dpp_mov<0x140>(v));     // row_mirror: the other half of the 16-lane row
    if constexpr (W >= 32) { float a, b; lane_swap_pair<true>(v, a, b); v = OP::op(a, b); }
    if constexpr (W >= 64) { float a, b; lane_swap_pair<false>(v, a, b); v = OP::op(a, b); }
    return v;
}
template <int W> __device__ __forceinline__ float group_sum(float v) { return group_reduce<W, RedSum>(v); }
template <int W> __device__ __forceinline__ float group_max(float v) { return group_reduce<W, RedMax>(v); }
// the value of lane l ^ 16 / l ^ 32 alone (softmax statistics of the 16-wide tiles)
__device__ __forceinline__ float xor16_max(float v) { float a, b; lane_swap_pair<true>(v, a, b); return fmaxf(a, b); }
__device__ __forceinline__ float xor16_sum(float v) { float a, b; lane_swap_pair<true>(v, a, b); return a + b; }

// n / d for 32-bit unsigned n with a run-time divisor fixed for the kernel (rows_per_sample): one v_mul_hi + 4 cheap
// instructions instead of the ~130-instruction 64-bit division sequence (Granlund & Montgomery, fig. 4.1)
struct FastDiv {
    uint32_t mul, sh1, sh2;
    __device__ __forceinline__ explicit FastDiv(uint32_t d) {
        const uint32_t l = d > 1 ? 32 - __builtin_clz(d - 1) : 0;               // ceil(log2 d)
        mul = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << l) - d)) / (d ? d : 1) + 1);
        sh1 = l < 1 ? l : 1;
        sh2 = l > 1 ? l - 1 : 0;
    }
    __device__ __forceinline__ uint32_t div(uint32_t n) const {
        const uint32_t t = __umulhi(mul, n);
        return (t + ((n - t) >> sh1)) >> sh2;
    }
};

// A device-resident step counter mixed into every dropout seed (tgt_set_seed_counter, ABI 29).  A captured (hipGraph) training
// step bakes its host-drawn seeds into the graph; the counter -- bumped by the graph itself once per replay -- gives every replay
// its own drop patterns, and forward and backward of one step see the same value.  NULL (the default): seeds are used as given.
const uint64_t* seed_counter();                    // host side: what tgt_set_seed_counter registered (capi.hip)
__device__ __forceinline__ uint64_t step_seed(uint64_t seed, const uint64_t* ctr) {
    return ctr ? seed + *ctr * 0x9E3779B97F4A7C15ull : seed;
}

// keep/drop of the V consecutive elements of vector `vec` (= first element index / V): one hash
// of (seed, vec), then one 32-bit word per TWO elements, 16 bits each;
// P(keep) = 1 - thresh16 / 65536.
template <int V>
__device__ __forceinline__ void keep_vector(uint64_t seed, int64_t vec, uint32_t thresh16, bool (&keep)[V]) {
    const uint32_t lo = (uint32_t)vec, hi = (uint32_t)((uint64_t)vec >> 32);
    const uint32_t base = mix32(lo ^ (uint32_t)seed) ^ mix32(hi + (uint32_t)(seed >> 32) + 0x9e3779b9u);
#pragma unroll
    for (int w = 0; w < V / 2; ++w) {
        const uint32_t r = mix32(base + (uint32_t)(w + 1) * 0x9e3779b9u);
        keep[2 * w] = (r & 0xffffu) >= thresh16;
        keep[2 * w + 1] = (r >> 16) >= thresh16;
    }
}
// Phi(v) = 0.5 (1 + erf(v / sqrt2)) and exp(-v^2/2): erf by Abramowitz & Stegun 7.1.26
// (|error| <= 1.5e-7, below fp32 parity tolerance); the kernel is otherwise ALU-bound on erff.
__device__ __forceinline__ float gelu_cdf(float v, float& e) {
    const float ax = fabsf(v) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * ax);
    e = __expf(-ax * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    return 0.5f + copysignf(0.5f - 0.5f * poly * e, v);
}

template <typename T> struct DType;
template <> struct DType<float>  { static constexpr int id = TGT_F32; };
template <> struct DType<bf16_t> { static constexpr int id = TGT_BF16; };
template <> struct DType<f16_t>  { static constexpr int id = TGT_F16; };

// host-side error plumbing (capi.hip)
int set_error(int code, const char* fmt, ...);
int check_launch(const char* what);

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: remember it per (kernel instantiation, device).
// `done` is a function-local `static bool attr_set[16] = {}` of the launcher of ONE kernel instantiation.
static inline bool dyn_lds_once(bool (&done)[16], const void* fn, int lds) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    if (dev >= 16 || !done[dev]) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return false;
        if (dev < 16) done[dev] = true;
    }
    return true;
}

}  // namespace tgt
