// Node attention with edge bias and gate (EGT_Attention core), KEY-BLOCKED: the forward for 16-bit dtypes, any N, H a multiple of
// 32, D in {8, 12, 16} -- the bias / softmax path of BASELINE.json's north_star at BASELINE width (H = 64, D = 12).
//
// Replaces reference lib/tgt/layers/layers.py:62-77 (einsum -> +E -> softmax * sigmoid gate -> einsum -> degree scaler).
// Math: SURVEY.md App. A.1; same arithmetic as node_attention_mfma.hip / node_attention16.hip up to the order of the softmax sums.
//
// Why.  One pair's E | G row at BASELINE width is 64 + 64 heads x 2 bytes = two 128-byte lines.  node_attention_mfma.hip (16 heads per
// workgroup) uses 32 bytes of each line per workgroup, node_attention16.hip (8 heads) 16 bytes, and the other head groups' workgroups
// fetch the rest from L2.  tools/probes/piece_probe.hip (profiles/r06q_piece_probe.txt) prices that: a stream read in 16-byte pieces
// of each line reaches 2.0 TB/s, 32-byte pieces 3.6, 64-byte pieces 4.4, whole lines 6.8 (read + write: 2.9 / 3.7 / 4.5 / 5.0) --
// the vector memory path moves whole lines per request, so the piece size caps those two kernels at 0.28-0.36 of HBM.
// Here a workgroup takes HALF rows (32 heads = 64 bytes of each line), and what no longer fits the LDS is made up by walking the keys
// in blocks of 16 with an online softmax:
//   workgroup = (graph b, query block qb of 16 queries, chunk of 32 heads), 8 waves, wave = 4 heads; 72 KB of LDS and <= 128
//   registers: TWO workgroups per CU, one's loads and stores under the other's tile math;
//   per key block kb: the (16 x 16 pairs) x (E | G) image lands in LDS KEY-MAJOR PER HEAD (plane[query][head][key]: the staging
//   threads transpose four keys x eight heads in registers, node_tiles16.hpp), so a lane's four keys of a head are one 8-byte access;
//   heads are rotated inside their octet by the octet index so that the staging writes of a half-wave fall on 64 different banks.
//   K / V rows of the key block and the block's Q rows likewise [row][head][d];
//   per head: S^T tile = one v_mfma_f32_16x16x16, H_hat = S + E back into the E slot, running maximum / sum / gate sum per query in
//   registers (lanes x, x + 16, x + 32, x + 48 share a query: two VALU lane swaps per tile for the maximum, sums stay per lane until
//   the end), O^T accumulators rescaled when the maximum moves.
// Measured alone (tools/probes/r06r.sh, profiles/r06r_node_kernels.txt): B = 256, N = 32: 0.044 ms = 3.45 TB/s against 0.050
// (node_attention_mfma.hip); B = 128, N = 48: 0.0435 ms = 3.5 TB/s against 0.057 (node_attention16.hip).  The WHOLE-row form (64 heads,
// 8 waves x 8 heads at 228-244 registers, 142 KB of LDS: one workgroup per CU with the next key block prefetched into registers)
// was built first and measured 0.052 / 0.058 ms: with one workgroup per CU the land / math / store phases of a tile add up.
// The backward stays on the other two files: its accumulators (dQ of every query block or dK / dV of every key block, per head)
// do not leave room for 32 heads per workgroup at two workgroups per CU.
#include <cstdlib>
#include "node_tiles16.hpp"

namespace tgt {
namespace nkb {

using namespace na16;

constexpr int kWaves = 8, kThreads = kWaves * 64;

template <int D, int HPW>
struct Lay {
    static constexpr int HW = kWaves * HPW, OC = HW / 8, DQ = D / 4;
    static constexpr int kPitchP = HW * 32 + 16;          // pair plane, per query: HW heads x 16 keys x 2 bytes (pitch / 16 odd)
    static constexpr int kPitchM = 80;                    // mask tile (fp32): 16 keys
    static constexpr int kHeadN = D * 2;
    static constexpr int kPitchN = HW * kHeadN + 16;      // node rows: HW heads x D
    static constexpr int kOffE = 0;
    static constexpr int kOffG = kOffE + 16 * kPitchP + 16;                  // (+16: the G octets of a staging half-wave take the banks the E octets leave)
    static constexpr int kOffM = kOffG + 16 * kPitchP;
    static constexpr int kOffQ = kOffM + 16 * kPitchM;                       // the block's Q rows; V_att leaves through it
    static constexpr int kOffK = kOffQ + 16 * kPitchN;
    static constexpr int kOffV = kOffK + 16 * kPitchN;
    static constexpr int kFwdBytes = kOffV + 16 * kPitchN;
    static_assert((kPitchP / 16) % 2 == 1 && (kPitchN / 16) % 2 == 1, "odd pitches");
};
// position of local head hl inside a pair plane: rotated inside its octet by the octet index
__device__ __forceinline__ int head_pos(int hl) { return (hl & ~7) | ((hl + (hl >> 3)) & 7); }

struct Unit { int b, hc, qb; };                            // graph, head chunk, query block

// ---- the pair image of one key block: task = (query l, key quad mq, octet o of [E octets | G octets])
template <typename T, int D, int HPW>
struct ImageIO {
    using L = Lay<D, HPW>;
    static constexpr int OCT = 2 * L::OC, kTasks = 16 * 4 * OCT, kIters = (kTasks + kThreads - 1) / kThreads;
    uint4 v[kIters][4];

    __device__ __forceinline__ void issue(const tgt_node_attention_args& a, const Unit& u, int kb, int tid) {
        asm volatile("" : "+v"(tid));
        const int N = a.N;
        const __amdgpu_buffer_rsrc_t rs = graph_rsrc(a.eg, (int64_t)N * N * a.ld_eg * sizeof(T), u.b);
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            const int task = it * kThreads + tid, o = task % OCT, mq = (task / OCT) & 3, l = task / (4 * OCT), q = 16 * u.qb + l;
            const int ch = (o >= L::OC ? a.g_off + (o - L::OC) * 8 : a.e_off + o * 8) + u.hc * L::HW;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = 16 * kb + 4 * mq + i;
                const bool ok = task < kTasks && q < N && m < N;
                v[it][i] = buf_ld16(rs, ok ? (uint32_t)(((int64_t)(q * N + m) * a.ld_eg + ch) * (int64_t)sizeof(T)) : kOob);
            }
        }
    }
    __device__ __forceinline__ void land(char* lds, int tid) {
        asm volatile("" : "+v"(tid));
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            const int task = it * kThreads + tid, o = task % OCT, mq = (task / OCT) & 3, l = task / (4 * OCT);
            const int oh = o >= L::OC ? o - L::OC : o;
            char* base = lds + (o >= L::OC ? L::kOffG : L::kOffE) + l * L::kPitchP + oh * 256 + mq * 8;
            uint2 t[8];
            tr4x8(v[it], t);
            if (task < kTasks) {
#pragma unroll
                for (int j = 0; j < 8; ++j) *reinterpret_cast<uint2*>(base + ((j + oh) & 7) * 32) = t[j];
            }
        }
    }
    // H_hat (the E plane after the tile math) -> hhat[b, q, m, heads of the chunk]
    static __device__ __forceinline__ void store_hhat(const char* lds, const tgt_node_attention_args& a, const Unit& u, int kb, int tid) {
        asm volatile("" : "+v"(tid));
        constexpr int kT = 16 * 4 * L::OC, kI = (kT + kThreads - 1) / kThreads;
        const int N = a.N;
        const __amdgpu_buffer_rsrc_t rs = graph_rsrc(a.hhat, (int64_t)N * N * a.H * sizeof(T), u.b);
#pragma unroll
        for (int it = 0; it < kI; ++it) {
            const int task = it * kThreads + tid, oh = task % L::OC, mq = (task / L::OC) & 3, l = task / (4 * L::OC), q = 16 * u.qb + l;
            if (task < kT) {
                const char* base = lds + L::kOffE + l * L::kPitchP + oh * 256 + mq * 8;
                uint2 t[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] = *reinterpret_cast<const uint2*>(base + ((j + oh) & 7) * 32);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = 16 * kb + 4 * mq + i;
                    const bool ok = q < N && m < N;
                    buf_st16(rs, ok ? (uint32_t)(((int64_t)(q * N + m) * a.H + u.hc * L::HW + oh * 8) * (int64_t)sizeof(T)) : kOob, tr8x4_row(t, i));
                }
            }
        }
    }
};

// ---- node rows: task = (segment, row, d quad dq, octet oh): the 16-byte (8 heads) records of (row0 + row, d = 4 dq + i), i < 4
template <typename T, int D, int HPW, int SEGS>
struct RowsIO {
    using L = Lay<D, HPW>;
    static constexpr int kPerSeg = 16 * L::DQ * L::OC, kTasks = SEGS * kPerSeg, kIters = (kTasks + kThreads - 1) / kThreads;
    uint4 v[kIters][4];

    __device__ __forceinline__ void issue(const void* x, int64_t ld, const int (&off)[SEGS], int row0, const tgt_node_attention_args& a, const Unit& u,
                                          int tid) {
        asm volatile("" : "+v"(tid));
        const __amdgpu_buffer_rsrc_t rs = graph_rsrc(x, (int64_t)a.N * ld * sizeof(T), u.b);
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            const int task = it * kThreads + tid, oh = task % L::OC, dq = (task / L::OC) % L::DQ, row = (task / (L::OC * L::DQ)) & 15, seg = task / kPerSeg;
            int o = off[0];
#pragma unroll
            for (int s = 1; s < SEGS; ++s) o = seg == s ? off[s] : o;
            const bool ok = task < kTasks && row0 + row < a.N;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                v[it][i] = buf_ld16(rs, ok ? (uint32_t)(((int64_t)(row0 + row) * ld + o + (4 * dq + i) * a.H + u.hc * L::HW + oh * 8) * (int64_t)sizeof(T))
                                           : kOob);
        }
    }
    __device__ __forceinline__ void land(char* const (&region)[SEGS], int tid) {
        asm volatile("" : "+v"(tid));
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            const int task = it * kThreads + tid, oh = task % L::OC, dq = (task / L::OC) % L::DQ, row = (task / (L::OC * L::DQ)) & 15, seg = task / kPerSeg;
            char* base = region[0];
#pragma unroll
            for (int s = 1; s < SEGS; ++s) base = seg == s ? region[s] : base;
            uint2 t[8];
            tr4x8(v[it], t);
            if (task < kTasks) lds_put8x8(base + row * L::kPitchN + oh * 8 * L::kHeadN + dq * 8, L::kHeadN, t);
        }
    }
    static __device__ __forceinline__ void store(const char* region, void* x, int64_t ld, int off, int row0, const tgt_node_attention_args& a,
                                                 const Unit& u, int tid) {
        static_assert(SEGS == 1, "one segment");
        asm volatile("" : "+v"(tid));
        const __amdgpu_buffer_rsrc_t rs = graph_rsrc(x, (int64_t)a.N * ld * sizeof(T), u.b);
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            const int task = it * kThreads + tid, oh = task % L::OC, dq = (task / L::OC) % L::DQ, row = (task / (L::OC * L::DQ)) & 15;
            if (task < kTasks) {
                uint2 t[8];
                lds_get8x8(region + row * L::kPitchN + oh * 8 * L::kHeadN + dq * 8, L::kHeadN, t);
                const bool ok = row0 + row < a.N;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    buf_st16(rs, ok ? (uint32_t)(((int64_t)(row0 + row) * ld + off + (4 * dq + i) * a.H + u.hc * L::HW + oh * 8) * (int64_t)sizeof(T)) : kOob,
                             tr8x4_row(t, i));
            }
        }
    }
};

// mask tile of (query block, key block): threads 0..63 hold one key quad of one query; pairs past N get -inf
struct MaskIO {
    float mk[4];
    __device__ __forceinline__ void issue(const tgt_node_attention_args& a, const Unit& u, int kb, int tid) {
        const int N = a.N, l = (tid >> 2) & 15, mq = tid & 3, q = 16 * u.qb + l;
        const __amdgpu_buffer_rsrc_t rs = graph_rsrc(a.mask, (int64_t)N * N * 4, u.b);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = 16 * kb + 4 * mq + i;
            const bool ok = tid < 64 && q < N && m < N;
            const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, ok ? (q * N + m) * 4 : (int)kOob, 0, 0));
            mk[i] = ok ? v : -INFINITY;
        }
    }
    template <int PITCH>
    __device__ __forceinline__ void land(char* lds_m, int tid) {
        const int l = (tid >> 2) & 15, mq = tid & 3;
        if (tid < 64) *reinterpret_cast<float4*>(lds_m + l * PITCH + mq * 16) = make_float4(mk[0], mk[1], mk[2], mk[3]);
    }
};

template <typename T, int D, int PITCH>
__device__ __forceinline__ frag4_t<T> row_frag(const char* region, int row, int g, int hl) {
    const int gg = 4 * g < D ? g : 0;
    uint2 u = *reinterpret_cast<const uint2*>(region + row * PITCH + hl * (D * 2) + gg * 8);
    if (4 * g >= D) u = make_uint2(0u, 0u);
    frag4_t<T> f;
    __builtin_memcpy(&f, &u, 8);
    return f;
}

template <typename T, int D, int HPW>
__global__ void __launch_bounds__(kThreads, 4) node_att_kb_fwd_kernel(const tgt_node_attention_args a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    using L = Lay<D, HPW>;
    using F = frag4_t<T>;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), x = lane & 15, g = lane >> 4;
    const int N = a.N, H = a.H, nkb = (N + 15) / 16, chunks = H / L::HW;
    int b, sub;
    if (!unit_of_block(a.B, chunks * nkb, b, sub)) return;
    const Unit u{b, sub % chunks, sub / chunks};
    char* rQ = lds + L::kOffQ;
    {
        RowsIO<T, D, HPW, 1> qio;
        const int off[1] = {a.q_off};
        qio.issue(a.qkv, a.ld_qkv, off, 16 * u.qb, a, u, tid);
        char* const reg[1] = {rQ};
        qio.land(reg, tid);
    }
    ImageIO<T, D, HPW> img;
    RowsIO<T, D, HPW, 2> kv;
    MaskIO msk;
    const int off_kv[2] = {a.k_off, a.v_off};
    char* const reg_kv[2] = {lds + L::kOffK, lds + L::kOffV};

    const F id = ident4<T>(x, g);
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const float hs = a.hhat_scale ? a.hhat_scale[u.b] : 1.f;
    f32x4 o[HPW];
    float mx[HPW], lsum[HPW], gs[HPW];
#pragma unroll
    for (int i = 0; i < HPW; ++i) { o[i] = z; mx[i] = -INFINITY; lsum[i] = 0.f; gs[i] = 0.f; }
    const char* pm = lds + L::kOffM + x * L::kPitchM + g * 16;

    for (int kb = 0; kb < nkb; ++kb) {
        img.issue(a, u, kb, tid);                          // (the other workgroup of the CU computes under these loads)
        kv.issue(a.qkv, a.ld_qkv, off_kv, 16 * kb, a, u, tid);
        msk.issue(a, u, kb, tid);
        img.land(lds, tid);
        kv.land(reg_kv, tid);
        msk.template land<L::kPitchM>(lds + L::kOffM, tid);
        __syncthreads();
        const float4 mk4 = *reinterpret_cast<const float4*>(pm);
        const float mk[4] = {mk4.x, mk4.y, mk4.z, mk4.w};
#pragma unroll
        for (int i = 0; i < HPW; ++i) {
            const int hl = w * HPW + i, pos = head_pos(hl);
            char* pe = lds + L::kOffE + x * L::kPitchP + pos * 32 + g * 8;
            const char* pg = lds + L::kOffG + x * L::kPitchP + pos * 32 + g * 8;
            const F fk = row_frag<T, D, L::kPitchN>(lds + L::kOffK, x, g, hl);
            const F fq = row_frag<T, D, L::kPitchN>(rQ, x, g, hl);
            const f32x4 st = mma16(fk, fq, z);             // S^T[key 16 kb + 4g + q][query x]
            float e[4], gg[4], hh4[4], p[4], gt[4];
            unpack4<T>(*reinterpret_cast<const uint2*>(pe), e);
            unpack4<T>(*reinterpret_cast<const uint2*>(pg), gg);
            float tmax = -INFINITY;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float sv = st[q] * a.scale + e[q];
                hh4[q] = sv * hs;                          // H_hat (times the branch's DropPath factor) leaves through the E slot
                p[q] = sv + mk[q];                         // (mk = -inf past N)
                gt[q] = fast_sigmoid(gg[q] + mk[q]);
                tmax = fmaxf(tmax, p[q]);
            }
            *reinterpret_cast<uint2*>(pe) = pack4u<T>(hh4);
            const float mnew = fmaxf(mx[i], qmax(tmax));
            const float mref = mnew == -INFINITY ? 0.f : mnew;
            const float alpha = fast_exp(mx[i] - mref);    // (first block: exp(-inf) = 0 on zeros)
            mx[i] = mnew;
            float ps = 0.f;
            f32x4 wv;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                p[q] = fast_exp(p[q] - mref);
                ps += p[q];
                gs[i] += gt[q];
                wv[q] = p[q] * gt[q];
            }
            lsum[i] = lsum[i] * alpha + ps;                // per lane (its keys 4g .. 4g+3 of every block): summed over g at the end
            const F fv = row_frag<T, D, L::kPitchN>(lds + L::kOffV, x, g, hl);
            const f32x4 vt = mma16(fv, id, z);             // V[key][d] in accumulator layout = the A operand of V^T
#pragma unroll
            for (int q = 0; q < 4; ++q) o[i][q] *= alpha;
            o[i] = mma16(pack4<T>(vt), pack4<T>(wv), o[i]);  // O^T[d 4g + q][query x]
        }
        __syncthreads();
        if (a.hhat) ImageIO<T, D, HPW>::store_hhat(lds, a, u, kb, tid);
        __syncthreads();                                   // (the planes are restaged for the next key block)
    }
    const int row = 16 * u.qb + x;
#pragma unroll
    for (int i = 0; i < HPW; ++i) {
        const int hl = w * HPW + i, h = u.hc * L::HW + hl;
        const float sum = qsum(lsum[i]), gsum = qsum(gs[i]);
        const float inv = sum > 0.f ? fast_rcp(sum) : 0.f;
        const float f = (a.scale_degree ? __logf(1.f + gsum) : 1.f) * inv;
        const float ov[4] = {o[i][0] * f, o[i][1] * f, o[i][2] * f, o[i][3] * f};
        if (4 * g < D) *reinterpret_cast<uint2*>(rQ + x * L::kPitchN + hl * (D * 2) + g * 8) = pack4u<T>(ov);   // V_att leaves through this head's Q columns
        if (row < N && g == 0) {
            a.lse[((int64_t)u.b * N + row) * H + h] = mx[i] + __logf(sum);
            a.gsum[((int64_t)u.b * N + row) * H + h] = gsum;
        }
    }
    __syncthreads();
    RowsIO<T, D, HPW, 1>::store(rQ, a.vatt, (int64_t)D * H, 0, 16 * u.qb, a, u, tid);
}

constexpr int kLdsMax = 160 * 1024;

template <typename T, int D, int HPW>
static int launch_fwd(const tgt_node_attention_args& a, hipStream_t st) {
    using L = Lay<D, HPW>;
    constexpr int kLds = L::kFwdBytes;
    if constexpr (kLds > kLdsMax) {
        return -1;
    } else {
        const int nqb = (a.N + 15) / 16, chunks = a.H / L::HW;
        const int grid = ((a.B + 7) / 8) * 8 * chunks * nqb;
        static bool attr_set[16] = {};
        if (!dyn_lds_once(attr_set, reinterpret_cast<const void*>(&node_att_kb_fwd_kernel<T, D, HPW>), kLds))
            return set_error(TGT_ERR_LAUNCH, "node_att_kb_fwd_kernel: cannot reserve %d bytes of LDS", kLds);
        hipLaunchKernelGGL((node_att_kb_fwd_kernel<T, D, HPW>), dim3(grid), dim3(kThreads), kLds, st, a);
        return check_launch("node_att_kb_fwd_kernel");
    }
}
template <typename T>
static int dispatch_fwd(const tgt_node_attention_args& a, hipStream_t st) {
    switch (a.D) {
        case 8: return launch_fwd<T, 8, 4>(a, st);
        case 12: return launch_fwd<T, 12, 4>(a, st);
        case 16: return launch_fwd<T, 16, 4>(a, st);
        default: return -1;
    }
}

}  // namespace nkb

// Shapes the key-blocked forward takes: 16-bit, any N <= 1024 (nothing in the kernel is sized by N), H a multiple of 32, D in {8, 12, 16}.  TGT_NODE_KB (A/B): 0 off, 1 N > 32
// only, 2 (default) every N.
bool node_attention_kb_eligible(const tgt_node_attention_args& a, bool bwd) {
    static const int mode = getenv("TGT_NODE_KB") ? atoi(getenv("TGT_NODE_KB")) : 2;
    if (!mode || bwd || a.logits_only || a.dtype == TGT_F32) return false;
    if (a.N > 1024 || (a.N <= 32 && mode < 2) || a.H % 32 || !(a.D == 8 || a.D == 12 || a.D == 16)) return false;
    if (!a.mask || !a.vatt || !a.lse || !a.gsum) return false;
    // per-graph buffer resources: every in-range byte offset must stay below the out-of-range sentinel kOob (node_tiles16.hpp), or the
    // "invalid lane reads offset kOob" trick could alias a real element for very wide rows
    const int64_t per_graph = (int64_t)a.N * a.N * (a.ld_eg > a.H ? a.ld_eg : a.H) * 2, per_graph_q = (int64_t)a.N * a.ld_qkv * 2;
    if (per_graph >= (int64_t)0x7ffffff0 || per_graph_q >= (int64_t)0x7ffffff0) return false;
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    if (a.ld_qkv % 8 || a.q_off % 8 || a.k_off % 8 || a.v_off % 8 || a.ld_eg % 8 || a.e_off % 8 || a.g_off % 8) return false;
    if (!al16(a.qkv) || !al16(a.eg) || !al16(a.vatt) || (a.hhat && !al16(a.hhat))) return false;
    return true;
}

int node_attention_kb_run(const tgt_node_attention_args& a, bool bwd, hipStream_t st) {
    int e = -1;
    if (!bwd) e = a.dtype == TGT_BF16 ? nkb::dispatch_fwd<bf16_t>(a, st) : nkb::dispatch_fwd<f16_t>(a, st);
    if (e < 0) return set_error(TGT_ERR_UNSUPPORTED, "node attention (key-blocked): unsupported N=%d H=%d D=%d", a.N, a.H, a.D);
    return e;
}

}  // namespace tgt
