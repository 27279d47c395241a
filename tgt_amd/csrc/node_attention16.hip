// Node attention with edge bias and gate (EGT_Attention core) on 16-wide matrix-core tiles: forward and single-pass backward for
// 16-bit dtypes, N <= 64, H a multiple of 8, D in {8, 12, 16} -- BASELINE config 4 (graphs padded to 33..64 nodes).
//
// Replaces reference lib/tgt/layers/layers.py:62-77 (einsum -> +E -> softmax * sigmoid gate -> einsum -> degree scaler) and its
// autograd backward for graphs padded to more than 32 nodes.  Math: SURVEY.md App. A.1 / A.4; same arithmetic as
// node_attention_mfma.hip (which keeps N <= 32) and node_attention.hip (every other shape).
//
// Why a third form.  node_attention_mfma.hip holds ONE 32 x 32 (query, key) tile per head and the whole graph's E / G image of
// 16 heads in LDS: at N = 48 that image would be 147 KB and the tiles 64 x 64 padded (1.78x the pairs).  Here the unit of work is a
// BLOCK OF 16 QUERIES of 8 heads:
//   workgroup = (graph b, 8 heads, [forward: query block qb]), 8 waves, wave = head; 50-70 KB of LDS: 2-3 workgroups per CU, so one
//   workgroup's loads and stores run under another's tile math;
//   LDS image of one query block, KEY-MAJOR PER HEAD: plane[query l][head hh][key m] (E, G, [backward: dH_hat]) -- the staging threads
//   load the 16-byte (8 heads) records of four consecutive keys and transpose them in registers (v_perm), so that a lane's four
//   keys of its head are ONE 8-byte LDS access (read E, read G, write H_hat / dE / dG); node rows likewise [row][head][d];
//   per head: S^T[key][query] per key block = one v_mfma_f32_16x16x16 (K rows x Q rows): lane (query x = l & 15, g = l >> 4)
//   holds keys 4g .. 4g+3 of each key block, so the softmax over keys is in-lane values + two lane exchanges (l ^ 16, l ^ 32); with
//   16x16x16 the accumulator layout (lane = column, rows 4g + q) IS the B-operand layout and the A-operand layout of the
//   transposed tile, so the weights feed P V directly and every other re-layout is one product with the identity.
// The forward runs one workgroup per (graph, head group, query block); the backward one per (graph, head group) that walks the
// query blocks with dK^T / dV^T accumulators in registers (complete sums over all queries: no partial tiles, no atomics), K / V
// rows staged once.  Every lane works on real pairs: N = 48 is 3 x 3 blocks instead of 2 x 2 padded 32-wide tiles.
// HBM-bound on E, G, H_hat, dH_hat (algorithmic traffic only); the bias/softmax path of BASELINE.json's north_star at N > 32.
#include <cstdlib>
#include "node_tiles16.hpp"

namespace tgt {
namespace na16 {

constexpr int HG = 8, kThreads = HG * 64;

// LDS map.  Pitches are 16 bytes past a multiple of 32 with pitch / 16 odd, so the 16 queries (rows) of a half-wave's 8-byte
// accesses fall on 16 different 4-bank groups -- conflict-free per half-wave.
template <int NQ, int D>
struct Lay {
    static constexpr int NK = 16 * NQ, DQ = D / 4;
    static constexpr int kHeadP = NK * 2;                 // bytes of one head's keys in a pair plane
    static constexpr int kPitchP = HG * kHeadP + 16;      // pair plane, per query
    static constexpr int kPitchM = NK * 4 + 16;           // mask tile (fp32), per query
    static constexpr int kHeadN = D * 2;                  // bytes of one head's d in a node row
    static constexpr int kPitchN = HG * kHeadN + 16;      // node rows
    static constexpr int kOffE = 0;
    static constexpr int kOffG = kOffE + 16 * kPitchP;
    static constexpr int kOffM = kOffG + 16 * kPitchP;
    static constexpr int kOffQ = kOffM + 16 * kPitchM;                       // the block's Q rows; V_att / dQ leave through it
    static constexpr int kOffK = kOffQ + 16 * kPitchN;
    static constexpr int kOffV = kOffK + NK * kPitchN;
    static constexpr int kFwdBytes = kOffV + NK * kPitchN;
    static constexpr int kOffO = kFwdBytes;                                  // backward: the block's dV_att rows
    static constexpr int kOffH = kOffO + 16 * kPitchN;                       //           and its dH_hat plane
    static constexpr int kBwdBytes = kOffH + 16 * kPitchP;
    static_assert((kPitchP / 16) % 2 == 1 && (kPitchN / 16) % 2 == 1 && (kPitchM / 16) % 2 == 1, "odd pitches");
};

struct Unit { int b, hg; };

// ---- pair planes.  A staging task = (plane, query l, key quad mq): the 16-byte records of pairs (16 qb + l, 4 mq + i), i < 4.
// `chan[plane]` = element offset of this head group's 8 channels inside a pair's row of the tensor (ld elements per pair).
template <typename T, int NQ, int PLANES>
struct PairIO {
    static constexpr int NK = 16 * NQ, MQ = NK / 4, kPerPlane = 16 * MQ, kTasks = PLANES * kPerPlane;
    static constexpr int kIters = (kTasks + kThreads - 1) / kThreads;
    uint4 v[kIters][4];

    __device__ __forceinline__ void issue(const void* x, int64_t ld, const int (&chan)[PLANES], int N, int b, int qb, int tid) {
        asm volatile("" : "+v"(tid));            // (opaque: the task's addresses are recomputed here, not kept live across the query-block walk)
        const __amdgpu_buffer_rsrc_t rs = graph_rsrc(x, (int64_t)N * N * ld * sizeof(T), b);
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            const int task = it * kThreads + tid, plane = task / kPerPlane, r = task % kPerPlane, l = r / MQ, mq = r % MQ, q = 16 * qb + l;
            int ch = chan[0];
#pragma unroll
            for (int p = 1; p < PLANES; ++p) ch = plane == p ? chan[p] : ch;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = 4 * mq + i;
                const bool ok = x && task < kTasks && q < N && m < N;
                v[it][i] = buf_ld16(rs, ok ? (uint32_t)(((int64_t)(q * N + m) * ld + ch) * (int64_t)sizeof(T)) : kOob);
            }
        }
    }
    template <int PITCH>
    __device__ __forceinline__ void land(char* const (&planes)[PLANES], int tid) {
        asm volatile("" : "+v"(tid));
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            const int task = it * kThreads + tid, plane = task / kPerPlane, r = task % kPerPlane, l = r / MQ, mq = r % MQ;
            char* base = planes[0];
#pragma unroll
            for (int p = 1; p < PLANES; ++p) base = plane == p ? planes[p] : base;
            uint2 o[8];
            tr4x8(v[it], o);
            if (task < kTasks) lds_put8x8(base + l * PITCH + mq * 8, NK * 2, o);
        }
    }
    // LDS planes -> the tensor
    template <int PITCH>
    static __device__ __forceinline__ void store(const char* const (&planes)[PLANES], void* x, int64_t ld, const int (&chan)[PLANES], int N, int b,
                                                 int qb, int tid) {
        asm volatile("" : "+v"(tid));            // (opaque: the task's addresses are recomputed here, not kept live across the query-block walk)
        const __amdgpu_buffer_rsrc_t rs = graph_rsrc(x, (int64_t)N * N * ld * sizeof(T), b);
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            const int task = it * kThreads + tid, plane = task / kPerPlane, r = task % kPerPlane, l = r / MQ, mq = r % MQ, q = 16 * qb + l;
            const char* base = planes[0];
            int ch = chan[0];
#pragma unroll
            for (int p = 1; p < PLANES; ++p) { base = plane == p ? planes[p] : base; ch = plane == p ? chan[p] : ch; }
            if (task < kTasks) {
                uint2 o[8];
                lds_get8x8(base + l * PITCH + mq * 8, NK * 2, o);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = 4 * mq + i;
                    const bool ok = q < N && m < N;
                    buf_st16(rs, ok ? (uint32_t)(((int64_t)(q * N + m) * ld + ch) * (int64_t)sizeof(T)) : kOob, tr8x4_row(o, i));
                }
            }
        }
    }
};

// ---- node rows.  A staging task = (segment, row, d quad dq): the 16-byte (8 heads) records of (row, d = 4 dq + i), i < 4, of a
// (B, N, ld) tensor whose row holds [d][H heads] from element `off[segment]`.  Segment s covers `rows[s]` rows from row0[s] into region[s].
template <typename T, int D, int SEGS, int MAXROWS>
struct NodeIO {
    static constexpr int DQ = D / 4, kTasks = MAXROWS * DQ, kIters = (kTasks + kThreads - 1) / kThreads;
    uint4 v[kIters][4];

    static __device__ __forceinline__ void decode(int task, const int (&rows)[SEGS], int& seg, int& row, int& dq) {
        int r = task / DQ;
        dq = task % DQ;
        seg = 0;
#pragma unroll
        for (int s = 0; s + 1 < SEGS; ++s)
            if (seg == s && r >= rows[s]) { r -= rows[s]; seg = s + 1; }
        row = r;
    }
    __device__ __forceinline__ void issue(const void* x, int64_t ld, const int (&off)[SEGS], const int (&rows)[SEGS], const int (&row0)[SEGS], int N,
                                          int H, const Unit& u, int tid) {
        asm volatile("" : "+v"(tid));            // (opaque: the task's addresses are recomputed here, not kept live across the query-block walk)
        const __amdgpu_buffer_rsrc_t rs = graph_rsrc(x, (int64_t)N * ld * sizeof(T), u.b);
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            int seg, row, dq;
            decode(it * kThreads + tid, rows, seg, row, dq);
            int o = off[0], r0 = row0[0], nr = rows[0];
#pragma unroll
            for (int s = 1; s < SEGS; ++s) { o = seg == s ? off[s] : o; r0 = seg == s ? row0[s] : r0; nr = seg == s ? rows[s] : nr; }
            const bool ok = x && row < nr && r0 + row < N;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                v[it][i] = buf_ld16(rs, ok ? (uint32_t)(((int64_t)(r0 + row) * ld + o + (4 * dq + i) * H + u.hg * HG) * (int64_t)sizeof(T)) : kOob);
        }
    }
    template <int PITCH>
    __device__ __forceinline__ void land(char* const (&region)[SEGS], const int (&rows)[SEGS], int tid) {
        asm volatile("" : "+v"(tid));
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            int seg, row, dq;
            decode(it * kThreads + tid, rows, seg, row, dq);
            char* base = region[0];
            int nr = rows[0];
#pragma unroll
            for (int s = 1; s < SEGS; ++s) { base = seg == s ? region[s] : base; nr = seg == s ? rows[s] : nr; }
            uint2 o[8];
            tr4x8(v[it], o);
            if (row < nr) lds_put8x8(base + row * PITCH + dq * 8, D * 2, o);
        }
    }
    template <int PITCH>
    static __device__ __forceinline__ void store(const char* const (&region)[SEGS], void* x, int64_t ld, const int (&off)[SEGS], const int (&rows)[SEGS],
                                                 const int (&row0)[SEGS], int N, int H, const Unit& u, int tid) {
        asm volatile("" : "+v"(tid));            // (opaque: the task's addresses are recomputed here, not kept live across the query-block walk)
        const __amdgpu_buffer_rsrc_t rs = graph_rsrc(x, (int64_t)N * ld * sizeof(T), u.b);
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            int seg, row, dq;
            decode(it * kThreads + tid, rows, seg, row, dq);
            const char* base = region[0];
            int o = off[0], r0 = row0[0], nr = rows[0];
#pragma unroll
            for (int s = 1; s < SEGS; ++s) {
                base = seg == s ? region[s] : base; o = seg == s ? off[s] : o; r0 = seg == s ? row0[s] : r0; nr = seg == s ? rows[s] : nr;
            }
            if (row < nr) {
                uint2 t[8];
                lds_get8x8(base + row * PITCH + dq * 8, D * 2, t);
                const bool ok = r0 + row < N;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    buf_st16(rs, ok ? (uint32_t)(((int64_t)(r0 + row) * ld + o + (4 * dq + i) * H + u.hg * HG) * (int64_t)sizeof(T)) : kOob,
                             tr8x4_row(t, i));
            }
        }
    }
};

// mask tile of the query block: pairs past N get -inf (weight exactly 0, gate sigmoid(-inf) = 0)
template <int NQ, int PITCH>
__device__ __forceinline__ void mask_load(char* lds_m, const tgt_node_attention_args& a, int b, int qb, int tid) {
    constexpr int NK = 16 * NQ, MQ = NK / 4;
    const int N = a.N;
    asm volatile("" : "+v"(tid));
    const __amdgpu_buffer_rsrc_t rs = graph_rsrc(a.mask, (int64_t)N * N * 4, b);
    for (int task = tid; task < 16 * MQ; task += kThreads) {
        const int l = task / MQ, mq = task % MQ, q = 16 * qb + l;
        float mk[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = 4 * mq + i;
            const bool ok = q < N && m < N;
            const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, ok ? (q * N + m) * 4 : (int)kOob, 0, 0));
            mk[i] = ok ? v : -INFINITY;
        }
        *reinterpret_cast<float4*>(lds_m + l * PITCH + mq * 16) = make_float4(mk[0], mk[1], mk[2], mk[3]);
    }
}

// operand fragment of head hh from a node region: lane (x, g) holds X[row][d = 4g + t], t = 0..3 (0 past D)
template <typename T, int D, int PITCH>
__device__ __forceinline__ frag4_t<T> node_frag(const char* region, int row, int g, int hh) {
    const int gg = 4 * g < D ? g : 0;
    uint2 u = *reinterpret_cast<const uint2*>(region + row * PITCH + hh * (D * 2) + gg * 8);
    if (4 * g >= D) u = make_uint2(0u, 0u);
    frag4_t<T> f;
    __builtin_memcpy(&f, &u, 8);
    return f;
}
// transposed result X^T[d = 4g + q][row] into head hh of a node region
template <typename T, int D, int PITCH>
__device__ __forceinline__ void node_put(char* region, const f32x4& acc, int row, int g, int hh) {
    const float v[4] = {acc[0], acc[1], acc[2], acc[3]};
    if (4 * g < D) *reinterpret_cast<uint2*>(region + row * PITCH + hh * (D * 2) + g * 8) = pack4u<T>(v);
}

// ---------------------------------------------------------------------------
// forward tile of head hh for query block qb: H_hat into the E slots, V_att into the head's columns of the Q region
// ---------------------------------------------------------------------------
template <typename T, int NQ, int D>
__device__ __forceinline__ void tile_fwd(char* lds, const tgt_node_attention_args& a, const Unit& u, int qb, int x, int g, int hh) {
    using L = Lay<NQ, D>;
    using F = frag4_t<T>;
    const int N = a.N, H = a.H, h = u.hg * HG + hh, row = 16 * qb + x;
    char* rQ = lds + L::kOffQ;
    const char* rK = lds + L::kOffK;
    const char* rV = lds + L::kOffV;
    const F fq = node_frag<T, D, L::kPitchN>(rQ, x, g, hh);
    const F id = ident4<T>(x, g);
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const float hs = a.hhat_scale ? a.hhat_scale[u.b] : 1.f;
    float s[NQ][4], gt[NQ][4], mx = -INFINITY;
    char* pe = lds + L::kOffE + x * L::kPitchP + hh * L::kHeadP + g * 8;
    const char* pg = lds + L::kOffG + x * L::kPitchP + hh * L::kHeadP + g * 8;
    const char* pm = lds + L::kOffM + x * L::kPitchM + g * 16;
#pragma unroll
    for (int kb = 0; kb < NQ; ++kb) {
        const F fk = node_frag<T, D, L::kPitchN>(rK, 16 * kb + x, g, hh);
        const f32x4 st = mma16(fk, fq, z);                 // S^T[key 16 kb + 4g + q][query x]
        float e[4], gg[4], hh4[4];
        unpack4<T>(*reinterpret_cast<const uint2*>(pe + kb * 32), e);
        unpack4<T>(*reinterpret_cast<const uint2*>(pg + kb * 32), gg);
        const float4 mk4 = *reinterpret_cast<const float4*>(pm + kb * 64);
        const float mk[4] = {mk4.x, mk4.y, mk4.z, mk4.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float sv = st[q] * a.scale + e[q];
            hh4[q] = sv * hs;                              // H_hat (times the branch's DropPath factor) leaves through the E slot
            const float xx = sv + mk[q];                   // (mk = -inf past N)
            gt[kb][q] = fast_sigmoid(gg[q] + mk[q]);
            s[kb][q] = xx;
            mx = fmaxf(mx, xx);
        }
        *reinterpret_cast<uint2*>(pe + kb * 32) = pack4u<T>(hh4);
    }
    mx = qmax(mx);
    const float mref = mx == -INFINITY ? 0.f : mx;
    float sum = 0.f, gsum = 0.f;
#pragma unroll
    for (int kb = 0; kb < NQ; ++kb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            s[kb][q] = fast_exp(s[kb][q] - mref);
            sum += s[kb][q];
            gsum += gt[kb][q];
        }
    sum = qsum(sum);
    gsum = qsum(gsum);
    const float inv = sum > 0.f ? fast_rcp(sum) : 0.f;
    f32x4 o = z;
#pragma unroll
    for (int kb = 0; kb < NQ; ++kb) {
        const F fv = node_frag<T, D, L::kPitchN>(rV, 16 * kb + x, g, hh);
        const f32x4 vt = mma16(fv, id, z);                 // V[key][d] in accumulator layout = the A operand of V^T
        f32x4 w;
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q] = s[kb][q] * inv * gt[kb][q];
        o = mma16(pack4<T>(vt), pack4<T>(w), o);           // O^T[d 4g + q][query x]
    }
    const float f = a.scale_degree ? __logf(1.f + gsum) : 1.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] *= f;
    node_put<T, D, L::kPitchN>(rQ, o, x, g, hh);           // V_att leaves through this head's Q columns (only this wave reads them)
    if (row < N && g == 0) {
        a.lse[((int64_t)u.b * N + row) * H + h] = mx + __logf(sum);
        a.gsum[((int64_t)u.b * N + row) * H + h] = gsum;
    }
}

// ---------------------------------------------------------------------------
// backward tile of head hh for query block qb: dE, dG into the E / G slots, dQ into the head's columns of the Q region,
// dK^T / dV^T of every key block accumulated in registers (written after the last query block).  With P, the gates and
// their sum recomputed (node_attention_mfma.hip: tile_bwd):
//   dA^T[m][l] = dsc_l * V[m,:].dV_att[l,:]      dsc = log(1 + sum_m g)   (1 without the degree scaler)
//   delta_l = sum_m P dA g        d_dsc = delta / dsc        dgsum = d_dsc / (1 + sum g)
//   dS = P (dA g - delta)         dG = (dA P + dgsum) g (1 - g)            dE = dH_hat + dS
//   dQ^T = s K^T dE^T             dK^T = s Q^T dE                          dV^T = dV_att^T (P g dsc)
// ---------------------------------------------------------------------------
template <typename T, int NQ, int D>
__device__ __forceinline__ void tile_bwd(char* lds, const tgt_node_attention_args& a, const Unit& u, int x, int g, int hh,
                                         f32x4 (&dk)[NQ], f32x4 (&dv)[NQ]) {
    using L = Lay<NQ, D>;
    using F = frag4_t<T>;
    char* rQ = lds + L::kOffQ;
    const char* rK = lds + L::kOffK;
    const char* rV = lds + L::kOffV;
    const char* rO = lds + L::kOffO;
    const F fq = node_frag<T, D, L::kPitchN>(rQ, x, g, hh), fo = node_frag<T, D, L::kPitchN>(rO, x, g, hh);
    const F id = ident4<T>(x, g);
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    float s[NQ][4], gt[NQ][4], da[NQ][4], mx = -INFINITY;
    char* pe = lds + L::kOffE + x * L::kPitchP + hh * L::kHeadP + g * 8;
    char* pg = lds + L::kOffG + x * L::kPitchP + hh * L::kHeadP + g * 8;
    const char* ph = lds + L::kOffH + x * L::kPitchP + hh * L::kHeadP + g * 8;
    const char* pm = lds + L::kOffM + x * L::kPitchM + g * 16;
#pragma unroll
    for (int kb = 0; kb < NQ; ++kb) {
        const F fk = node_frag<T, D, L::kPitchN>(rK, 16 * kb + x, g, hh);
        const F fv = node_frag<T, D, L::kPitchN>(rV, 16 * kb + x, g, hh);
        const f32x4 st = mma16(fk, fq, z);                 // S^T[key][query]
        const f32x4 dt = mma16(fv, fo, z);                 // (V dV_att^T)[key][query]
        float e[4], gg[4];
        unpack4<T>(*reinterpret_cast<const uint2*>(pe + kb * 32), e);
        unpack4<T>(*reinterpret_cast<const uint2*>(pg + kb * 32), gg);
        const float4 mk4 = *reinterpret_cast<const float4*>(pm + kb * 64);
        const float mk[4] = {mk4.x, mk4.y, mk4.z, mk4.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float xx = st[q] * a.scale + e[q] + mk[q];        // (mk = -inf past N)
            gt[kb][q] = fast_sigmoid(gg[q] + mk[q]);
            s[kb][q] = xx;
            da[kb][q] = dt[q];
            mx = fmaxf(mx, xx);
        }
    }
    mx = qmax(mx);
    if (mx == -INFINITY) mx = 0.f;                        // padding query: every weight exactly 0
    float sum = 0.f, gsum = 0.f;
#pragma unroll
    for (int kb = 0; kb < NQ; ++kb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            s[kb][q] = fast_exp(s[kb][q] - mx);
            sum += s[kb][q];
            gsum += gt[kb][q];
        }
    sum = qsum(sum);
    gsum = qsum(gsum);
    const float inv = sum > 0.f ? fast_rcp(sum) : 0.f;
    const float dsc = a.scale_degree ? __logf(1.f + gsum) : 1.f;
    float delta = 0.f;
#pragma unroll
    for (int kb = 0; kb < NQ; ++kb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            s[kb][q] *= inv;                              // P
            da[kb][q] *= dsc;                             // dA (gradient wrt the unscaled V_att folded in)
            delta += s[kb][q] * da[kb][q] * gt[kb][q];
        }
    delta = qsum(delta);
    const float d_dsc = dsc != 0.f ? delta * fast_rcp(dsc) : 0.f;      // zero scaler <=> every gate 0 <=> V_att 0
    const float dgsum = a.scale_degree ? d_dsc * fast_rcp(1.f + gsum) : 0.f;
    const float hs = a.hhat_scale ? a.hhat_scale[u.b] : 1.f;           // d_hhat is the gradient of hhat_scale * H_hat

    // per-pair gradients; dE (times the logit scale) and the weights P g dsc stay in registers as the MFMA operands of this block
    float amax = 0.f;
#pragma unroll
    for (int kb = 0; kb < NQ; ++kb) {
        float dh[4], dE[4], dG[4];
        unpack4<T>(*reinterpret_cast<const uint2*>(ph + kb * 32), dh);  // (zeros when there is no d_hhat)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float p = s[kb][q], gg = gt[kb][q];
            const float dS = p * (da[kb][q] * gg - delta);
            dG[q] = (da[kb][q] * p + dgsum) * gg * (1.f - gg);
            dE[q] = dh[q] * hs + dS;
            da[kb][q] = p * gg * dsc;                                   // the weights (operand of dV)
            s[kb][q] = dE[q] * a.scale;                                 // the logit gradient (operand of dQ, dK)
            amax = fmaxf(amax, fabsf(s[kb][q]));
        }
        *reinterpret_cast<uint2*>(pe + kb * 32) = pack4u<T>(dE);        // dE, dG leave through the E / G slots of this lane
        *reinterpret_cast<uint2*>(pg + kb * 32) = pack4u<T>(dG);
    }
    float c = 1.f, unscale = 1.f;
    if constexpr (!kIsBf16<T>) {
        // fp16 operands: bring the head's largest |dE| of this query block to 2^13 (an exact power of two, undone on dQ / dK) so that
        // small gradients do not sink into fp16 subnormals on their way through the matrix core (node_attention_mfma.hip)
        amax = group_max<64>(amax);
        const int ex = (int)((__builtin_bit_cast(uint32_t, amax) >> 23) & 0xffu);
        if (ex >= 14 && ex <= 253) {
            c = __builtin_bit_cast(float, (uint32_t)(267 - ex) << 23);                     // 2^(13 - (ex - 127))
            unscale = __builtin_bit_cast(float, (uint32_t)(ex - 13) << 23);                // 1 / c
        }
    }
    const f32x4 qt = mma16(fq, id, z), ot = mma16(fo, id, z);          // Q / dV_att rows in accumulator layout = A operands of Q^T / dV_att^T
    const F fqt = pack4<T>(qt), fot = pack4<T>(ot);
    f32x4 dq = z;
#pragma unroll
    for (int kb = 0; kb < NQ; ++kb) {
        f32x4 zs, ws;
#pragma unroll
        for (int q = 0; q < 4; ++q) { zs[q] = s[kb][q] * c; ws[q] = da[kb][q]; }
        const F zf = pack4<T>(zs), wf = pack4<T>(ws);                   // dE^T / W^T [key 4g + q][query x]: B operands as they are
        const F fk = node_frag<T, D, L::kPitchN>(rK, 16 * kb + x, g, hh);  // (read again: 8 bytes of LDS against a register pair held across the tile)
        const f32x4 kt = mma16(fk, id, z);                              // K[key][d] in accumulator layout = the A operand of K^T
        dq = mma16(pack4<T>(kt), zf, dq);                               // dQ^T[d][query] += K^T[d][key] dE^T[key][query]
        // the same registers are the A operand of the TRANSPOSED tile: (dE)[query][key] = (dE^T)^T . I in accumulator layout
        const f32x4 zT = mma16(zf, id, z), wT = mma16(wf, id, z);
        f32x4 t = mma16(fqt, pack4<T>(zT), z);                          // dK^T[d][key] of this query block
        if constexpr (!kIsBf16<T>) {
#pragma unroll
            for (int q = 0; q < 4; ++q) t[q] *= unscale;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) dk[kb][q] += t[q];
        dv[kb] = mma16(fot, pack4<T>(wT), dv[kb]);                      // dV^T[d][key] += dV_att^T[d][query] W[query][key]
    }
    if constexpr (!kIsBf16<T>) {
#pragma unroll
        for (int q = 0; q < 4; ++q) dq[q] *= unscale;
    }
    node_put<T, D, L::kPitchN>(rQ, dq, x, g, hh);         // dQ leaves through this head's Q columns
}

template <typename T, int NQ, int D>
__global__ void __launch_bounds__(kThreads, 6) node_att16_fwd_kernel(const tgt_node_attention_args a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    using L = Lay<NQ, D>;
    const int tid = threadIdx.x, lane = tid & 63, hh = tid >> 6, x = lane & 15, g = lane >> 4;
    const int nqb = (a.N + 15) / 16, groups = a.H / HG;
    int b, sub;
    if (!unit_of_block(a.B, groups * nqb, b, sub)) return;
    const Unit u{b, sub % groups};
    const int qb = sub / groups;
    {
        PairIO<T, NQ, 2> pio;
        const int chan[2] = {a.e_off + u.hg * HG, a.g_off + u.hg * HG};
        pio.issue(a.eg, a.ld_eg, chan, a.N, b, qb, tid);
        NodeIO<T, D, 3, 16 + 2 * L::NK> nio;
        const int off[3] = {a.q_off, a.k_off, a.v_off}, rows[3] = {16, L::NK, L::NK}, row0[3] = {16 * qb, 0, 0};
        nio.issue(a.qkv, a.ld_qkv, off, rows, row0, a.N, a.H, u, tid);
        mask_load<NQ, L::kPitchM>(lds + L::kOffM, a, b, qb, tid);
        char* const planes[2] = {lds + L::kOffE, lds + L::kOffG};
        pio.template land<L::kPitchP>(planes, tid);
        char* const regions[3] = {lds + L::kOffQ, lds + L::kOffK, lds + L::kOffV};
        nio.template land<L::kPitchN>(regions, rows, tid);
    }
    __syncthreads();
    tile_fwd<T, NQ, D>(lds, a, u, qb, x, g, hh);
    __syncthreads();
    if (a.hhat) {
        const char* const planes[1] = {lds + L::kOffE};
        const int chan[1] = {u.hg * HG};
        PairIO<T, NQ, 1>::template store<L::kPitchP>(planes, a.hhat, a.H, chan, a.N, b, qb, tid);
    }
    {
        const char* const regions[1] = {lds + L::kOffQ};
        const int off[1] = {0}, rows[1] = {16}, row0[1] = {16 * qb};
        NodeIO<T, D, 1, 16>::template store<L::kPitchN>(regions, a.vatt, (int64_t)D * a.H, off, rows, row0, a.N, a.H, u, tid);
    }
}

template <typename T, int NQ, int D>
__global__ void __launch_bounds__(kThreads, NQ == 4 ? 2 : 4) node_att16_bwd_kernel(const tgt_node_attention_args a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    using L = Lay<NQ, D>;
    const int tid = threadIdx.x, lane = tid & 63, hh = tid >> 6, x = lane & 15, g = lane >> 4;
    const int nqb = (a.N + 15) / 16, groups = a.H / HG;
    int b, sub;
    if (!unit_of_block(a.B, groups, b, sub)) return;
    const Unit u{b, sub};
    const int chan_eg[2] = {a.e_off + u.hg * HG, a.g_off + u.hg * HG}, chan_h[1] = {u.hg * HG};
    {
        NodeIO<T, D, 2, 2 * L::NK> nio;
        const int off[2] = {a.k_off, a.v_off}, rows[2] = {L::NK, L::NK}, row0[2] = {0, 0};
        nio.issue(a.qkv, a.ld_qkv, off, rows, row0, a.N, a.H, u, tid);
        char* const regions[2] = {lds + L::kOffK, lds + L::kOffV};
        nio.template land<L::kPitchN>(regions, rows, tid);
    }
    f32x4 dk[NQ], dv[NQ];
#pragma unroll
    for (int kb = 0; kb < NQ; ++kb) dk[kb] = dv[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int qb = 0; qb < nqb; ++qb) {
        {
            PairIO<T, NQ, 2> pio;
            pio.issue(a.eg, a.ld_eg, chan_eg, a.N, b, qb, tid);
            PairIO<T, NQ, 1> hio;
            hio.issue(a.d_hhat, a.H, chan_h, a.N, b, qb, tid);
            char* const planes[2] = {lds + L::kOffE, lds + L::kOffG};
            pio.template land<L::kPitchP>(planes, tid);
            char* const hplane[1] = {lds + L::kOffH};
            hio.template land<L::kPitchP>(hplane, tid);
        }
        {
            const int off1[1] = {a.q_off}, rows1[1] = {16}, row01[1] = {16 * qb}, off0[1] = {0};
            NodeIO<T, D, 1, 16> qio, oio;
            qio.issue(a.qkv, a.ld_qkv, off1, rows1, row01, a.N, a.H, u, tid);
            oio.issue(a.d_vatt, (int64_t)D * a.H, off0, rows1, row01, a.N, a.H, u, tid);
            mask_load<NQ, L::kPitchM>(lds + L::kOffM, a, b, qb, tid);
            char* const rq[1] = {lds + L::kOffQ};
            char* const ro[1] = {lds + L::kOffO};
            qio.template land<L::kPitchN>(rq, rows1, tid);
            oio.template land<L::kPitchN>(ro, rows1, tid);
        }
        __syncthreads();
        tile_bwd<T, NQ, D>(lds, a, u, x, g, hh, dk, dv);
        __syncthreads();
        {
            const char* const planes[2] = {lds + L::kOffE, lds + L::kOffG};
            PairIO<T, NQ, 2>::template store<L::kPitchP>(planes, a.d_eg, a.ld_eg, chan_eg, a.N, b, qb, tid);
            const char* const rq[1] = {lds + L::kOffQ};
            const int off1[1] = {a.q_off}, rows1[1] = {16}, row01[1] = {16 * qb};
            NodeIO<T, D, 1, 16>::template store<L::kPitchN>(rq, a.d_qkv, a.ld_qkv, off1, rows1, row01, a.N, a.H, u, tid);
        }
        __syncthreads();                                   // (the image is restaged for the next query block)
    }
    // dK^T / dV^T [d 4g + q][key 16 kb + x] of this head: complete sums over every query of the graph
#pragma unroll
    for (int kb = 0; kb < NQ; ++kb) {
        node_put<T, D, L::kPitchN>(lds + L::kOffK, dk[kb], 16 * kb + x, g, hh);
        node_put<T, D, L::kPitchN>(lds + L::kOffV, dv[kb], 16 * kb + x, g, hh);
    }
    __syncthreads();
    {
        const char* const regions[2] = {lds + L::kOffK, lds + L::kOffV};
        const int off[2] = {a.k_off, a.v_off}, rows[2] = {L::NK, L::NK}, row0[2] = {0, 0};
        NodeIO<T, D, 2, 2 * L::NK>::template store<L::kPitchN>(regions, a.d_qkv, a.ld_qkv, off, rows, row0, a.N, a.H, u, tid);
    }
}

constexpr int kLdsMax = 160 * 1024;

template <typename T, int NQ, int D>
static int launch(const tgt_node_attention_args& a, bool bwd, hipStream_t st) {
    using L = Lay<NQ, D>;
    const int nqb = (a.N + 15) / 16, groups = a.H / HG;
    if (!bwd) {
        constexpr int kLds = L::kFwdBytes;
        static_assert(kLds <= kLdsMax, "forward LDS");
        const int grid = ((a.B + 7) / 8) * 8 * groups * nqb;
        static bool attr_set[16] = {};
        if (!dyn_lds_once(attr_set, reinterpret_cast<const void*>(&node_att16_fwd_kernel<T, NQ, D>), kLds))
            return set_error(TGT_ERR_LAUNCH, "node_att16_fwd_kernel: cannot reserve %d bytes of LDS", kLds);
        hipLaunchKernelGGL((node_att16_fwd_kernel<T, NQ, D>), dim3(grid), dim3(kThreads), kLds, st, a);
        return check_launch("node_att16_fwd_kernel");
    } else {
        constexpr int kLds = L::kBwdBytes;
        static_assert(kLds <= kLdsMax, "backward LDS");
        const int grid = ((a.B + 7) / 8) * 8 * groups;
        static bool attr_set[16] = {};
        if (!dyn_lds_once(attr_set, reinterpret_cast<const void*>(&node_att16_bwd_kernel<T, NQ, D>), kLds))
            return set_error(TGT_ERR_LAUNCH, "node_att16_bwd_kernel: cannot reserve %d bytes of LDS", kLds);
        hipLaunchKernelGGL((node_att16_bwd_kernel<T, NQ, D>), dim3(grid), dim3(kThreads), kLds, st, a);
        return check_launch("node_att16_bwd_kernel");
    }
}

template <typename T, int NQ>
static int dispatch_d(const tgt_node_attention_args& a, bool bwd, hipStream_t st) {
    switch (a.D) {
        case 8: return launch<T, NQ, 8>(a, bwd, st);
        case 12: return launch<T, NQ, 12>(a, bwd, st);
        case 16: return launch<T, NQ, 16>(a, bwd, st);
        default: return -1;
    }
}
template <typename T>
static int dispatch(const tgt_node_attention_args& a, bool bwd, hipStream_t st) {
    switch ((a.N + 15) / 16) {
        case 1: return dispatch_d<T, 1>(a, bwd, st);
        case 2: return dispatch_d<T, 2>(a, bwd, st);
        case 3: return dispatch_d<T, 3>(a, bwd, st);
        case 4: return dispatch_d<T, 4>(a, bwd, st);
        default: return -1;
    }
}

}  // namespace na16

// Is this call one of the shapes the 16-wide kernels take?  Default: 33 <= N <= 64 (N <= 32 stays on node_attention_mfma.hip);
// TGT_NODE_MFMA16=2 takes every N <= 64 (A/B), 0 none.
bool node_attention16_eligible(const tgt_node_attention_args& a, bool bwd) {
    static const int mode = getenv("TGT_NODE_MFMA16") ? atoi(getenv("TGT_NODE_MFMA16")) : 1;
    if (!mode || a.logits_only || a.dtype == TGT_F32) return false;
    if (a.N > 64 || (a.N <= 32 && mode < 2) || a.H % na16::HG || !(a.D == 8 || a.D == 12 || a.D == 16)) return false;
    if (!a.mask || !a.vatt || !a.lse || !a.gsum) return false;
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    if (a.ld_qkv % 8 || a.q_off % 8 || a.k_off % 8 || a.v_off % 8 || a.ld_eg % 8 || a.e_off % 8 || a.g_off % 8) return false;
    if (!al16(a.qkv) || !al16(a.eg) || !al16(a.vatt) || (a.hhat && !al16(a.hhat))) return false;
    if (bwd && (!al16(a.d_qkv) || !al16(a.d_eg) || !al16(a.d_vatt) || (a.d_hhat && !al16(a.d_hhat)))) return false;
    return true;
}

// returns TGT_OK / an error; call only when node_attention16_eligible()
int node_attention16_run(const tgt_node_attention_args& a, bool bwd, hipStream_t st) {
    int e = a.dtype == TGT_BF16 ? na16::dispatch<bf16_t>(a, bwd, st) : na16::dispatch<f16_t>(a, bwd, st);
    if (e < 0) return set_error(TGT_ERR_UNSUPPORTED, "node attention (16-wide tiles): unsupported N=%d D=%d", a.N, a.D);
    return e;
}

}  // namespace tgt
