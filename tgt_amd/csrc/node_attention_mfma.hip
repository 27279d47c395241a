// Node attention with edge bias and gate (EGT_Attention core) on the matrix core: forward and a
// SINGLE-PASS backward for the 16-bit hot shapes (N <= 32, D <= 16, H a multiple of 16, the
// reference's head-minor channel order).  Same arithmetic as node_attention.hip, which keeps every
// other shape (fp32, N > 32, odd head counts, head-major rows, the logits-only EdgeUpdate).
//
// Replaces reference lib/tgt/layers/layers.py:62-77 (einsum -> +E -> softmax * sigmoid gate ->
// einsum -> degree scaler) and its autograd backward.  Math: SURVEY.md App. A.1 / A.4.
//
// Why a second implementation.  The lane <-> head kernels of node_attention.hip stream E, G and
// H_hat perfectly coalesced, but (1) every lane walks the keys with 2*D four-byte K/V fetches per
// key -- they are bound by vector-memory instructions, not bytes -- and (2) the backward needs two
// passes (rows: dE, dG, dQ; columns: dK, dV) that together move 2.17x the algorithmic bytes
// (profiles/r02_pmc_summary.json).  Per (graph, head) the op is ONE 32x32 attention tile with a
// depth-D contraction: exactly the tile of the triplet kernels (triplet_attention.hip).  Here:
//   workgroup = (graph b, group of HG heads), HG waves, wave = head (HG = 16: one workgroup per CU, 32-byte row
//   segments; HG = 8 for the shape whose 16-head image does not fit the LDS).
//   1. the workgroup pulls everything it needs through LDS ONCE with 16-byte accesses: the 2*HG-byte
//      segments (HG heads) of every E / G / dH_hat row of the graph, the mask, and the HG-head
//      segments of the Q / K / V / dV_att rows;
//   2. each wave gathers its head's operand fragments and its (query, key) tile in accumulator
//      layout from LDS (2-byte reads, bank-conflict free by the +4-byte row pitch), runs the tile
//      on the matrix core -- S^T = K Q^T, softmax over keys in-lane + one half-wave exchange,
//      gate, degree scaler; backward: dA^T = V dV_att^T, dS, dG, then dQ, dK, dV with the
//      identity-MFMA re-layouts of the triplet backward -- and writes its results back into the
//      SAME LDS words it read (its own head's column: no cross-wave hazard);
//   3. the workgroup stores H_hat / V_att (forward) or dE, dG, dQ, dK, dV (backward) as 16-byte
//      segments.
// The backward recomputes the softmax statistics and the gate sum inside the tile (nothing but
// Q, K, V, E, G, mask is needed from the forward) and produces dK / dV in the same pass:
// algorithmic traffic only.  HBM-bound; the workgroups that share a graph's 128-byte E/G
// rows are given the same XCD (block index -> unit map) so that all but one of them hit in its L2.
#include <cstdlib>
#include "common.hpp"
#include "triplet_common.hpp"

namespace tgt {

namespace nmf {

template <typename T> constexpr bool kIsBf16 = false;
template <> constexpr bool kIsBf16<bf16_t> = true;

// LDS map (bytes) of a workgroup of HG heads (= waves).  Row pitches are 4 bytes past a multiple of 128: lanes
// that differ in the query / node row hit consecutive banks in the 2-byte tile reads (ds_read_u16: 32 banks,
// 32-lane groups).
template <int HG>
struct Lay {
    static constexpr int kThreads = HG * 64;
    static constexpr int kRecEG = HG * 4, kRecH = HG * 2;     // bytes per pair: [E HG heads | G HG heads]; dH_hat HG heads
    static constexpr int kPitchEG = 32 * kRecEG + 4;          // per query l: 32 keys
    static constexpr int kPitchH = 32 * kRecH + 4;
    static constexpr int kPitchM = 32 * 4 + 4;                // per query l: 32 mask floats
    static constexpr int kOffEG = 0;
    static constexpr int kOffM = kOffEG + 32 * kPitchEG;
    static constexpr int kOffH = kOffM + 32 * kPitchM;        // backward only
    static constexpr int pitch_n(int D) { return D * kRecH + 4; }                       // per node row: D x HG heads
    static constexpr int off_n(int D, bool bwd, int t) { return (bwd ? kOffH + 32 * kPitchH : kOffH) + t * 32 * pitch_n(D); }
    static constexpr int lds_bytes(int D, bool bwd) { return off_n(D, bwd, bwd ? 4 : 3); }
};

// 16 bytes <-> LDS at a 4-byte-aligned address.  Consecutive lanes handle consecutive 16-byte chunks;
// moving dword (i + lane/8) & 3 in step i spreads a 32-lane group over all 32 banks.  The rotation is
// two conditional-move stages (no branches, static register indices).
__device__ __forceinline__ void rot4(uint32_t (&c)[4], const uint4& v, int rot) {     // c[i] = v[(i + rot) & 3]
    const bool r1 = rot & 1, r2 = rot & 2;
    const uint32_t b0 = r1 ? v.y : v.x, b1 = r1 ? v.z : v.y, b2 = r1 ? v.w : v.z, b3 = r1 ? v.x : v.w;
    c[0] = r2 ? b2 : b0; c[1] = r2 ? b3 : b1; c[2] = r2 ? b0 : b2; c[3] = r2 ? b1 : b3;
}
__device__ __forceinline__ void lds_put16(char* p, const uint4& v, int rot) {
    uint32_t c[4];
    rot4(c, v, rot);
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<uint32_t*>(p + (((i + rot) & 3) << 2)) = c[i];
}
__device__ __forceinline__ uint4 lds_get16(const char* p, int rot) {
    uint32_t x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = *reinterpret_cast<const uint32_t*>(p + (((i + rot) & 3) << 2));     // x[i] = dword (i + rot) & 3
    uint32_t c[4];
    rot4(c, make_uint4(x[0], x[1], x[2], x[3]), (4 - rot) & 3);                                            // c[k] = x[(k - rot) & 3] = dword k
    return make_uint4(c[0], c[1], c[2], c[3]);
}
// buffer-addressed 16-byte accesses: 32-bit byte offsets inside one graph of a tensor; an offset of kOob
// fails the range check of the resource (loads return 0, stores are dropped) -- no branches
constexpr uint32_t kOob = 0x7ffffff0u;
__device__ __forceinline__ uint4 buf_ld16(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void buf_st16(__amdgpu_buffer_rsrc_t r, uint32_t off, const uint4& v) {
    const u32x4_t d = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(d, r, (int)off, 0, TGT_NT_NODE ? TGT_ST_AUX : 0);
}

struct Unit { int b, hg; };
// ---- cooperative staging (NTHR threads) ---------------------------------------------------------
// Pair tensors (B,N,N,ld): a pair's record is SUBS 16-byte pieces; chunk c = it * NTHR + tid is piece
// c % SUBS of pair c / SUBS, so a thread keeps its key m and piece and walks the queries l = l0 + it * kLStep.
template <int NTHR, int SUBS>
struct PairMap {
    static constexpr int kIters = 1024 * SUBS / NTHR, kLStep = NTHR / SUBS / 32;
    static_assert(1024 * SUBS % NTHR == 0 && NTHR % (SUBS * 32) == 0, "pair chunks must tile the threads");
    int sub, m, l0;
    __device__ __forceinline__ explicit PairMap(int tid) : sub(tid % SUBS), m((tid / SUBS) & 31), l0(tid / SUBS / 32) {}
};
// byte offset, inside an eg row, of piece `sub` of the group's [E | G] record (E pieces first)
template <typename T, int HG>
__device__ __forceinline__ uint32_t eg_chan(const tgt_node_attention_args& a, const Unit& u, int sub) {
    constexpr int kE = HG / 8;       // 16-byte pieces of E (8 heads each)
    return (uint32_t)(((sub < kE ? a.e_off + sub * 8 : a.g_off + (sub - kE) * 8) + u.hg * HG) * (int)sizeof(T));
}
template <typename T, int HG, int NTHR = HG * 64>
__device__ __forceinline__ void stage_eg_issue(const tgt_node_attention_args& a, const Unit& u, int tid,
                                               uint4 (&v)[PairMap<NTHR, HG / 4>::kIters]) {
    using PM = PairMap<NTHR, HG / 4>;
    const PM pm(tid);
    const int N = a.N;
    const __amdgpu_buffer_rsrc_t rs = graph_rsrc(a.eg, (int64_t)N * N * a.ld_eg * sizeof(T), u.b);
    const uint32_t ldb = (uint32_t)(a.ld_eg * sizeof(T));
    const uint32_t off0 = (uint32_t)(pm.l0 * N + pm.m) * ldb + eg_chan<T, HG>(a, u, pm.sub);
#pragma unroll
    for (int it = 0; it < PM::kIters; ++it) {
        const bool ok = pm.m < N && pm.l0 + it * PM::kLStep < N;
        v[it] = buf_ld16(rs, ok ? off0 + (uint32_t)(it * PM::kLStep * N) * ldb : kOob);
    }
}
template <int HG, int NTHR = HG * 64>
__device__ __forceinline__ void stage_eg_commit(char* lds, int tid, uint4 (&v)[PairMap<NTHR, HG / 4>::kIters]) {
    using PM = PairMap<NTHR, HG / 4>;
    using L = Lay<HG>;
    const PM pm(tid);
    const int rot = (tid >> 3) & 3;
    char* dst = lds + L::kOffEG + pm.l0 * L::kPitchEG + pm.m * L::kRecEG + pm.sub * 16;
#pragma unroll
    for (int it = 0; it < PM::kIters; ++it) lds_put16(dst + it * PM::kLStep * L::kPitchEG, v[it], rot);
}
template <typename T, int HG, int NTHR = HG * 64>
__device__ __forceinline__ void unstage_eg(const char* lds, void* d_eg, const tgt_node_attention_args& a, const Unit& u, int tid) {
    using PM = PairMap<NTHR, HG / 4>;
    using L = Lay<HG>;
    const PM pm(tid);
    const int N = a.N, rot = (tid >> 3) & 3;
    const __amdgpu_buffer_rsrc_t rs = graph_rsrc(d_eg, (int64_t)N * N * a.ld_eg * sizeof(T), u.b);
    const uint32_t ldb = (uint32_t)(a.ld_eg * sizeof(T));
    const uint32_t off0 = (uint32_t)(pm.l0 * N + pm.m) * ldb + eg_chan<T, HG>(a, u, pm.sub);
    const char* src = lds + L::kOffEG + pm.l0 * L::kPitchEG + pm.m * L::kRecEG + pm.sub * 16;
#pragma unroll
    for (int it = 0; it < PM::kIters; ++it) {
        const bool ok = pm.m < N && pm.l0 + it * PM::kLStep < N;
        buf_st16(rs, ok ? off0 + (uint32_t)(it * PM::kLStep * N) * ldb : kOob, lds_get16(src + it * PM::kLStep * L::kPitchEG, rot));
    }
}
// (B,N,N,H) tensors (H_hat, dH_hat): HG/8 pieces per pair
template <typename T, int HG, int NTHR = HG * 64>
__device__ __forceinline__ void stage_h_issue(const void* x, const tgt_node_attention_args& a, const Unit& u, int tid,
                                              uint4 (&v)[PairMap<NTHR, HG / 8>::kIters]) {
    using PM = PairMap<NTHR, HG / 8>;
    const PM pm(tid);
    const int N = a.N;
    const __amdgpu_buffer_rsrc_t rs = graph_rsrc(x, (int64_t)N * N * a.H * sizeof(T), u.b);
    const uint32_t ldb = (uint32_t)(a.H * sizeof(T));
    const uint32_t off0 = (uint32_t)(pm.l0 * N + pm.m) * ldb + (uint32_t)((u.hg * HG + pm.sub * 8) * (int)sizeof(T));
#pragma unroll
    for (int it = 0; it < PM::kIters; ++it) {
        const bool ok = x && pm.m < N && pm.l0 + it * PM::kLStep < N;
        v[it] = buf_ld16(rs, ok ? off0 + (uint32_t)(it * PM::kLStep * N) * ldb : kOob);
    }
}
template <int HG, int NTHR = HG * 64>
__device__ __forceinline__ void stage_h_commit(char* lds, int tid, uint4 (&v)[PairMap<NTHR, HG / 8>::kIters]) {
    using PM = PairMap<NTHR, HG / 8>;
    using L = Lay<HG>;
    const PM pm(tid);
    const int rot = (tid >> 3) & 3;
    char* dst = lds + L::kOffH + pm.l0 * L::kPitchH + pm.m * L::kRecH + pm.sub * 16;
#pragma unroll
    for (int it = 0; it < PM::kIters; ++it) lds_put16(dst + it * PM::kLStep * L::kPitchH, v[it], rot);
}
// H_hat out of the E slots of the E|G image
template <typename T, int HG, int NTHR = HG * 64>
__device__ __forceinline__ void unstage_hhat(const char* lds, void* hhat, const tgt_node_attention_args& a, const Unit& u, int tid) {
    using PM = PairMap<NTHR, HG / 8>;
    using L = Lay<HG>;
    const PM pm(tid);
    const int N = a.N, rot = (tid >> 3) & 3;
    const __amdgpu_buffer_rsrc_t rs = graph_rsrc(hhat, (int64_t)N * N * a.H * sizeof(T), u.b);
    const uint32_t ldb = (uint32_t)(a.H * sizeof(T));
    const uint32_t off0 = (uint32_t)(pm.l0 * N + pm.m) * ldb + (uint32_t)((u.hg * HG + pm.sub * 8) * (int)sizeof(T));
    const char* src = lds + L::kOffEG + pm.l0 * L::kPitchEG + pm.m * L::kRecEG + pm.sub * 16;
#pragma unroll
    for (int it = 0; it < PM::kIters; ++it) {
        const bool ok = pm.m < N && pm.l0 + it * PM::kLStep < N;
        buf_st16(rs, ok ? off0 + (uint32_t)(it * PM::kLStep * N) * ldb : kOob, lds_get16(src + it * PM::kLStep * L::kPitchEG, rot));
    }
}
// rows of one (B,N,ld) node tensor: the HG-head segment of every (row, d); chunk c = it * NTHR + tid <-> (row, d, piece)
template <int HG, int NTHR, int D>
struct NodeMap {
    static constexpr int kPer = HG / 8, kChunks = 32 * D * kPer, kIters = (kChunks + NTHR - 1) / NTHR;
    int piece, d, row;
    bool in;
    __device__ __forceinline__ NodeMap(int tid, int it) {
        const int c = it * NTHR + tid;
        piece = c % kPer; d = (c / kPer) % D; row = c / kPer / D; in = c < kChunks;
    }
    __device__ __forceinline__ int lds_off() const { return row * Lay<HG>::pitch_n(D) + d * Lay<HG>::kRecH + piece * 16; }
    __device__ __forceinline__ uint32_t glb_off(int64_t ld, int off, const tgt_node_attention_args& a, const Unit& u, int esz) const {
        return (in && row < a.N) ? (uint32_t)((row * ld + off + d * a.H + u.hg * HG + piece * 8) * (int64_t)esz) : kOob;
    }
};
template <typename T, int HG, int D, int NTHR = HG * 64>
__device__ __forceinline__ void stage_node_issue(const void* x, int64_t ld, int off, const tgt_node_attention_args& a, const Unit& u,
                                                 int tid, uint4 (&v)[NodeMap<HG, NTHR, D>::kIters]) {
    const __amdgpu_buffer_rsrc_t rs = graph_rsrc(x, (int64_t)a.N * ld * sizeof(T), u.b);
#pragma unroll
    for (int it = 0; it < NodeMap<HG, NTHR, D>::kIters; ++it) v[it] = buf_ld16(rs, NodeMap<HG, NTHR, D>(tid, it).glb_off(ld, off, a, u, sizeof(T)));
}
template <int HG, int D, int NTHR = HG * 64>
__device__ __forceinline__ void stage_node_commit(char* region, int tid, uint4 (&v)[NodeMap<HG, NTHR, D>::kIters]) {
#pragma unroll
    for (int it = 0; it < NodeMap<HG, NTHR, D>::kIters; ++it) {
        const NodeMap<HG, NTHR, D> nm(tid, it);
        if (nm.in) lds_put16(region + nm.lds_off(), v[it], (tid >> 3) & 3);
    }
}
template <typename T, int HG, int D, int NTHR = HG * 64>
__device__ __forceinline__ void unstage_node(const char* region, void* x, int64_t ld, int off, const tgt_node_attention_args& a,
                                             const Unit& u, int tid) {
    const __amdgpu_buffer_rsrc_t rs = graph_rsrc(x, (int64_t)a.N * ld * sizeof(T), u.b);
#pragma unroll
    for (int it = 0; it < NodeMap<HG, NTHR, D>::kIters; ++it) {
        const NodeMap<HG, NTHR, D> nm(tid, it);
        if (nm.in) buf_st16(rs, nm.glb_off(ld, off, a, u, sizeof(T)), lds_get16(region + nm.lds_off(), (tid >> 3) & 3));
    }
}
// mask tile: pairs past N get -inf: their logits become -inf (weight exactly 0) and their gates sigmoid(-inf) = 0
// without a select per tile element
template <int NTHR>
__device__ __forceinline__ void stage_mask_issue(const tgt_node_attention_args& a, const Unit& u, int tid, float (&mk)[1024 / NTHR]) {
    const int N = a.N;
#pragma unroll
    for (int it = 0; it < 1024 / NTHR; ++it) {
        const int idx = it * NTHR + tid, l = idx >> 5, m = idx & 31;
        mk[it] = -INFINITY;
        if (l < N && m < N) mk[it] = a.mask[((int64_t)u.b * N + l) * N + m];
    }
}
template <int HG, int NTHR>
__device__ __forceinline__ void stage_mask_commit(char* lds, int tid, const float (&mk)[1024 / NTHR]) {
#pragma unroll
    for (int it = 0; it < 1024 / NTHR; ++it) {
        const int idx = it * NTHR + tid, l = idx >> 5, m = idx & 31;
        *reinterpret_cast<float*>(lds + Lay<HG>::kOffM + l * Lay<HG>::kPitchM + m * 4) = mk[it];
    }
}

// ---- per-wave pieces ---------------------------------------------------------------------------
// operand fragment of head hh: lane (r, hi) holds X[row r][d = 8 hi + t], t = 0..7 (0 past D)
template <typename T, int HG, int D>
__device__ __forceinline__ frag_t<T> node_frag(const char* region, int r, int hi, int hh) {
    frag_t<T> f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int d = 8 * hi + t;
        T v = from_f32<T>(0.f);
        if (d < D) v = *reinterpret_cast<const T*>(region + r * Lay<HG>::pitch_n(D) + d * Lay<HG>::kRecH + hh * 2);
        f[t] = v;
    }
    return f;
}
// transposed result X^T[d][row] (lane = row r, register q <-> d = acc_row(q, hi)) into column hh
template <typename T, int HG, int D>
__device__ __forceinline__ void node_put(char* region, const f32x16& acc, int r, int hi, int hh) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int d = acc_row(q, hi);
        if (d < D) *reinterpret_cast<T*>(region + r * Lay<HG>::pitch_n(D) + d * Lay<HG>::kRecH + hh * 2) = from_f32<T>(acc[q]);
    }
}

// ---------------------------------------------------------------------------
// forward tile of head hh (one wave): H_hat into the E slots, V_att into the head's Q column
// ---------------------------------------------------------------------------
template <typename T, int HG, int D>
__device__ __forceinline__ void tile_fwd(char* lds, const tgt_node_attention_args& a, const Unit& u, int r, int hi, int hh) {
    using F = frag_t<T>;
    using L = Lay<HG>;
    const int N = a.N, H = a.H, h = u.hg * HG + hh;
    char* rQ = lds + L::off_n(D, false, 0);
    char* rK = lds + L::off_n(D, false, 1);
    char* rV = lds + L::off_n(D, false, 2);
    {
        const F fq = node_frag<T, HG, D>(rQ, r, hi, hh), fk = node_frag<T, HG, D>(rK, r, hi, hh), fv = node_frag<T, HG, D>(rV, r, hi, hh);
        F ident_d[1];
        make_ident_d<T, 1>(ident_d, r, hi);
        const f32x16 z = {0};
        f32x16 s = mma32(fk, fq, z);                      // S^T[key][query]: lane = query l, register <-> key
        const f32x16 vt = mma32(fv, ident_d[0], z);       // V[key][d] -> lane d
        float gt[16], mx = -INFINITY;
        char* pe0 = lds + L::kOffEG + r * L::kPitchEG + hh * 2;
        const char* pm0 = lds + L::kOffM + r * L::kPitchM;
        const float hs = a.hhat_scale ? a.hhat_scale[u.b] : 1.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int m = acc_row(q, hi);
            char* pe = pe0 + m * L::kRecEG;
            const float e = to_f32(*reinterpret_cast<const T*>(pe)), g = to_f32(*reinterpret_cast<const T*>(pe + L::kRecH));
            const float mk = *reinterpret_cast<const float*>(pm0 + m * 4);
            const float sv = s[q] * a.scale + e;
            *reinterpret_cast<T*>(pe) = from_f32<T>(sv * hs);     // H_hat (times the branch's DropPath factor, if given) leaves through the E slot this lane just read
            const float x = sv + mk;                              // (mk = -inf past N)
            gt[q] = fast_sigmoid(g + mk);
            s[q] = x;
            mx = fmaxf(mx, x);
        }
        mx = fmaxf(mx, xhalf(mx));
        const float mref = mx == -INFINITY ? 0.f : mx;
        float sum = 0.f, gsum = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            s[q] = fast_exp(s[q] - mref);
            sum += s[q];
            gsum += gt[q];
        }
        sum += xhalf(sum);
        gsum += xhalf(gsum);
        const float inv = sum > 0.f ? fast_rcp(sum) : 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) s[q] = s[q] * inv * gt[q];
        f32x16 o = {0};
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) o = mma32(pack_chunk<T>(vt, cc), pack_chunk<T>(s, cc), o);     // O^T[d][query]
        const float f = a.scale_degree ? __logf(1.f + gsum) : 1.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] *= f;
        node_put<T, HG, D>(rQ, o, r, hi, hh);             // V_att leaves through this head's Q column
        if (r < N && hi == 0) {
            a.lse[((int64_t)u.b * N + r) * H + h] = mx + __logf(sum);
            a.gsum[((int64_t)u.b * N + r) * H + h] = gsum;
        }
    }
}

// ---------------------------------------------------------------------------
// backward tile of head hh (one wave): dE, dG into the E / G slots, dQ, dK, dV into the head's Q, K, V columns.
// With P, the gates and their sum recomputed:
//   dA^T[m][l] = dsc_l * V[m,:].dV_att[l,:]      dsc = log(1 + sum_m g)   (1 without the degree scaler)
//   delta_l = sum_m P dA g        d_dsc = delta / dsc        dgsum = d_dsc / (1 + sum g)
//   dS = P (dA g - delta)         dG = (dA P + dgsum) g (1 - g)            dE = dH_hat + dS
//   dQ^T = s K^T dE^T             dK^T = s Q^T dE                          dV^T = dV_att^T (P g dsc)
// ---------------------------------------------------------------------------
template <typename T, int HG, int D>
__device__ __forceinline__ void tile_bwd(char* lds, const tgt_node_attention_args& a, const Unit& u, int r, int hi, int hh) {
    using F = frag_t<T>;
    using L = Lay<HG>;
    char* rQ = lds + L::off_n(D, true, 0);
    char* rK = lds + L::off_n(D, true, 1);
    char* rV = lds + L::off_n(D, true, 2);
    char* rO = lds + L::off_n(D, true, 3);
    {
        const F fq = node_frag<T, HG, D>(rQ, r, hi, hh), fk = node_frag<T, HG, D>(rK, r, hi, hh);
        const F fv = node_frag<T, HG, D>(rV, r, hi, hh), fo = node_frag<T, HG, D>(rO, r, hi, hh);
        F ident_d[1];
        make_ident_d<T, 1>(ident_d, r, hi);
        const f32x16 z = {0};
        f32x16 s = mma32(fk, fq, z);                      // S^T[key][query]
        float gt[16], mx = -INFINITY;
        char* pe0 = lds + L::kOffEG + r * L::kPitchEG + hh * 2;
        const char* pm0 = lds + L::kOffM + r * L::kPitchM;
        const char* ph0 = lds + L::kOffH + r * L::kPitchH + hh * 2;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int m = acc_row(q, hi);
            const char* pe = pe0 + m * L::kRecEG;
            const float e = to_f32(*reinterpret_cast<const T*>(pe)), g = to_f32(*reinterpret_cast<const T*>(pe + L::kRecH));
            const float mk = *reinterpret_cast<const float*>(pm0 + m * 4);
            const float x = s[q] * a.scale + e + mk;              // (mk = -inf past N)
            gt[q] = fast_sigmoid(g + mk);
            s[q] = x;
            mx = fmaxf(mx, x);
            if ((q & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // (keeps the scheduler from hoisting all 48 LDS reads: 128 registers)
        }
        mx = fmaxf(mx, xhalf(mx));
        if (mx == -INFINITY) mx = 0.f;                    // padding query: every weight exactly 0
        float sum = 0.f, gsum = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            s[q] = fast_exp(s[q] - mx);
            sum += s[q];
            gsum += gt[q];
        }
        sum += xhalf(sum);
        gsum += xhalf(gsum);
        const float inv = sum > 0.f ? fast_rcp(sum) : 0.f;
        const float dsc = a.scale_degree ? __logf(1.f + gsum) : 1.f;
        f32x16 da = mma32(fv, fo, z);                     // (V dV_att^T)[key][query]
        float delta = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            s[q] *= inv;                                  // P
            da[q] *= dsc;                                 // dA (gradient wrt the unscaled V_att folded in)
            delta += s[q] * da[q] * gt[q];
        }
        delta += xhalf(delta);
        const float d_dsc = dsc != 0.f ? delta * fast_rcp(dsc) : 0.f;      // zero scaler <=> every gate 0 <=> V_att 0
        const float dgsum = a.scale_degree ? d_dsc * fast_rcp(1.f + gsum) : 0.f;

        F dsf[2], af[2];
        float tile_unscale = 1.f;
        const float hs = a.hhat_scale ? a.hhat_scale[u.b] : 1.f;       // d_hhat is the gradient of hhat_scale * H_hat
        {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int m = acc_row(q, hi);
                const float dh = to_f32(*reinterpret_cast<const T*>(ph0 + m * L::kRecH)) * hs;
                const float p = s[q], g = gt[q];
                const float dS = p * (da[q] * g - delta);
                const float dGl = (da[q] * p + dgsum) * g * (1.f - g);
                const float dH = dh + dS;
                char* pe = pe0 + m * L::kRecEG;
                *reinterpret_cast<T*>(pe) = from_f32<T>(dH);           // dE, dG leave through the E / G slots of this lane
                *reinterpret_cast<T*>(pe + L::kRecH) = from_f32<T>(dGl);
                af[q >> 3][q & 7] = from_f32<T>(p * g * dsc);          // operand fragments are packed as they appear (registers)
                if constexpr (kIsBf16<T>) dsf[q >> 3][q & 7] = from_f32<T>(dH * a.scale);
                else s[q] = dH * a.scale;
                if ((q & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (!kIsBf16<T>) {
                // fp16 operands: bring the tile's largest |dE| to 2^13 (an exact power-of-two factor, undone on dQ / dK) so
                // that small gradients do not sink into fp16 subnormals on their way through the matrix core
                float amax = 0.f;
#pragma unroll
                for (int q = 0; q < 16; ++q) amax = fmaxf(amax, fabsf(s[q]));
                amax = group_max<64>(amax);
                const int ex = (int)((__builtin_bit_cast(uint32_t, amax) >> 23) & 0xffu);
                float c = 1.f;
                if (ex >= 14 && ex <= 253) {
                    c = __builtin_bit_cast(float, (uint32_t)(267 - ex) << 23);                     // 2^(13 - (ex - 127))
                    tile_unscale = __builtin_bit_cast(float, (uint32_t)(ex - 13) << 23);           // 1 / c
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) dsf[q >> 3][q & 7] = from_f32<T>(s[q] * c);
            }
        }
        // (one product at a time from here on: the 128-register budget of 4 waves per SIMD)
        {   // dQ^T[d][query] = s sum_key K^T[d][key] dE^T[key][query]   (K^T through the identity)
            const f32x16 kT = mma32(fk, ident_d[0], z);
            f32x16 dq = {0};
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) dq = mma32(pack_chunk<T>(kT, cc), dsf[cc], dq);
            if constexpr (!kIsBf16<T>) {
#pragma unroll
                for (int q = 0; q < 8; ++q) dq[q] *= tile_unscale;
            }
            node_put<T, HG, D>(rQ, dq, r, hi, hh);
        }
        F ident_k[2];
        make_ident_k<T>(ident_k, r, hi);
        {   // dK^T[d][key] = s sum_query Q^T[d][query] dE[query][key]   (dE re-laid out to lane = key)
            f32x16 ds2 = {0};
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) ds2 = mma32(dsf[cc], ident_k[cc], ds2);
            const f32x16 qT = mma32(fq, ident_d[0], z);
            f32x16 dk = {0};
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) dk = mma32(pack_chunk<T>(qT, cc), pack_chunk<T>(ds2, cc), dk);
            if constexpr (!kIsBf16<T>) {
#pragma unroll
                for (int q = 0; q < 8; ++q) dk[q] *= tile_unscale;
            }
            node_put<T, HG, D>(rK, dk, r, hi, hh);
        }
        {   // dV^T[d][key] = sum_query dV_att^T[d][query] (P g dsc)[query][key]
            f32x16 a2 = {0};
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) a2 = mma32(af[cc], ident_k[cc], a2);
            const f32x16 oT = mma32(fo, ident_d[0], z);
            f32x16 dv = {0};
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) dv = mma32(pack_chunk<T>(oT, cc), pack_chunk<T>(a2, cc), dv);
            node_put<T, HG, D>(rV, dv, r, hi, hh);
        }
    }
}

// ---------------------------------------------------------------------------
// The kernels: one workgroup per unit (graph, head group), NW = HG waves, wave = head.
// Unit order: XCD x (= block index & 7) owns the graphs b = x mod 8; its workgroups take that list's units
// (graph-major, head group minor) in order, so the head groups of one graph run at the same time on the same XCD
// and the 128-byte E / G / dH rows they share come from HBM once.
// What bounds them (B=256, N=32, H=64, D=12, bf16, rocprofv3; TGT_NODE_ABLATE): load, tile math and store of a
// unit cannot overlap (one LDS image) and workgroups that start together stay in step, so the phases add up --
// backward: loads alone 30 us + tile math 43 us + stores 24 us, the kernel 86-103 us.  Built against that,
// measured and removed again (git history): 8-head workgroups, two per CU (they stay in step too, and a load
// request per 16 bytes is too fine for the L2: 117 us; the form remains for D = 16, whose 16-head backward image
// does not fit the LDS); a delayed start for every other first-round workgroup (slower by the delay); a persistent
// 8-wave workgroup prefetching the NEXT unit's loads into registers, then into accumulation registers by inline
// assembly, under the tile math (forward 48.7-51.0 us against 50.0-50.5; the backward's tile math spills under the
// 128 + 128 register split hipcc applies once accumulation registers are in use: 136-153 us).
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool unit_of_block(const tgt_node_attention_args& a, int HG, Unit& u) {
    const int groups = a.H / HG, x = blockIdx.x & 7, t = blockIdx.x >> 3;
    const int nt = ((a.B - x + 7) >> 3) * groups;       // units of the graphs b = x, x + 8, ... < B
    u = Unit{(t / groups) * 8 + x, t % groups};
    return t < nt;
}

template <typename T, int HG, int D>
__global__ void __launch_bounds__(HG * 64, 4) node_att_mfma_fwd_kernel(const tgt_node_attention_args a, const int ablate) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    using L = Lay<HG>;
    constexpr int NTHR = HG * 64;
    const int tid = threadIdx.x, lane = tid & 63, hh = tid >> 6, r = lane & 31, hi = lane >> 5;
    char* rQ = lds + L::off_n(D, false, 0);
    char* rK = lds + L::off_n(D, false, 1);
    char* rV = lds + L::off_n(D, false, 2);
    Unit u;
    if (!unit_of_block(a, HG, u)) return;

    if (!(ablate & 2)) {   // stage: every load is issued before the first LDS write
        uint4 veg[PairMap<NTHR, HG / 4>::kIters];
        uint4 vq[NodeMap<HG, NTHR, D>::kIters], vk[NodeMap<HG, NTHR, D>::kIters], vv[NodeMap<HG, NTHR, D>::kIters];
        float mk[1024 / NTHR];
        stage_eg_issue<T, HG>(a, u, tid, veg);
        stage_node_issue<T, HG, D>(a.qkv, a.ld_qkv, a.q_off, a, u, tid, vq);
        stage_node_issue<T, HG, D>(a.qkv, a.ld_qkv, a.k_off, a, u, tid, vk);
        stage_node_issue<T, HG, D>(a.qkv, a.ld_qkv, a.v_off, a, u, tid, vv);
        stage_mask_issue<NTHR>(a, u, tid, mk);
        stage_eg_commit<HG>(lds, tid, veg);
        stage_node_commit<HG, D>(rQ, tid, vq);
        stage_node_commit<HG, D>(rK, tid, vk);
        stage_node_commit<HG, D>(rV, tid, vv);
        stage_mask_commit<HG, NTHR>(lds, tid, mk);
    }
    __syncthreads();
    if (!(ablate & 1)) tile_fwd<T, HG, D>(lds, a, u, r, hi, hh);
    __syncthreads();
    if (ablate & 4) return;
    if (a.hhat) unstage_hhat<T, HG>(lds, a.hhat, a, u, tid);
    unstage_node<T, HG, D>(rQ, a.vatt, (int64_t)D * a.H, 0, a, u, tid);
}

template <typename T, int HG, int D>
__global__ void __launch_bounds__(HG * 64, 4) node_att_mfma_bwd_kernel(const tgt_node_attention_args a, const int ablate) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    using L = Lay<HG>;
    constexpr int NTHR = HG * 64;
    const int tid = threadIdx.x, lane = tid & 63, hh = tid >> 6, r = lane & 31, hi = lane >> 5;
    char* rQ = lds + L::off_n(D, true, 0);
    char* rK = lds + L::off_n(D, true, 1);
    char* rV = lds + L::off_n(D, true, 2);
    char* rO = lds + L::off_n(D, true, 3);
    Unit u;
    if (!unit_of_block(a, HG, u)) return;

    if (!(ablate & 2)) {
        uint4 veg[PairMap<NTHR, HG / 4>::kIters], vh[PairMap<NTHR, HG / 8>::kIters];
        uint4 vq[NodeMap<HG, NTHR, D>::kIters], vk[NodeMap<HG, NTHR, D>::kIters], vv[NodeMap<HG, NTHR, D>::kIters],
            vo[NodeMap<HG, NTHR, D>::kIters];
        float mk[1024 / NTHR];
        stage_eg_issue<T, HG>(a, u, tid, veg);
        stage_h_issue<T, HG>(a.d_hhat, a, u, tid, vh);
        stage_node_issue<T, HG, D>(a.qkv, a.ld_qkv, a.q_off, a, u, tid, vq);
        stage_node_issue<T, HG, D>(a.qkv, a.ld_qkv, a.k_off, a, u, tid, vk);
        stage_node_issue<T, HG, D>(a.qkv, a.ld_qkv, a.v_off, a, u, tid, vv);
        stage_node_issue<T, HG, D>(a.d_vatt, (int64_t)D * a.H, 0, a, u, tid, vo);
        stage_mask_issue<NTHR>(a, u, tid, mk);
        stage_eg_commit<HG>(lds, tid, veg);
        stage_h_commit<HG>(lds, tid, vh);
        stage_node_commit<HG, D>(rQ, tid, vq);
        stage_node_commit<HG, D>(rK, tid, vk);
        stage_node_commit<HG, D>(rV, tid, vv);
        stage_node_commit<HG, D>(rO, tid, vo);
        stage_mask_commit<HG, NTHR>(lds, tid, mk);
    }
    __syncthreads();
    if (!(ablate & 1)) tile_bwd<T, HG, D>(lds, a, u, r, hi, hh);
    __syncthreads();
    if (ablate & 4) return;
    unstage_eg<T, HG>(lds, a.d_eg, a, u, tid);
    unstage_node<T, HG, D>(rQ, a.d_qkv, a.ld_qkv, a.q_off, a, u, tid);
    unstage_node<T, HG, D>(rK, a.d_qkv, a.ld_qkv, a.k_off, a, u, tid);
    unstage_node<T, HG, D>(rV, a.d_qkv, a.ld_qkv, a.v_off, a, u, tid);
}

constexpr int kLdsMax = 160 * 1024;

template <typename T, int HG, int D>
static int launch(const tgt_node_attention_args& a, bool bwd, hipStream_t st) {
    using L = Lay<HG>;
    const int grid = ((a.B + 7) / 8) * 8 * (a.H / HG);       // a multiple of 8: the XCD lists of unit_of_block cover every unit
#ifdef TGT_PROBES
    static const int ablate = getenv("TGT_NODE_ABLATE") ? atoi(getenv("TGT_NODE_ABLATE")) : 0;    // probe builds only: 1 no tile math, 2 no loads, 4 no stores
#else
    constexpr int ablate = 0;                 // (the shipped library has no ablation switch: a stray environment variable cannot corrupt results)
#endif
    if (!bwd) {
        constexpr int kLds = L::lds_bytes(D, false);
        static_assert(kLds <= kLdsMax, "forward LDS");
        static bool attr_set[16] = {};                 // per device (common.hpp: dyn_lds_once)
        if (!dyn_lds_once(attr_set, reinterpret_cast<const void*>(&node_att_mfma_fwd_kernel<T, HG, D>), kLds))
            return set_error(TGT_ERR_LAUNCH, "node_att_mfma_fwd_kernel: cannot reserve %d bytes of LDS", kLds);
        hipLaunchKernelGGL((node_att_mfma_fwd_kernel<T, HG, D>), dim3(grid), dim3(HG * 64), kLds, st, a, ablate);
        return check_launch("node_att_mfma_fwd_kernel");
    } else {
        constexpr int kLds = L::lds_bytes(D, true);
        if constexpr (kLds <= kLdsMax) {
            static bool attr_set[16] = {};                 // per device (common.hpp: dyn_lds_once)
            if (!dyn_lds_once(attr_set, reinterpret_cast<const void*>(&node_att_mfma_bwd_kernel<T, HG, D>), kLds))
                return set_error(TGT_ERR_LAUNCH, "node_att_mfma_bwd_kernel: cannot reserve %d bytes of LDS", kLds);
            hipLaunchKernelGGL((node_att_mfma_bwd_kernel<T, HG, D>), dim3(grid), dim3(HG * 64), kLds, st, a, ablate);
            return check_launch("node_att_mfma_bwd_kernel");
        }
        return -1;
    }
}

template <typename T, int HG>
static int dispatch_d(const tgt_node_attention_args& a, bool bwd, hipStream_t st) {
    switch (a.D) {
        case 8: return launch<T, HG, 8>(a, bwd, st);
        case 12: return launch<T, HG, 12>(a, bwd, st);
        case 16: return launch<T, HG, 16>(a, bwd, st);
        default: return -1;
    }
}

// heads per workgroup: 16 (32-byte row segments) unless the shape says 8 (16-byte segments; the
// 16-head backward image of D = 16 does not fit the LDS, that shape takes the 8-head form)
static int heads_per_group(const tgt_node_attention_args& a, bool bwd) {
    if (a.H % 16 == 0 && !(bwd && a.D == 16)) return 16;
    return 8;
}

template <typename T>
static int dispatch(const tgt_node_attention_args& a, bool bwd, hipStream_t st) {
    return heads_per_group(a, bwd) == 16 ? dispatch_d<T, 16>(a, bwd, st) : dispatch_d<T, 8>(a, bwd, st);
}

}  // namespace nmf

// Is this call one of the shapes the matrix-core kernels take?  (the rest stays on node_attention.hip)
bool node_attention_mfma_eligible(const tgt_node_attention_args& a, bool bwd) {
    static const int on = getenv("TGT_NODE_MFMA") ? atoi(getenv("TGT_NODE_MFMA")) : 1;
    if (!on || a.logits_only || a.dtype == TGT_F32) return false;
    if (a.N < 1 || a.N > 32 || a.H % 8 || !(a.D == 8 || a.D == 12 || a.D == 16)) return false;
    if (!a.mask || !a.vatt || !a.lse || !a.gsum) return false;
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    if (a.ld_qkv % 8 || a.q_off % 8 || a.k_off % 8 || a.v_off % 8 || a.ld_eg % 8 || a.e_off % 8 || a.g_off % 8) return false;
    if (!al16(a.qkv) || !al16(a.eg) || !al16(a.vatt) || (a.hhat && !al16(a.hhat))) return false;
    if (bwd && (!al16(a.d_qkv) || !al16(a.d_eg) || !al16(a.d_vatt) || (a.d_hhat && !al16(a.d_hhat)))) return false;
    return true;
}

// returns TGT_OK / an error; call only when node_attention_mfma_eligible()
int node_attention_mfma_run(const tgt_node_attention_args& a, bool bwd, hipStream_t st) {
    int e = a.dtype == TGT_BF16 ? nmf::dispatch<bf16_t>(a, bwd, st) : nmf::dispatch<f16_t>(a, bwd, st);
    if (e < 0) return set_error(TGT_ERR_UNSUPPORTED, "node attention (matrix core): unsupported D=%d", a.D);
    return e;
}

}  // namespace tgt
