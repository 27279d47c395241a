// torch dispatcher registration of the hot-path ops: a shared object for
// torch.ops.load_library() that registers  tgt::egt_attention[_fwd/_bwd],  tgt::triplet_attention[_fwd/_bwd]  and
// tgt::triplet_aggregate[_fwd/_bwd]  on top of the C ABI of libtgt_hip.so (include/tgt_hip.h).
//
// The reference has no such seam (its ops are einsum chains inside lib/tgt/layers/layers.py:62-77 and
// lib/tgt/layers/triplet.py:45-73, 205-250); this is the form a maintainer would call from there.  Conventions:
//   * tensors are borrowed for the call; outputs come from the torch caching allocator (at::empty), nothing is retained;
//   * the kernels run on the CURRENT stream of the tensors' device (forward: the caller's thread; backward: the
//     autograd engine's device thread);
//   * shape / dtype / device / alignment violations raise through TORCH_CHECK; there is no eager fallback.
// Host-only translation unit: no device code here, the kernels live in libtgt_hip.so.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/autograd.h>
#include <torch/library.h>

#include <tuple>
#include <vector>

#include "../../include/tgt_hip.h"

namespace {

using at::Tensor;

int dtype_code(const Tensor& t) {
    switch (t.scalar_type()) {
        case at::kFloat: return TGT_F32;
        case at::kBFloat16: return TGT_BF16;
        case at::kHalf: return TGT_F16;
        default: TORCH_CHECK(false, "tgt ops: dtype ", t.scalar_type(), " not in {float32, bfloat16, float16}");
    }
}
void ok(int code, const char* what) { TORCH_CHECK(code == 0, what, " failed (code ", code, "): ", tgt_last_error()); }
void* stream_of(const Tensor& t) { return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.get_device()).stream(); }
void on_gpu(const Tensor& t, const char* name) { TORCH_CHECK(t.is_cuda(), "tgt ops: ", name, " must be a GPU tensor (there is no CPU path)"); }
Tensor mask3(const Tensor& mask, int64_t B, int64_t N) {
    on_gpu(mask, "mask");
    TORCH_CHECK(mask.numel() == B * N * N, "tgt ops: mask must hold B*N*N elements");
    return mask.reshape({B, N, N}).to(at::kFloat).contiguous();
}

// ---------------------------------------------------------------------------
// node attention with edge bias and gate: qkv (B,N,3W) head-minor [Q|K|V], eg (B,N,N,2H) [E|G], mask (B,N,N[,1])
// ---------------------------------------------------------------------------
tgt_node_attention_args node_args(const Tensor& qkv, const Tensor& eg, const Tensor& m3, int64_t H, bool scale_degree) {
    TORCH_CHECK(qkv.dim() == 3 && eg.dim() == 4 && qkv.size(2) % (3 * H) == 0 && eg.size(3) == 2 * H && eg.size(1) == qkv.size(1),
                "tgt::egt_attention: qkv (B,N,3W), eg (B,N,N,2H) expected");
    tgt_node_attention_args a{};
    const int64_t W = qkv.size(2) / 3;
    a.B = (int32_t)qkv.size(0); a.N = (int32_t)qkv.size(1); a.H = (int32_t)H; a.D = (int32_t)(W / H);
    a.dtype = dtype_code(qkv); a.scale_degree = scale_degree; a.logits_only = 0;
    a.scale = 1.f / std::sqrt((float)a.D);
    a.qkv = qkv.data_ptr(); a.ld_qkv = qkv.size(2); a.q_off = 0; a.k_off = (int32_t)W; a.v_off = (int32_t)(2 * W);
    a.eg = eg.data_ptr(); a.ld_eg = eg.size(3); a.e_off = 0; a.g_off = (int32_t)H;
    a.mask = m3.data_ptr<float>();
    return a;
}

std::tuple<Tensor, Tensor, Tensor, Tensor> egt_attention_fwd(const Tensor& qkv_, const Tensor& eg_, const Tensor& mask, int64_t H,
                                                             bool scale_degree, bool want_edges) {
    on_gpu(qkv_, "qkv"); on_gpu(eg_, "eg");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(qkv_.device());
    const Tensor qkv = qkv_.contiguous(), eg = eg_.to(qkv_.scalar_type()).contiguous();
    const int64_t B = qkv.size(0), N = qkv.size(1), W = qkv.size(2) / 3;
    const Tensor m3 = mask3(mask, B, N);
    tgt_node_attention_args a = node_args(qkv, eg, m3, H, scale_degree);
    Tensor vatt = at::empty({B, N, W}, qkv.options());
    Tensor hhat = want_edges ? at::empty({B, N, N, H}, qkv.options()) : at::empty({0}, qkv.options());
    Tensor lse = at::empty({B, N, H}, qkv.options().dtype(at::kFloat)), gsum = at::empty({B, N, H}, qkv.options().dtype(at::kFloat));
    a.vatt = vatt.data_ptr(); a.hhat = want_edges ? hhat.data_ptr() : nullptr;
    a.lse = lse.data_ptr<float>(); a.gsum = gsum.data_ptr<float>();
    ok(tgt_node_attention_fwd(&a, stream_of(qkv)), "tgt_node_attention_fwd");
    return {vatt, hhat, lse, gsum};
}

std::tuple<Tensor, Tensor> egt_attention_bwd(const Tensor& qkv_, const Tensor& eg_, const Tensor& mask, const Tensor& vatt,
                                             const Tensor& lse, const Tensor& gsum, const Tensor& d_vatt_,
                                             const c10::optional<Tensor>& d_hhat_, int64_t H, bool scale_degree) {
    on_gpu(qkv_, "qkv"); on_gpu(eg_, "eg"); on_gpu(d_vatt_, "d_vatt");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(qkv_.device());
    const Tensor qkv = qkv_.contiguous(), eg = eg_.to(qkv_.scalar_type()).contiguous();
    const int64_t B = qkv.size(0), N = qkv.size(1);
    const Tensor m3 = mask3(mask, B, N);
    tgt_node_attention_args a = node_args(qkv, eg, m3, H, scale_degree);
    const Tensor d_vatt = d_vatt_.to(qkv.scalar_type()).contiguous();
    Tensor d_hhat;
    if (d_hhat_.has_value() && d_hhat_->defined() && d_hhat_->numel()) d_hhat = d_hhat_->to(qkv.scalar_type()).contiguous();
    Tensor d_qkv = at::empty_like(qkv), d_eg = at::empty_like(eg);
    const Tensor va = vatt.contiguous(), ls = lse.contiguous(), gs = gsum.contiguous();
    a.vatt = va.data_ptr(); a.lse = ls.data_ptr<float>(); a.gsum = gs.data_ptr<float>();
    a.d_vatt = d_vatt.data_ptr(); a.d_hhat = d_hhat.defined() ? d_hhat.data_ptr() : nullptr;
    a.d_qkv = d_qkv.data_ptr(); a.d_eg = d_eg.data_ptr();
    ok(tgt_node_attention_bwd(&a, stream_of(qkv)), "tgt_node_attention_bwd");
    return {d_qkv, d_eg};
}

// ---------------------------------------------------------------------------
// triplet attention: per direction qkv (B,N,N,3C) HEAD-MAJOR [Q|K|V] (channel = h*D + d), eg (B,N,N,2Ht) [E|G];
// out (B,N,N,2C) = [O_in | O_out], head-major
// ---------------------------------------------------------------------------
tgt_triplet_attention_args tri_args(const Tensor& qi, const Tensor& ei, const Tensor& qo, const Tensor& eo, const Tensor& m3, int64_t H) {
    TORCH_CHECK(qi.dim() == 4 && qi.sizes() == qo.sizes() && ei.sizes() == eo.sizes() && qi.size(3) % (3 * H) == 0 && ei.size(3) == 2 * H,
                "tgt::triplet_attention: qkv_in/out (B,N,N,3C), eg_in/out (B,N,N,2H) expected");
    tgt_triplet_attention_args a{};
    const int64_t C = qi.size(3) / 3;
    a.B = (int32_t)qi.size(0); a.N = (int32_t)qi.size(1); a.H = (int32_t)H; a.D = (int32_t)(C / H);
    a.dtype = dtype_code(qi); a.flags = TGT_TRI_BIASED | TGT_TRI_GATED; a.scale = 1.f / std::sqrt((float)a.D);
    const Tensor* q[2] = {&qi, &qo};
    const Tensor* e[2] = {&ei, &eo};
    for (int d = 0; d < 2; ++d) {
        a.qkv[d] = q[d]->data_ptr(); a.ld_qkv[d] = 3 * C; a.q_off[d] = 0; a.k_off[d] = (int32_t)C; a.v_off[d] = (int32_t)(2 * C);
        a.eg[d] = e[d]->data_ptr(); a.ld_eg[d] = 2 * H; a.e_off[d] = 0; a.g_off[d] = (int32_t)H;
        a.o_off[d] = (int32_t)(d * C);
    }
    a.mask = m3.data_ptr<float>();
    a.ld_out = 2 * C;
    return a;
}

Tensor triplet_attention_fwd(const Tensor& qkv_in, const Tensor& eg_in, const Tensor& qkv_out, const Tensor& eg_out, const Tensor& mask,
                             int64_t H) {
    on_gpu(qkv_in, "qkv_in"); on_gpu(qkv_out, "qkv_out"); on_gpu(eg_in, "eg_in"); on_gpu(eg_out, "eg_out");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(qkv_in.device());
    const auto dt = qkv_in.scalar_type();
    const Tensor qi = qkv_in.contiguous(), qo = qkv_out.to(dt).contiguous(), ei = eg_in.to(dt).contiguous(), eo = eg_out.to(dt).contiguous();
    const int64_t B = qi.size(0), N = qi.size(1), C = qi.size(3) / 3;
    const Tensor m3 = mask3(mask, B, N);
    tgt_triplet_attention_args a = tri_args(qi, ei, qo, eo, m3, H);
    Tensor out = at::empty({B, N, N, 2 * C}, qi.options());
    a.out = out.data_ptr();
    ok(tgt_triplet_attention_fwd(&a, stream_of(qi)), "tgt_triplet_attention_fwd");
    return out;
}

std::tuple<Tensor, Tensor, Tensor, Tensor> triplet_attention_bwd(const Tensor& qkv_in, const Tensor& eg_in, const Tensor& qkv_out,
                                                                 const Tensor& eg_out, const Tensor& mask, const Tensor& d_out_, int64_t H) {
    on_gpu(qkv_in, "qkv_in"); on_gpu(d_out_, "d_out");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(qkv_in.device());
    const auto dt = qkv_in.scalar_type();
    const Tensor qi = qkv_in.contiguous(), qo = qkv_out.to(dt).contiguous(), ei = eg_in.to(dt).contiguous(), eo = eg_out.to(dt).contiguous();
    const int64_t B = qi.size(0), N = qi.size(1);
    const Tensor m3 = mask3(mask, B, N);
    tgt_triplet_attention_args a = tri_args(qi, ei, qo, eo, m3, H);
    const Tensor d_out = d_out_.to(dt).contiguous();
    Tensor dqi = at::empty_like(qi), dqo = at::empty_like(qo), dei = at::empty_like(ei), deo = at::empty_like(eo);
    a.d_out = d_out.data_ptr();
    a.out = d_out.data_ptr();          // (checked non-null; the backward does not touch it)
    a.d_qkv[0] = dqi.data_ptr(); a.d_qkv[1] = dqo.data_ptr(); a.d_eg[0] = dei.data_ptr(); a.d_eg[1] = deo.data_ptr();
    ok(tgt_triplet_attention_bwd(&a, stream_of(qi)), "tgt_triplet_attention_bwd");
    return {dqi, dei, dqo, deo};
}

// ---------------------------------------------------------------------------
// triplet aggregate: per direction v (B,N,N,C) head-major, eg (B,N,N,2Ht) [E|G]; out (B,N,N,2C)
// ---------------------------------------------------------------------------
tgt_triplet_aggregate_args agg_args(const Tensor& vi, const Tensor& ei, const Tensor& vo, const Tensor& eo, const Tensor& m3, int64_t H,
                                    bool mask_out) {
    TORCH_CHECK(vi.dim() == 4 && vi.sizes() == vo.sizes() && ei.sizes() == eo.sizes() && vi.size(3) % H == 0 && ei.size(3) == 2 * H,
                "tgt::triplet_aggregate: v_in/out (B,N,N,C), eg_in/out (B,N,N,2H) expected");
    tgt_triplet_aggregate_args a{};
    const int64_t C = vi.size(3);
    a.B = (int32_t)vi.size(0); a.N = (int32_t)vi.size(1); a.H = (int32_t)H; a.D = (int32_t)(C / H);
    a.dtype = dtype_code(vi); a.flags = TGT_TRI_BIASED | TGT_TRI_GATED | (mask_out ? TGT_TRI_MASK_OUT : 0);
    const Tensor* v[2] = {&vi, &vo};
    const Tensor* e[2] = {&ei, &eo};
    for (int d = 0; d < 2; ++d) {
        a.v[d] = v[d]->data_ptr(); a.ld_v[d] = C; a.v_off[d] = 0;
        a.eg[d] = e[d]->data_ptr(); a.ld_eg[d] = 2 * H; a.e_off[d] = 0; a.g_off[d] = (int32_t)H;
        a.o_off[d] = (int32_t)(d * C);
    }
    a.mask = m3.data_ptr<float>();
    a.ld_out = 2 * C;
    return a;
}

Tensor triplet_aggregate_fwd(const Tensor& v_in, const Tensor& eg_in, const Tensor& v_out, const Tensor& eg_out, const Tensor& mask, int64_t H,
                             bool mask_out) {
    on_gpu(v_in, "v_in"); on_gpu(v_out, "v_out"); on_gpu(eg_in, "eg_in"); on_gpu(eg_out, "eg_out");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(v_in.device());
    const auto dt = v_in.scalar_type();
    const Tensor vi = v_in.contiguous(), vo = v_out.to(dt).contiguous(), ei = eg_in.to(dt).contiguous(), eo = eg_out.to(dt).contiguous();
    const int64_t B = vi.size(0), N = vi.size(1), C = vi.size(3);
    const Tensor m3 = mask3(mask, B, N);
    tgt_triplet_aggregate_args a = agg_args(vi, ei, vo, eo, m3, H, mask_out);
    Tensor out = at::empty({B, N, N, 2 * C}, vi.options());
    a.out = out.data_ptr();
    ok(tgt_triplet_aggregate_fwd(&a, stream_of(vi)), "tgt_triplet_aggregate_fwd");
    return out;
}

std::tuple<Tensor, Tensor, Tensor, Tensor> triplet_aggregate_bwd(const Tensor& v_in, const Tensor& eg_in, const Tensor& v_out,
                                                                 const Tensor& eg_out, const Tensor& mask, const Tensor& d_out_, int64_t H,
                                                                 bool mask_out) {
    on_gpu(v_in, "v_in"); on_gpu(d_out_, "d_out");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(v_in.device());
    const auto dt = v_in.scalar_type();
    const Tensor vi = v_in.contiguous(), vo = v_out.to(dt).contiguous(), ei = eg_in.to(dt).contiguous(), eo = eg_out.to(dt).contiguous();
    const int64_t B = vi.size(0), N = vi.size(1);
    const Tensor m3 = mask3(mask, B, N);
    tgt_triplet_aggregate_args a = agg_args(vi, ei, vo, eo, m3, H, mask_out);
    const Tensor d_out = d_out_.to(dt).contiguous();
    Tensor dvi = at::empty_like(vi), dvo = at::empty_like(vo), dei = at::empty_like(ei), deo = at::empty_like(eo);
    a.d_out = d_out.data_ptr();
    a.out = d_out.data_ptr();          // (checked non-null; the backward does not touch it)
    a.d_v[0] = dvi.data_ptr(); a.d_v[1] = dvo.data_ptr(); a.d_eg[0] = dei.data_ptr(); a.d_eg[1] = deo.data_ptr();
    ok(tgt_triplet_aggregate_bwd(&a, stream_of(vi)), "tgt_triplet_aggregate_bwd");
    return {dvi, dei, dvo, deo};
}

// ---------------------------------------------------------------------------
// differentiable entry points (the autograd node owns what the backward needs)
// ---------------------------------------------------------------------------
struct EgtAttention : public torch::autograd::Function<EgtAttention> {
    static torch::autograd::variable_list forward(torch::autograd::AutogradContext* ctx, const Tensor& qkv, const Tensor& eg, const Tensor& mask,
                                                  int64_t H, bool scale_degree, bool want_edges) {
        at::AutoDispatchBelowADInplaceOrView g;
        auto [vatt, hhat, lse, gsum] = egt_attention_fwd(qkv, eg, mask, H, scale_degree, want_edges);
        ctx->save_for_backward({qkv, eg, mask, vatt, lse, gsum});
        ctx->saved_data["H"] = H;
        ctx->saved_data["sd"] = scale_degree;
        return {vatt, hhat};
    }
    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
        const auto s = ctx->get_saved_variables();
        Tensor d_vatt = grads[0].defined() ? grads[0] : at::zeros_like(s[3]);
        c10::optional<Tensor> d_hhat;
        if (grads[1].defined()) d_hhat = grads[1];
        auto [d_qkv, d_eg] = egt_attention_bwd(s[0], s[1], s[2], s[3], s[4], s[5], d_vatt, d_hhat, ctx->saved_data["H"].toInt(),
                                               ctx->saved_data["sd"].toBool());
        return {d_qkv, d_eg.to(s[1].scalar_type()), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};
std::tuple<Tensor, Tensor> egt_attention(const Tensor& qkv, const Tensor& eg, const Tensor& mask, int64_t H, bool scale_degree, bool want_edges) {
    auto r = EgtAttention::apply(qkv, eg, mask, H, scale_degree, want_edges);
    return {r[0], r[1]};
}

struct TripletAttention : public torch::autograd::Function<TripletAttention> {
    static Tensor forward(torch::autograd::AutogradContext* ctx, const Tensor& qi, const Tensor& ei, const Tensor& qo, const Tensor& eo,
                          const Tensor& mask, int64_t H) {
        at::AutoDispatchBelowADInplaceOrView g;
        ctx->save_for_backward({qi, ei, qo, eo, mask});
        ctx->saved_data["H"] = H;
        return triplet_attention_fwd(qi, ei, qo, eo, mask, H);
    }
    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
        const auto s = ctx->get_saved_variables();
        auto [dqi, dei, dqo, deo] = triplet_attention_bwd(s[0], s[1], s[2], s[3], s[4], grads[0], ctx->saved_data["H"].toInt());
        return {dqi, dei.to(s[1].scalar_type()), dqo.to(s[2].scalar_type()), deo.to(s[3].scalar_type()), Tensor(), Tensor()};
    }
};
Tensor triplet_attention(const Tensor& qi, const Tensor& ei, const Tensor& qo, const Tensor& eo, const Tensor& mask, int64_t H) {
    return TripletAttention::apply(qi, ei, qo, eo, mask, H);
}

struct TripletAggregate : public torch::autograd::Function<TripletAggregate> {
    static Tensor forward(torch::autograd::AutogradContext* ctx, const Tensor& vi, const Tensor& ei, const Tensor& vo, const Tensor& eo,
                          const Tensor& mask, int64_t H, bool mask_out) {
        at::AutoDispatchBelowADInplaceOrView g;
        ctx->save_for_backward({vi, ei, vo, eo, mask});
        ctx->saved_data["H"] = H;
        ctx->saved_data["mo"] = mask_out;
        return triplet_aggregate_fwd(vi, ei, vo, eo, mask, H, mask_out);
    }
    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
        const auto s = ctx->get_saved_variables();
        auto [dvi, dei, dvo, deo] = triplet_aggregate_bwd(s[0], s[1], s[2], s[3], s[4], grads[0], ctx->saved_data["H"].toInt(),
                                                          ctx->saved_data["mo"].toBool());
        return {dvi, dei.to(s[1].scalar_type()), dvo.to(s[2].scalar_type()), deo.to(s[3].scalar_type()), Tensor(), Tensor(), Tensor()};
    }
};
Tensor triplet_aggregate(const Tensor& vi, const Tensor& ei, const Tensor& vo, const Tensor& eo, const Tensor& mask, int64_t H, bool mask_out) {
    return TripletAggregate::apply(vi, ei, vo, eo, mask, H, mask_out);
}

}  // namespace

TORCH_LIBRARY(tgt, m) {
    m.def("abi_version() -> int", []() -> int64_t { return tgt_abi_version(); });
    m.def("egt_attention_fwd(Tensor qkv, Tensor eg, Tensor mask, int num_heads, bool scale_degree, bool want_edges) -> (Tensor, Tensor, Tensor, Tensor)");
    m.def("egt_attention_bwd(Tensor qkv, Tensor eg, Tensor mask, Tensor vatt, Tensor lse, Tensor gsum, Tensor d_vatt, Tensor? d_hhat, int num_heads, bool scale_degree) -> (Tensor, Tensor)");
    m.def("egt_attention(Tensor qkv, Tensor eg, Tensor mask, int num_heads, bool scale_degree, bool want_edges) -> (Tensor, Tensor)");
    m.def("triplet_attention_fwd(Tensor qkv_in, Tensor eg_in, Tensor qkv_out, Tensor eg_out, Tensor mask, int num_heads) -> Tensor");
    m.def("triplet_attention_bwd(Tensor qkv_in, Tensor eg_in, Tensor qkv_out, Tensor eg_out, Tensor mask, Tensor d_out, int num_heads) -> (Tensor, Tensor, Tensor, Tensor)");
    m.def("triplet_attention(Tensor qkv_in, Tensor eg_in, Tensor qkv_out, Tensor eg_out, Tensor mask, int num_heads) -> Tensor");
    m.def("triplet_aggregate_fwd(Tensor v_in, Tensor eg_in, Tensor v_out, Tensor eg_out, Tensor mask, int num_heads, bool mask_out) -> Tensor");
    m.def("triplet_aggregate_bwd(Tensor v_in, Tensor eg_in, Tensor v_out, Tensor eg_out, Tensor mask, Tensor d_out, int num_heads, bool mask_out) -> (Tensor, Tensor, Tensor, Tensor)");
    m.def("triplet_aggregate(Tensor v_in, Tensor eg_in, Tensor v_out, Tensor eg_out, Tensor mask, int num_heads, bool mask_out) -> Tensor");
}

// the kernels only exist for the GPU: registered for the CUDA (= HIP on ROCm) dispatch key; a CPU tensor gets the
// dispatcher's "no kernel for backend CPU" error
TORCH_LIBRARY_IMPL(tgt, CUDA, m) {
    m.impl("egt_attention_fwd", &egt_attention_fwd);
    m.impl("egt_attention_bwd", &egt_attention_bwd);
    m.impl("triplet_attention_fwd", &triplet_attention_fwd);
    m.impl("triplet_attention_bwd", &triplet_attention_bwd);
    m.impl("triplet_aggregate_fwd", &triplet_aggregate_fwd);
    m.impl("triplet_aggregate_bwd", &triplet_aggregate_bwd);
}
TORCH_LIBRARY_IMPL(tgt, Autograd, m) {
    m.impl("egt_attention", &egt_attention);
    m.impl("triplet_attention", &triplet_attention);
    m.impl("triplet_aggregate", &triplet_aggregate);
}
