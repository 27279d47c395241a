"""torch.ops.tgt.* (csrc/torch_ops.cpp): the dispatcher registrations on top of the C ABI."""
import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import core

OPS = ['egt_attention', 'egt_attention_fwd', 'egt_attention_bwd', 'triplet_attention', 'triplet_attention_fwd',
       'triplet_attention_bwd', 'triplet_aggregate', 'triplet_aggregate_fwd', 'triplet_aggregate_bwd']


def test_op_library_loads_and_registers_every_schema():
    from tgt_amd import torch_ops, _lib
    torch_ops.build_op_library()
    t = torch_ops.load()
    assert t.abi_version() == _lib.ABI_VERSION
    for name in OPS:
        assert hasattr(t, name), name
        assert str(getattr(t, name).default._schema).startswith(f'tgt::{name}(')


def test_op_library_exports_the_gemm_dispatch_abi():
    """include/tgt_gemm.h: every declared symbol is exported by libtgt_torch_ops.so, and a C program compiles against the header"""
    import ctypes
    import os
    import re
    import subprocess
    import tempfile
    from tgt_amd import torch_ops
    torch_ops.build_op_library()
    torch_ops.load()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, 'include', 'tgt_gemm.h')).read()
    declared = set(re.findall(r'\b(tgt_gemm_[a-z_0-9]+)\s*\(', hdr))
    assert declared == {'tgt_gemm_plan', 'tgt_gemm_run', 'tgt_gemm_last_error'}
    L = ctypes.CDLL(torch_ops.OPS_LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, 't.c')
        open(c, 'w').write('#include "tgt_gemm.h"\nint main(void){ return tgt_gemm_plan == 0; }\n')
        subprocess.check_call(['gcc', '-c', '-I', os.path.join(root, 'include'), c, '-o', os.path.join(td, 't.o')])
    # without a GPU nothing can be planned; the host wrappers then go through torch (same library, same call)
    from tgt_amd import gemm
    x, w, b = torch.randn(6, 4), torch.randn(3, 4), torch.randn(3)
    assert torch.allclose(gemm.linear_tn(x, w, b), x @ w.t() + b) and torch.allclose(gemm.matmul_nn(x, w.t().contiguous()), x @ w.t())


def test_cpu_tensors_raise():
    from tgt_amd import torch_ops
    t = torch_ops.load()
    z = torch.zeros
    with pytest.raises((RuntimeError, NotImplementedError)):
        t.triplet_attention_fwd(z(1, 2, 2, 48), z(1, 2, 2, 4), z(1, 2, 2, 48), z(1, 2, 2, 4), z(1, 2, 2), 2)


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_registered_ops_equal_the_ctypes_path(dtype):
    """same kernels behind both bindings: outputs and gradients must be bit-identical"""
    from tgt_amd import ops, torch_ops
    t = torch_ops.load()
    B, N, C, H = 3, 20, 64, 4
    rng = np.random.default_rng(5)
    rnd = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32))
    mask = gu.additive_mask([20, 13, 7], N, torch.float32).reshape(B, N, N).cuda()
    # triplet attention
    L = ops.TripletLayout(C, H)
    fused = rnd(B, N, N, L.width).to(dtype).cuda()
    d_out = rnd(B, N, N, 2 * C).to(dtype).cuda()
    f1 = fused.clone().requires_grad_(True)
    va1 = ops.triplet_attention(f1, mask, L)
    va1.backward(d_out)
    parts = [fused[..., :3 * C], fused[..., 6 * C:6 * C + 2 * H], fused[..., 3 * C:6 * C], fused[..., 6 * C + 2 * H:6 * C + 4 * H]]
    parts = [p.clone().requires_grad_(True) for p in parts]
    va2 = t.triplet_attention(parts[0], parts[1], parts[2], parts[3], mask, H)
    va2.backward(d_out)
    assert torch.equal(va1, va2)
    g = f1.grad
    assert torch.equal(g[..., :3 * C], parts[0].grad) and torch.equal(g[..., 3 * C:6 * C], parts[2].grad)
    assert torch.equal(g[..., 6 * C:6 * C + 2 * H], parts[1].grad) and torch.equal(g[..., 6 * C + 2 * H:6 * C + 4 * H], parts[3].grad)
    # triplet aggregate
    LA = ops.AggregateLayout(C, H)
    fa = rnd(B, N, N, LA.width).to(dtype).cuda()
    a1 = fa.clone().requires_grad_(True)
    o1 = ops.triplet_aggregate(a1, mask, LA)
    o1.backward(d_out)
    ap = [fa[..., :C], fa[..., 2 * C:2 * C + 2 * H], fa[..., C:2 * C], fa[..., 2 * C + 2 * H:2 * C + 4 * H]]
    ap = [p.clone().requires_grad_(True) for p in ap]
    o2 = t.triplet_aggregate(ap[0], ap[1], ap[2], ap[3], mask, H, False)
    o2.backward(d_out)
    assert torch.equal(o1, o2)
    assert torch.equal(a1.grad[..., :C], ap[0].grad) and torch.equal(a1.grad[..., 2 * C:2 * C + 2 * H], ap[1].grad)
    # node attention
    W, Hn = 96, 8
    qkv, eg = rnd(B, N, 3 * W).to(dtype).cuda(), rnd(B, N, N, 2 * Hn).to(dtype).cuda()
    gv, gh = rnd(B, N, W).to(dtype).cuda(), rnd(B, N, N, Hn).to(dtype).cuda()
    q1, e1 = qkv.clone().requires_grad_(True), eg.clone().requires_grad_(True)
    v1, h1 = ops.node_attention(q1, e1, mask, Hn, True, True)
    torch.autograd.backward([v1, h1], [gv, gh])
    q2, e2 = qkv.clone().requires_grad_(True), eg.clone().requires_grad_(True)
    v2, h2 = t.egt_attention(q2, e2, mask, Hn, True, True)
    torch.autograd.backward([v2, h2], [gv, gh])
    assert torch.equal(v1, v2) and torch.equal(h1, h2)
    assert torch.equal(q1.grad, q2.grad) and torch.equal(e1.grad, e2.grad)


@pytest.mark.gpu
def test_registered_triplet_attention_vs_oracle():
    """the registered op against the float64 oracle directly (reference lib/tgt/layers/triplet.py:213-246)"""
    from tgt_amd import torch_ops, layout
    t = torch_ops.load()
    B, N, C, H = 2, 9, 32, 4
    rng = np.random.default_rng(11)
    rnd = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32))
    mask = gu.additive_mask([9, 5], N, torch.float32)
    qkv_in, qkv_out, eg_in, eg_out = rnd(B, N, N, 3 * C), rnd(B, N, N, 3 * C), rnd(B, N, N, 2 * H), rnd(B, N, N, 2 * H)
    d_out = rnd(B, N, N, 2 * C)
    ins = [x.cuda().requires_grad_(True) for x in (qkv_in, eg_in, qkv_out, eg_out)]
    va = t.triplet_attention(ins[0], ins[1], ins[2], ins[3], mask.reshape(B, N, N).cuda(), H)
    va.backward(d_out.cuda())
    # oracle works in the reference's head-minor channel order: permute the head-major tensors
    idx = layout.head_major_index(C, H)                     # reference channel of head-major position
    oidx = layout.va_cols_head_major(C, H)
    ref_in = [x.double().requires_grad_(True) for x in (qkv_in, eg_in, qkv_out, eg_out)]

    def to_ref(x):
        out = torch.empty_like(x)
        out[..., idx] = x
        return out
    blk = lambda x: torch.cat([to_ref(x[..., i * C:(i + 1) * C]) for i in range(3)], -1)
    va_ref = core.triplet_attention_core(blk(ref_in[0]), ref_in[1], blk(ref_in[2]), ref_in[3], mask.double(), H)
    va_ref_hm = va_ref[..., oidx]
    (va_ref_hm * d_out.double()).sum().backward()
    assert rel(va, va_ref_hm) < 2e-5
    for a, b in zip(ins, ref_in):
        assert rel(a.grad, b.grad) < 4e-5
