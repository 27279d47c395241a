#!/usr/bin/env python
"""Micro-benchmark of the hand-written kernels at the BASELINE shapes
(B=256, N=32, C=256, Ht=16; W=768, Hn=64), HIP events on the launch stream.
Prints algorithmic GB/s per kernel (bytes as in DESIGN.md §4)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgt_amd import ops  # noqa: E402


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--B', type=int, default=256)
    ap.add_argument('--N', type=int, default=32)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--only', default='')
    a = ap.parse_args()
    dt = {'bf16': torch.bfloat16, 'fp16': torch.float16, 'fp32': torch.float32}[a.dtype]
    esz = 4 if a.dtype == 'fp32' else 2
    B, N, C, Ht, W, Hn = a.B, a.N, 256, 16, 768, 64
    dev = 'cuda'
    torch.manual_seed(0)
    mask = torch.zeros(B, N, N, device=dev)
    out = {}
    n2 = N * N

    if not a.only or 'tri' in a.only:
        L = ops.TripletLayout(C, Ht)
        fused = torch.randn(B, N, N, L.width, device=dev, dtype=dt).requires_grad_(True)
        va = ops.triplet_attention(fused, mask, L)
        g = torch.randn_like(va)
        fb = B * (2 * (4 * n2 * C + 2 * n2 * Ht) * esz + n2 * 4)
        bb = B * (2 * (7 * n2 * C + 4 * n2 * Ht) * esz + n2 * 4)
        t = timeit(lambda: ops.triplet_attention(fused, mask, L), a.iters)
        out['tri_att_fwd'] = dict(ms=round(t, 4), GBs=round(fb / t / 1e6, 1))
        t2 = timeit(lambda: torch.autograd.grad(ops.triplet_attention(fused, mask, L), fused, g), a.iters)
        out['tri_att_bwd'] = dict(ms=round(t2 - t, 4), GBs=round(bb / (t2 - t) / 1e6, 1))
        del fused, va, g

    if 'tricol' in a.only:
        # the backward as the training step runs it: projection + attention as one autograd node,
        # bias gradient accumulated inside the backward kernel (CS variant)
        L = ops.TripletLayout(C, Ht)
        x = torch.randn(B, N, N, C, device=dev, dtype=dt).requires_grad_(True)
        w = (torch.randn(L.width, C, device=dev) * C ** -0.5).to(dt).requires_grad_(True)
        bias = torch.randn(L.width, device=dev).to(dt).requires_grad_(True)
        g = torch.randn(B, N, N, 2 * C, device=dev, dtype=dt)
        prof = ops.profile_kernels(True)
        for _ in range(a.iters):
            torch.autograd.grad(ops.projected_triplet_attention(x, w, bias, mask, L), (x, w, bias), g)
        torch.cuda.synchronize()
        ops.profile_kernels(False)
        out['tricol'] = {k: round(sum(v[1:]) / max(1, len(v) - 1), 4) for k, v in ops.kernel_times_ms(prof).items()}
        del x, w, bias, g

    if 'proj' in a.only:
        # projection + attention forward as the training step runs it: fused kernel (TGT_TRI_PROJ=1, default) or GEMM + GEMM + attention
        L = ops.TripletLayout(C, Ht)
        x = torch.randn(B, N, N, C, device=dev, dtype=dt)
        w = (torch.randn(L.width, C, device=dev) * C ** -0.5).to(dt)
        bias = torch.randn(L.width, device=dev).to(dt)
        with torch.no_grad():
            t = timeit(lambda: ops.projected_triplet_attention(x, w, bias, mask, L), a.iters)
        out['proj+tri_att_fwd'] = dict(ms=round(t, 4), fused=bool(ops._TRI_PROJ))

    if not a.only or 'agg' in a.only:
        L = ops.AggregateLayout(C, Ht)
        fused = torch.randn(B, N, N, L.width, device=dev, dtype=dt).requires_grad_(True)
        g = torch.randn(B, N, N, 2 * C, device=dev, dtype=dt)
        fb = B * (2 * (2 * n2 * C + 2 * n2 * Ht) * esz + n2 * 4)
        t = timeit(lambda: ops.triplet_aggregate(fused, mask, L), a.iters)
        out['tri_agg_fwd'] = dict(ms=round(t, 4), GBs=round(fb / t / 1e6, 1))
        t2 = timeit(lambda: torch.autograd.grad(ops.triplet_aggregate(fused, mask, L), fused, g), a.iters)
        bb = B * (2 * (3 * n2 * C + 4 * n2 * Ht) * esz + n2 * 4)
        out['tri_agg_bwd'] = dict(ms=round(t2 - t, 4), GBs=round(bb / (t2 - t) / 1e6, 1))
        del fused, g

    if not a.only or 'node' in a.only:
        qkv = torch.randn(B, N, 3 * W, device=dev, dtype=dt).requires_grad_(True)
        eg = torch.randn(B, N, N, 2 * Hn, device=dev, dtype=dt).requires_grad_(True)
        gv = torch.randn(B, N, W, device=dev, dtype=dt)
        gh = torch.randn(B, N, N, Hn, device=dev, dtype=dt)
        fb = B * ((4 * N * W + 3 * n2 * Hn) * esz + n2 * 4)
        bb = B * ((7 * N * W + 5 * n2 * Hn) * esz + n2 * 4)
        t = timeit(lambda: ops.node_attention(qkv, eg, mask, Hn), a.iters)
        out['node_att_fwd'] = dict(ms=round(t, 4), GBs=round(fb / t / 1e6, 1))

        def fb_():
            v, h = ops.node_attention(qkv, eg, mask, Hn)
            torch.autograd.grad([v, h], [qkv, eg], [gv, gh])
        t2 = timeit(fb_, a.iters)
        out['node_att_bwd'] = dict(ms=round(t2 - t, 4), GBs=round(bb / (t2 - t) / 1e6, 1))
        del qkv, eg

    if not a.only or 'ln' in a.only:
        x = torch.randn(B * n2, C, device=dev, dtype=dt).requires_grad_(True)
        w, b = torch.ones(C, device=dev, requires_grad=True), torch.zeros(C, device=dev, requires_grad=True)
        g = torch.randn(B * n2, C, device=dev, dtype=dt)
        t = timeit(lambda: ops.layer_norm(x, w, b, 1e-5, dt), a.iters)
        out['ln_fwd_edge'] = dict(ms=round(t, 4), GBs=round(2 * x.numel() * esz / t / 1e6, 1))
        t2 = timeit(lambda: torch.autograd.grad(ops.layer_norm(x, w, b, 1e-5, dt), [x, w, b], g), a.iters)
        out['ln_bwd_edge'] = dict(ms=round(t2 - t, 4), GBs=round(3 * x.numel() * esz / (t2 - t) / 1e6, 1))
        res = torch.randn(B, N, N, C, device=dev, dtype=dt)
        x4 = x.detach().view(B, N, N, C)
        sc = torch.ones(B, device=dev)
        t = timeit(lambda: ops.add_layer_norm(x4, res, sc, w, b, 1e-5, dt), a.iters)
        out['add_ln_fwd_edge'] = dict(ms=round(t, 4), GBs=round(4 * x.numel() * esz / t / 1e6, 1))
        xr, rr = x4.clone().requires_grad_(True), res.clone().requires_grad_(True)
        g4 = g.view(B, N, N, C)

        def add_ln_fb():
            s_, y_ = ops.add_layer_norm(xr, rr, sc, w, b, 1e-5, dt)
            return torch.autograd.grad([s_, y_], [xr, rr], [g4, g4])
        t2 = timeit(add_ln_fb, a.iters)
        out['add_ln_bwd_edge'] = dict(ms=round(t2 - t, 4), GBs=round(5 * x.numel() * esz / (t2 - t) / 1e6, 1))
        t = timeit(lambda: ops.gelu_dropout(x4, 0.1, True), a.iters)
        out['gelu_dropout_fwd'] = dict(ms=round(t, 4), GBs=round(2 * x.numel() * esz / t / 1e6, 1))
    print(json.dumps(out))


if __name__ == '__main__':
    main()
