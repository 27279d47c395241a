#!/usr/bin/env python
"""Does the 256 MiB Infinity Cache (MALL) pay for producer -> consumer hand-overs?

(1) streaming: a buffer of S bytes is written by one kernel and read by the next; time of the reader vs S.
(2) the triplet block: projection GEMM -> attention kernel on the whole batch vs. on chunks of graphs whose Q/K/V
    (6 x 0.5 MB per graph) fit the cache.
(3) backward: attention backward -> data-gradient GEMM + weight-gradient GEMM, whole vs chunked.
Prints one JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tgt_amd import ops  # noqa: E402


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
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
    dev = 'cuda'
    out = {}
    # ---- (1) write S then read S (sum) / copy S
    stream = {}
    for mb in (16, 32, 64, 96, 128, 192, 256, 384, 512, 1024):
        n = mb * 1024 * 1024 // 2
        a = torch.randn(n, device=dev, dtype=torch.bfloat16)
        b = torch.empty_like(a)
        c = torch.empty_like(a)
        big = torch.empty(1024 * 1024 * 1024 // 2, device=dev, dtype=torch.bfloat16)   # 1 GiB flusher

        def wr_rd():
            torch.add(a, 1, out=b)
            torch.add(b, 1, out=c)

        def wr_flush_rd():
            torch.add(a, 1, out=b)
            big.zero_()
            torch.add(b, 1, out=c)

        def flush_only():
            torch.add(a, 1, out=b)
            big.zero_()

        def wr_only():
            torch.add(a, 1, out=b)

        t_pair = timeit(wr_rd)
        t_wr = timeit(wr_only)
        t_fl = timeit(wr_flush_rd) - timeit(flush_only)
        # repeated read of the same buffer
        t_rd = timeit(lambda: torch.add(a, 1, out=c))
        stream[mb] = dict(copy_after_write_us=round((t_pair - t_wr) * 1e3, 1), copy_after_flush_us=round(t_fl * 1e3, 1),
                          copy_same_src_us=round(t_rd * 1e3, 1),
                          GBs_after_write=round(2 * mb * 1.048576 / (t_pair - t_wr), 1),
                          GBs_after_flush=round(2 * mb * 1.048576 / t_fl, 1))
        del a, b, c, big
    out['stream'] = stream

    # ---- (2) projection GEMM -> attention forward, whole vs chunks
    B, N, C, Ht = 256, 32, 256, 16
    dt = torch.bfloat16
    L = ops.TripletLayout(C, Ht)
    x = torch.randn(B, N, N, C, device=dev, dtype=dt)
    w = (torch.randn(L.width, C, device=dev) * C ** -0.5).to(dt)
    wq, weg = w[:6 * C].contiguous(), w[6 * C:].contiguous()
    mask = torch.zeros(B, N, N, device=dev)
    qkv = torch.empty(B, N, N, 6 * C, device=dev, dtype=dt)
    eg = torch.empty(B, N, N, L.width - 6 * C, device=dev, dtype=dt)
    va = torch.empty(B, N, N, 2 * C, device=dev, dtype=dt)
    from tgt_amd import _lib
    import ctypes
    lib = _lib.lib()

    def attn(s, e):
        args = ops._tri_args(qkv[s:e], mask[s:e], va[s:e], L, eg=eg[s:e])
        _lib.check(lib.tgt_triplet_attention_fwd(ctypes.byref(args), ops._stream()), 'fwd')

    def fwd(G):
        step = B // G
        for s in range(0, B, step):
            torch.mm(x[s:s + step].view(-1, C), wq.t(), out=qkv[s:s + step].view(-1, 6 * C))
            torch.mm(x[s:s + step].view(-1, C), weg.t(), out=eg[s:s + step].view(-1, eg.shape[-1]))
            attn(s, s + step)

    def gemm_only(G):
        step = B // G
        for s in range(0, B, step):
            torch.mm(x[s:s + step].view(-1, C), wq.t(), out=qkv[s:s + step].view(-1, 6 * C))
            torch.mm(x[s:s + step].view(-1, C), weg.t(), out=eg[s:s + step].view(-1, eg.shape[-1]))

    def attn_only(G):
        step = B // G
        for s in range(0, B, step):
            attn(s, s + step)

    fw = {}
    for G in (1, 2, 4, 8, 16):
        fw[G] = dict(pair_ms=round(timeit(lambda: fwd(G)), 4), gemm_ms=round(timeit(lambda: gemm_only(G)), 4),
                     attn_ms=round(timeit(lambda: attn_only(G)), 4))
    out['proj_attn_fwd'] = fw

    # ---- (3) attention backward -> dgrad + wgrad
    d_out = torch.randn(B, N, N, 2 * C, device=dev, dtype=dt)
    d_fused = torch.empty(B, N, N, L.width, device=dev, dtype=dt)
    dx = torch.empty(B, N, N, C, device=dev, dtype=dt)
    fused = torch.randn(B, N, N, L.width, device=dev, dtype=dt)

    def bwd_k(s, e):
        args = ops._tri_args(fused[s:e], mask[s:e], va[s:e], L, d_out=d_out[s:e], d_fused=d_fused[s:e])
        _lib.check(lib.tgt_triplet_attention_bwd(ctypes.byref(args), ops._stream()), 'bwd')

    def bwd(G, gemms=True, kern=True):
        step = B // G
        for s in range(0, B, step):
            if kern:
                bwd_k(s, s + step)
            if gemms:
                dy = d_fused[s:s + step].view(-1, L.width)
                torch.mm(dy, w, out=dx[s:s + step].view(-1, C))
                xs = x[s:s + step].view(-1, C)
                P = max(1, 64 // G)
                M = dy.shape[0]
                torch.bmm(dy.view(P, M // P, -1).transpose(1, 2), xs.view(P, M // P, -1), out_dtype=torch.float32)

    bw = {}
    for G in (1, 2, 4, 8, 16):
        bw[G] = dict(all_ms=round(timeit(lambda: bwd(G)), 4), kernel_ms=round(timeit(lambda: bwd(G, gemms=False)), 4),
                     gemms_ms=round(timeit(lambda: bwd(G, kern=False)), 4))
    out['attn_bwd_gemms'] = bw
    print(json.dumps(out))


if __name__ == '__main__':
    main()
