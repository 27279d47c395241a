"""Fused data + weight gradient of a 256 x 256 edge Linear (csrc/edge_wgrad.hip) against the kernel pair it replaces, alone on the
GPU at the BASELINE shape (M = 256 * 32 * 32 rows, bf16):
    pair  = tgt_edge_linear (GELU_BWD / LN_BWD epilogue)  +  torch.bmm over 128 row chunks (fp32 partials)  +  tgt_sum_planes
    fused = tgt_edge_linear with dw_partial                +  tgt_sum_planes over the per-workgroup planes
python tools/edge_wgrad_bench.py [--m ROWS]"""
import argparse
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgt_amd import _lib, ops


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3          # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--m', type=int, default=256 * 32 * 32)
    args = ap.parse_args()
    M, N, dt, dev = args.m, 256, torch.bfloat16, 'cuda'
    g = torch.Generator(device=dev).manual_seed(0)
    a = torch.randn(M, N, device=dev, generator=g).to(dt)
    w = (torch.randn(N, N, device=dev, generator=g) / 16).to(dt)
    res = torch.randn(M, N, device=dev, generator=g).to(dt)
    x = torch.randn(M, N, device=dev, generator=g).to(dt)
    ds = torch.randn(M, N, device=dev, generator=g).to(dt)
    out, out2 = torch.empty(M, N, dtype=dt, device=dev), torch.empty(M, N, dtype=dt, device=dev)
    gamma, beta = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)
    mean, rstd = torch.zeros(M, device=dev), torch.ones(M, device=dev)
    sc = torch.ones(M // 1024 if M >= 1024 else 1, device=dev)
    rps = 1024 if M >= 1024 else M
    parts = _lib.lib().tgt_edge_linear_parts(M, N)
    dwp = torch.empty(parts, N, N, device=dev)
    dw = torch.empty(N, N, device=dev)
    P = ops._wgrad_chunks(M, N * N)
    print(f'M = {M}, {parts} persistent workgroups, library wgrad in {P} row chunks')
    for name, epi, kw in (
            ('GELU_BWD (lin_W2)', _lib.EPI_GELU_BWD, dict(res=res, out_scale=sc, rows_per_sample=rps, dropout=(0.1, 1234),
                                                         colsum_partial=torch.empty(parts, N, device=dev))),
            ('LN_BWD (lin_W1)', _lib.EPI_LN_BWD, dict(ln=(gamma, beta, 1e-5), stats=(mean, rstd), res=res, ds_in=ds, out2=out2, row_scale=sc,
                                                      rows_per_sample=rps, colsum_partial=torch.empty(parts, 3 * N, device=dev)))):
        t_d = timeit(lambda: ops.edge_linear_raw(a, w, None, epi, out=out, **kw))

        def wgrad():
            part = torch.bmm(a.view(P, M // P, -1).transpose(1, 2), x.view(P, M // P, -1), out_dtype=torch.float32)
            ops.sum_planes(part, dw, defer=False)
        t_w = timeit(wgrad)
        t_pair = timeit(lambda: (ops.edge_linear_raw(a, w, None, epi, out=out, **kw), wgrad()))
        t_f = timeit(lambda: ops.edge_linear_raw(a, w, None, epi, out=out, dw_partial=dwp, **kw))
        t_fs = timeit(lambda: (ops.edge_linear_raw(a, w, None, epi, out=out, dw_partial=dwp, **kw), ops.sum_planes(dwp, dw, defer=False)))
        print(f'{name:20s} dgrad {t_d:7.1f} us | wgrad (bmm + sum) {t_w:7.1f} | pair back to back {t_pair:7.1f} || fused launch {t_f:7.1f} | '
              f'fused + sum of {parts} planes {t_fs:7.1f}  ->  {t_pair - t_fs:+.1f} us per Linear')


if __name__ == '__main__':
    main()
