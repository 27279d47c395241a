"""Micro-benchmark of tgt_edge_linear against the library GEMM (+ the passes it absorbs) at the BASELINE shapes
(M = 256*32*32 edge rows, bf16).  python tools/edge_gemm_bench.py"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgt_amd import _lib, ops


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def main():
    M = 256 * 32 * 32
    dt = torch.bfloat16
    dev = 'cuda'
    g = torch.Generator(device=dev).manual_seed(0)
    rows = []
    for name, K, N, epi, ln in [('lin_EG', 256, 128, _lib.EPI_BIAS, False), ('tri QKV', 256, 1536, _lib.EPI_BIAS, False),
                                ('tri EG', 256, 64, _lib.EPI_BIAS, False),
                                ('W1 (GELU)', 256, 256, _lib.EPI_GELU, False), ('W2 (+res+LN)', 256, 256, 'resid_ln', False),
                                ('lin_O_e (+res+LN)', 64, 256, 'resid_ln', False),
                                ('dgrad lin_EG + LN_BWD', 128, 256, _lib.EPI_LN_BWD, False),
                                ('W2 (+res)', 256, 256, _lib.EPI_RESID, False), ('lin_O (+res+LN)', 512, 256, 'resid_ln', False), ('lin_O_e (+res)', 64, 256, _lib.EPI_RESID, False),
                                ('plain 256x256', 256, 256, _lib.EPI_BIAS, False), ('dgrad lin_O (256->512)', 256, 512, _lib.EPI_BIAS, False),
                                ('dgrad W1 + LN_BWD', 256, 256, _lib.EPI_LN_BWD, False),
                                ('dgrad W2 + GELU_BWD', 256, 256, _lib.EPI_GELU_BWD, False)]:
        a = torch.randn(M, K, device=dev, generator=g).to(dt)
        w = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).to(dt)
        b = torch.randn(N, device=dev, generator=g).to(dt)
        gamma, beta = torch.rand(K if ln else N, device=dev) + 0.5, torch.randn(K if ln else N, device=dev)
        mean, rstd = torch.zeros(M, device=dev), torch.ones(M, device=dev)
        res = torch.randn(M, N, device=dev, generator=g).to(dt)
        out, out2, y = torch.empty(M, N, dtype=dt, device=dev), torch.empty(M, N, dtype=dt, device=dev), torch.empty(M, K, dtype=dt, device=dev)
        sc = torch.ones(256, device=dev)
        kw = {}
        if epi == 'resid_ln':
            epi = _lib.EPI_RESID
            kw.update(res=res, row_scale=sc, rows_per_sample=1024, ln=(gamma, beta, 1e-5), stats=(mean, rstd), y=out2)
        elif ln:
            kw.update(ln=(gamma, beta, 1e-5), stats=(mean, rstd), y=y)
        if epi == _lib.EPI_GELU:
            kw.update(out2=out2, dropout=(0.1, 1234))
        if epi == _lib.EPI_RESID and 'res' not in kw:
            kw.update(res=res, row_scale=sc, rows_per_sample=1024)
        if epi == _lib.EPI_GELU_BWD:
            kw.update(res=res, dropout=(0.1, 1234))
        if epi == _lib.EPI_LN_BWD:
            parts = _lib.lib().tgt_edge_linear_parts(M, N)
            kw.update(ln=(gamma, None, 1e-5), stats=(mean, rstd), res=res, ds_in=out2.clone(), out2=out2, row_scale=sc, rows_per_sample=1024,
                      colsum_partial=torch.empty(parts, 3 * N, device=dev))
        t_mine = timeit(lambda: ops.edge_linear_raw(a, w, None if epi in (_lib.EPI_LN_BWD, _lib.EPI_GELU_BWD) else b, epi, out=out, **kw))
        t_lib = timeit(lambda: torch.addmm(b, a, w.t(), out=out))
        flops = 2.0 * M * K * N
        rows.append((name, K, N, t_mine, t_lib, flops / t_mine / 1e9, flops / t_lib / 1e9))
    print(f'{"op":24s} {"K":>5s} {"N":>5s} {"edge_linear ms":>15s} {"addmm ms":>10s} {"TF/s":>7s} {"lib TF/s":>9s}')
    for r in rows:
        print(f'{r[0]:24s} {r[1]:5d} {r[2]:5d} {r[3]:15.4f} {r[4]:10.4f} {r[5]:7.0f} {r[6]:9.0f}')
    # the standalone LayerNorm passes the fused entries compete with (same rows)
    x = torch.randn(M, 256, device=dev, generator=g).to(dt)
    res = torch.randn(M, 256, device=dev, generator=g).to(dt)
    gamma, beta = torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev)
    sc = torch.ones(256, device=dev)
    xr, rr = x.view(256, 32, 32, 256).clone().requires_grad_(True), res.view(256, 32, 32, 256).clone().requires_grad_(True)
    t_ln = timeit(lambda: ops.layer_norm(x, gamma, beta, 1e-5, out_dtype=dt))
    t_aln = timeit(lambda: ops.add_layer_norm(xr, rr, sc, gamma, beta, 1e-5))
    s_, y_ = ops.add_layer_norm(xr, rr, sc, gamma, beta, 1e-5)
    gs, gy = torch.randn_like(s_), torch.randn_like(y_)
    t_alnb = timeit(lambda: torch.autograd.grad([s_, y_], [xr, rr], [gs, gy], retain_graph=True))
    print(f'layer_norm fwd {t_ln:.4f} ms   add+layer_norm fwd {t_aln:.4f} ms   add+layer_norm bwd {t_alnb:.4f} ms')


if __name__ == '__main__':
    main()
