"""Are the GELU row kernels VALU-issue-bound?  The fused lin_W1 + GELU + dropout launch and the GELU_BWD data-gradient launch at the BASELINE
shape with dropout p = 0.1 (hash + compare per element) and p = 0 (hash skipped): python tools/probes/gelu_valu_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tgt_amd import _lib, ops


def timeit(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


M, N, dt, dev = 262144, 256, torch.bfloat16, 'cuda'
g = torch.Generator(device=dev).manual_seed(0)
a = torch.randn(M, N, device=dev, generator=g).to(dt)
w = (torch.randn(N, N, device=dev, generator=g) / 16).to(dt)
b = torch.randn(N, device=dev, generator=g).to(dt)
res = torch.randn(M, N, device=dev, generator=g).to(dt)
out, out2 = torch.empty(M, N, dtype=dt, device=dev), torch.empty(M, N, dtype=dt, device=dev)
sc = torch.ones(256, device=dev)
for p in (0.1, 0.0, 0.1, 0.0):
    t1 = timeit(lambda: ops.edge_linear_raw(a, w, b, _lib.EPI_GELU, out=out, out2=out2, dropout=(p, 1234), row_scale=sc, rows_per_sample=1024))
    t2 = timeit(lambda: ops.edge_linear_raw(a, w, None, _lib.EPI_GELU_BWD, out=out, res=res, out_scale=sc, rows_per_sample=1024, dropout=(p, 1234)))
    t3 = timeit(lambda: ops.edge_linear_raw(a, w, b, _lib.EPI_RESID, out=out, res=res, row_scale=sc, rows_per_sample=1024))
    print(f'p = {p}: lin_W1 + GELU + dropout {t1:6.1f} us | dgrad + GELU_BWD {t2:6.1f} us | (plain residual epilogue, no GELU: {t3:6.1f} us)')
