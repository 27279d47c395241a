"""Time the fused-projection GEMM shapes (forward / dgrad) under TunableOp: one 1600-wide GEMM
vs splits.  python tools/gemm_probe.py"""
import torch
import torch.nn.functional as F
t = torch.cuda.tunable
t.enable(True); t.tuning_enable(True); t.set_max_tuning_duration(100); t.set_max_tuning_iterations(30)
t.set_filename('/tmp/gemm_probe_tunable.csv', insert_device_ordinal=False)
M, K = 262144, 256
x = torch.randn(M, K, device='cuda', dtype=torch.bfloat16)


def timeit(fn, it=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


for N in (1600, 1536, 1664, 800, 768, 832, 1024, 576, 512, 256):
    w = torch.randn(N, K, device='cuda', dtype=torch.bfloat16)
    b = torch.randn(N, device='cuda', dtype=torch.bfloat16)
    dy = torch.randn(M, N, device='cuda', dtype=torch.bfloat16)
    out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    fwd = timeit(lambda: torch.addmm(b, x, w.t(), out=out))
    dg = timeit(lambda: dy @ w)
    ideal_f = (M * K + M * N) * 2 / 5e12 * 1e6
    print(f'N={N:5d} fwd {fwd:7.1f}us (ideal {ideal_f:6.1f})  dgrad {dg:7.1f}us  per-col fwd {fwd/N*1000:6.1f}ns', flush=True)
