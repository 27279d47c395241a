"""dW = dY^T X on 262144 rows: ONE library GEMM under TunableOp (stream-K / split-K solutions included in the candidates)
against the shipped form (batched partial products into fp32 planes + tgt_sum_planes).  python tools/wgrad_tunable_probe.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgt_amd import ops
t = torch.cuda.tunable
t.enable(True); t.tuning_enable(True); t.set_max_tuning_duration(150); t.set_max_tuning_iterations(50)
t.set_filename('/tmp/wgrad_tunable_probe.csv', insert_device_ordinal=False)


def timeit(fn, it=20):
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


M = 262144
for (N, K) in [(256, 256), (1536, 256), (256, 512), (128, 256), (64, 256), (256, 64), (256, 128)]:
    dY = torch.randn(M, N, device='cuda', dtype=torch.bfloat16)
    X = torch.randn(M, K, device='cuda', dtype=torch.bfloat16)
    out = torch.empty(N, K, device='cuda', dtype=torch.float32)
    plain = timeit(lambda: dY.t() @ X)
    try:
        plain32 = timeit(lambda: torch.mm(dY.t(), X, out_dtype=torch.float32))
    except Exception as ex:
        plain32 = float('nan')
    P = ops._wgrad_chunks(M, N * K)
    shipped = timeit(lambda: ops._wgrad_into(out, dY, X, P))
    ideal = M * (N + K) * 2 / 5e12 * 1e6
    print(f'N={N:5d} K={K:4d}  plain bf16-out {plain:7.1f}us  plain fp32-out {plain32:7.1f}us  shipped (P={P}) {shipped:7.1f}us  ideal@5TB/s {ideal:6.1f}us', flush=True)
