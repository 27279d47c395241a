"""host cost of one library GEMM dispatch through torch (TunableOp on / off), GPU starved (tiny problem)"""
import os, sys, time, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
dev = torch.device('cuda', 0)
a = torch.randn(1024, 256, device=dev, dtype=torch.bfloat16)
w = torch.randn(256, 256, device=dev, dtype=torch.bfloat16)
b = torch.randn(256, device=dev, dtype=torch.bfloat16)
o = torch.empty(1024, 256, device=dev, dtype=torch.bfloat16)
a3 = torch.randn(8, 128, 256, device=dev, dtype=torch.bfloat16)
def t(fn, n=3000):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    dt = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return round(dt, 2)
def suite(tag):
    print(tag, 'mm(out=)', t(lambda: torch.mm(a, w.t(), out=o)), 'addmm(out=)', t(lambda: torch.addmm(b, a, w.t(), out=o)),
          'a @ w', t(lambda: a @ w), 'bmm fp32 out', t(lambda: torch.bmm(a3.transpose(1, 2), a3, out_dtype=torch.float32)),
          'empty', t(lambda: torch.empty(1024, 256, device=dev, dtype=torch.bfloat16)), 'us per call')
suite('default (no TunableOp)')
from tgt_amd.training.gemm_tuning import enable_gemm_tuning
enable_gemm_tuning(online=True)
suite('TunableOp on (first pass tunes)')
suite('TunableOp on')
