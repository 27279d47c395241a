"""tgt_sum_planes vs ATen's sum(0) on the weight-gradient partial shapes.  python tools/probes/plane_sum_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tgt_amd import ops


def t(fn, it=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


for shape in [(8, 768, 768), (8, 2304, 768), (128, 256, 256), (128, 128, 256), (128, 256, 512), (32, 1536, 256), (128, 64, 256), (128, 256, 64)]:
    part = torch.randn(*shape, device='cuda')
    out = torch.empty(shape[1:], device='cuda')
    a = t(lambda: torch.sum(part, 0, out=out))
    b = t(lambda: ops.sum_planes(part, out))
    mb = part.numel() * 4 / 1e6
    print(f'{str(shape):18s} {mb:6.1f} MB  aten {a:6.1f} us  tgt {b:6.1f} us  ({mb / b * 1e-6 * 1e6 / 1e3:.2f} TB/s)')
