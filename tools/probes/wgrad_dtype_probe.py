"""fp32-out (untuned: TunableOp has no entry for it) vs bf16-out (TunableOp) batched wgrad GEMMs."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tgt_amd.training import gemm_tuning
gemm_tuning.enable_gemm_tuning(online=True, filename='/tmp/wg_tune.csv', max_ms=100, max_iters=30)
M = 262144
def timeit(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
for (N, K, P) in [(1600, 256, 64), (256, 512, 128), (256, 256, 128), (128, 256, 128), (256, 64, 128)]:
    dY = torch.randn(M, N, device='cuda', dtype=torch.bfloat16); X = torch.randn(M, K, device='cuda', dtype=torch.bfloat16)
    a, b = dY.view(P, M // P, N).transpose(1, 2), X.view(P, M // P, K)
    t32 = timeit(lambda: torch.bmm(a, b, out_dtype=torch.float32).sum(0))
    t16 = timeit(lambda: torch.bmm(a, b).sum(0, dtype=torch.float32))
    ref = (dY.float().t() @ X.float())
    e32 = float((torch.bmm(a, b, out_dtype=torch.float32).sum(0) - ref).norm() / ref.norm())
    e16 = float((torch.bmm(a, b).sum(0, dtype=torch.float32) - ref).norm() / ref.norm())
    print(f'out={N} in={K} P={P}: fp32 partials {t32:.0f}us (err {e32:.1e})   bf16 partials {t16:.0f}us (err {e16:.1e})', flush=True)
