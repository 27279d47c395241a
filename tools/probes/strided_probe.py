import torch, sys, os
sys.path.insert(0,'/root/repo')
M,K=262144,256
x=torch.randn(M,K,device='cuda',dtype=torch.bfloat16); w=torch.randn(1600,K,device='cuda',dtype=torch.bfloat16); b=torch.randn(1600,device='cuda',dtype=torch.bfloat16)
fused=torch.empty(M,1600,device='cuda',dtype=torch.bfloat16)
def t(fn,n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n*1e3
print('strided out, no tunable', t(lambda: torch.addmm(b[1536:], x, w[1536:].t(), out=fused[:,1536:])))
ref=torch.addmm(b[1536:], x, w[1536:].t())
print('max diff', float((fused[:,1536:].float()-ref.float()).abs().max()))
print('contig out', t(lambda: torch.addmm(b[1536:], x, w[1536:].t())))
from tgt_amd.training import gemm_tuning
gemm_tuning.enable_gemm_tuning(online=True)
try:
    print('strided out, tunable', t(lambda: torch.addmm(b[1536:], x, w[1536:].t(), out=fused[:,1536:])))
except Exception as ex:
    print('tunable strided failed:', str(ex)[:200])
