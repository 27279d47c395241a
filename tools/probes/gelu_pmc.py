import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from tgt_amd import _lib, ops
M, K, N = 262144, 256, 256
g = torch.Generator(device='cuda').manual_seed(0)
a = torch.randn(M, K, device='cuda', generator=g).bfloat16()
w = (torch.randn(N, K, device='cuda', generator=g) * K ** -0.5).bfloat16()
b = torch.randn(N, device='cuda', generator=g).bfloat16()
pre = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
p = float(os.environ.get('P', '0'))
for _ in range(5):
    ops.edge_linear_raw(a, w, b, _lib.EPI_GELU, out=out, out2=pre, dropout=(p, 123 if p else 0))
torch.cuda.synchronize()
