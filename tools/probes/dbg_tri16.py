import sys, torch, numpy as np
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import golden_util as gu
from tgt_amd import ops, _lib
dt = torch.float16 if len(sys.argv) < 2 else getattr(torch, sys.argv[1])
B, N, C, H = 2, 48, 64, 4
L = ops.TripletLayout(C, H)
g = torch.Generator(device='cuda').manual_seed(1)
fused = torch.randn(B, N, N, L.width, device='cuda', generator=g).to(dt)
d_out = torch.randn(B, N, N, 2 * C, device='cuda', generator=g).to(dt)
mask = gu.additive_mask([48, 37], N, torch.float32).reshape(B, N, N).cuda()
out = torch.empty(B, N, N, 2 * C, device='cuda', dtype=dt)
def run(split, cs):
    d_fused = torch.full_like(fused, float('nan'))
    colsum = torch.zeros(B, L.width, device='cuda') if cs else None
    if split:
        qkv = fused[..., :6 * C].contiguous(); eg = fused[..., 6 * C:L.used].contiguous()
        a = ops._tri_args(qkv, mask, out, L, d_out, d_fused, colsum, eg=eg)
    else:
        a = ops._tri_args(fused, mask, out, L, d_out, d_fused, colsum)
    ops._call('bwd', _lib.lib().tgt_triplet_attention_bwd, a)
    torch.cuda.synchronize()
    return d_fused, colsum
ref, _ = run(False, False)
for split, cs in ((False, True), (True, False), (True, True)):
    got, col = run(split, cs)
    d = (got.float() - ref.float())
    bad = ~torch.isfinite(d) | (d.abs() > 1e-2)
    bad[..., L.used:] = False
    print(split, cs, 'bad', int(bad.sum()), 'nan', int((~torch.isfinite(got[..., :L.used])).sum()))
    if bad.any():
        idx = bad.nonzero()
        for r in idx[:6].tolist():
            print('   at', r, 'got', float(got[tuple(r)]), 'ref', float(ref[tuple(r)]))
        print('   max |d|', float(d[bad].abs().max()), 'max |ref|', float(ref.float().abs().max()))
        print(' cols', sorted(set((idx[:, 3] // 16).tolist()))[:40], 'rows b', sorted(set(idx[:, 0].tolist())), 'x', sorted(set(idx[:, 1].tolist()))[:10], 'y', sorted(set(idx[:,2].tolist()))[:10])
    if cs:
        want = ref.float().sum((1, 2))
        print(' colsum err', float((col[:, :L.used] - want[:, :L.used]).abs().max()), float(want.abs().max()))
