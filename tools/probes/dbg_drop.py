import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch, numpy as np, os
from tgt_amd import ops
import golden_util as gu
seed=0x1234567890ABCDEF
PD=float(os.environ.get("PD","0.3")); REPS=8
for (B,N,nn,C,H) in [(2,48,[48,37],64,4),(2,32,[32,20],64,4),(2,40,[40,33],256,16),(1,64,[64],32,4),(2,33,[33,20],256,16)]:
    for dtype in (torch.bfloat16, torch.float16, torch.float32):
        L=ops.TripletLayout(C,H)
        rng=np.random.default_rng(0)
        rnd=lambda *s: torch.from_numpy(rng.standard_normal(s))
        d_out=rnd(B,N,N,2*C).to(dtype).cuda()
        m3=gu.additive_mask(nn,N,torch.float32).reshape(B,N,N).cuda()
        fz=rnd(B,N,N,L.width).to(dtype).cuda().requires_grad_(True)
        out=[]; gs=[]
        for rep in range(REPS):
            junk=torch.full((B,N,N,L.width), float('nan'), dtype=dtype, device='cuda'); del junk
            y = ops.triplet_attention(fz, m3, L, dropout=(PD, seed))
            g, = torch.autograd.grad(y, fz, d_out)
            out.append(int(torch.isnan(g).sum())); gs.append(g)
        print((B,N,nn,C,H), dtype, 'nan counts', out, 'deterministic', bool(all(torch.equal(gs[0],x) for x in gs)))
