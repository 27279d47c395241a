import torch, time, sys
torch.manual_seed(0)
dev='cuda'
def timeit(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/it
M=262144
for (N,K) in [(256,256),(1600,256),(512,256),(256,512),(128,256),(256,64),(256,128)]:
    dY=torch.randn(M,N,device=dev,dtype=torch.bfloat16); X=torch.randn(M,K,device=dev,dtype=torch.bfloat16)
    ref=timeit(lambda: dY.t() @ X)
    res=[f'N={N} K={K} plain {ref*1e3:.0f}us']
    for P in (32,64,128,256):
        def f():
            part=torch.bmm(dY.view(P,M//P,N).transpose(1,2), X.view(P,M//P,K))
            return part.sum(0)
        t=timeit(f)
        res.append(f'P{P}: {t*1e3:.0f}')
    try:
        def g():
            part=torch.bmm(dY.view(64,M//64,N).transpose(1,2), X.view(64,M//64,K), out_dtype=torch.float32)
            return part.sum(0)
        t=timeit(g); res.append(f'P64f32: {t*1e3:.0f}')
    except Exception as ex:
        res.append('f32out-unsupported:'+str(ex)[:40])
    ideal=(M*(N+K)*2)/5e12*1e6
    res.append(f'ideal@5TB/s {ideal:.0f}us')
    print(' | '.join(res), flush=True)
    a=(dY.t()@X).float(); b=torch.bmm(dY.view(64,M//64,N).transpose(1,2), X.view(64,M//64,K)).float().sum(0)
    print('   rel diff', float((a-b).norm()/a.norm()))
