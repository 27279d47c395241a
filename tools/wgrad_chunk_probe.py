"""dW = sum_p dY_p^T X_p as a batched GEMM (fp32 partials) + sum: time vs the number of chunks P
for the edge-sized linears, with TunableOp picking the GEMM.  python tools/wgrad_chunk_probe.py"""
import torch
t = torch.cuda.tunable
t.enable(True); t.tuning_enable(True); t.set_max_tuning_duration(60); t.set_max_tuning_iterations(20)
t.set_filename('/tmp/wgrad_chunk_tunable.csv', insert_device_ordinal=False)
M = 262144


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


for (N, K) in [(1600, 256), (256, 256), (256, 512), (128, 256), (256, 64)]:
    dY = torch.randn(M, N, device='cuda', dtype=torch.bfloat16)
    X = torch.randn(M, K, device='cuda', dtype=torch.bfloat16)
    res = [f'out={N} in={K} ideal@5TB/s {(M * (N + K) * 2) / 5e12 * 1e6:.0f}us']
    for P in (8, 16, 32, 64, 128):
        part = [None]

        def mm():
            part[0] = torch.bmm(dY.view(P, M // P, N).transpose(1, 2), X.view(P, M // P, K), out_dtype=torch.float32)
        tm = timeit(mm)
        ts = timeit(lambda: part[0].sum(0))
        res.append(f'P{P}: {tm:.0f}+{ts:.0f}={tm + ts:.0f}')
    print(' | '.join(res), flush=True)
