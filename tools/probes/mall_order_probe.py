"""Does the order in which a consumer walks a tensor larger than the 256 MB Infinity Cache matter?  The producer writes S bytes in
ascending order; the consumer reads them in chunks, ascending (starts at the bytes written FIRST: evicted long ago) or descending
(starts at the bytes written LAST: possibly still cached)."""
import json, torch
dev = 'cuda'
def run(mb, chunks=16):
    n = mb * 1024 * 1024 // 2
    src = torch.randn(n, device=dev, dtype=torch.bfloat16)
    a = torch.empty_like(src)
    outs = torch.empty(chunks, device=dev, dtype=torch.float32)
    cs = n // chunks
    res = {}
    for order in ('ascending', 'descending', 'ascending', 'descending'):
        idx = list(range(chunks)) if order == 'ascending' else list(reversed(range(chunks)))
        ts = []
        for it in range(6):
            a.copy_(src)                          # producer (ascending)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for c in idx:
                outs[c] = a[c * cs:(c + 1) * cs].sum(dtype=torch.float32)
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e))
        res.setdefault(order, []).append(round(min(ts[1:]), 4))
    return res
out = {mb: run(mb) for mb in (128, 256, 512, 805, 1600)}
print(json.dumps(out))
