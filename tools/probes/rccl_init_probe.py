"""does this torch take ProcessGroupNCCL.Options(is_high_priority_stream=True) the way bench.py passes it? (1 rank, 1 GPU)"""
import os
import torch
import torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
dist.init_process_group(backend='nccl', rank=0, world_size=1, device_id=dev, pg_options=opts)
t = torch.ones(1 << 20, device=dev)
h = dist.all_reduce(t, async_op=True); h.wait(); torch.cuda.synchronize()
g = [torch.zeros(2, dtype=torch.float64, device=dev)]
dist.all_gather(g, torch.tensor([1.0, 2.0], dtype=torch.float64, device=dev))
print('ok', float(t.sum()), g[0].tolist(), dist.get_backend())
dist.destroy_process_group()
