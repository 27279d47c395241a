"""Which ATen elementwise kernels are left inside the training step, and which line of tgt_amd issues them?
One profiled step (torch.profiler with Python stacks), aggregated by (op, first tgt_amd frame).  python tools/probes/aten_leftovers.py"""
import os
import sys
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tgt_amd.pcqm import TGT_Multi
from tgt_amd.training.configs import tgt_at_24l
from tgt_amd.training.step import Trainer, StepConfig, preprocess_batch
from tgt_amd.training.synthetic import make_batch, batch_seed
from tgt_amd.training.gemm_tuning import enable_gemm_tuning

enable_gemm_tuning(online=True)
dev = torch.device('cuda')
torch.manual_seed(0)
model = TGT_Multi(**tgt_at_24l()).to(dev).train()
cfg = StepConfig(mixed_precision='bf16')
tr = Trainer(model, cfg)
pool = [{k: v.to(dev) for k, v in make_batch(256, 32, batch_seed(s, 0)).items()} for s in range(2)]
gen = torch.Generator(device=dev)
for i in range(6):
    tr.training_step(preprocess_batch(pool[i % 2], dev, cfg, generator=gen))
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    tr.training_step(preprocess_batch(pool[0], dev, cfg, generator=gen))
    torch.cuda.synchronize()
rows = []
for ev in prof.key_averages(group_by_stack_n=12):
    t = getattr(ev, 'self_device_time_total', 0) or 0
    if t <= 0 or not ev.key.startswith('aten::') or any(k in ev.key for k in ('mm', 'matmul', 'linear')):
        continue
    where = '?'
    for fr in ev.stack or ():
        if 'tgt_amd/' in fr:
            where = fr.split('tgt_amd/')[-1]
            break
    rows.append((t, ev.count, ev.key, where))
rows.sort(reverse=True)
print(f'ATen non-GEMM device time in one step: {sum(r[0] for r in rows) / 1e3:.2f} ms in {sum(r[1] for r in rows)} launches')
for t, n, name, where in rows[:45]:
    print(f'{t / 1e3:8.3f} ms  {n:5d} x  {name:28s} {where}')

print('--- by (op, input shapes)')
rows = []
for ev in prof.key_averages(group_by_input_shape=True):
    t = getattr(ev, 'self_device_time_total', 0) or 0
    if t <= 0 or not ev.key.startswith('aten::') or any(k in ev.key for k in ('mm', 'matmul', 'linear')):
        continue
    rows.append((t, ev.count, ev.key, str(ev.input_shapes)[:110]))
rows.sort(reverse=True)
for t, n, name, shp in rows[:40]:
    print(f'{t / 1e3:8.3f} ms  {n:5d} x  {name:22s} {shp}')
