"""Which host ops issue the small DtoD copies / float reductions of one training step (torch.profiler,
grouped by Python stack).  python tools/probes/copy_probe.py"""
import os
import sys
import collections

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity

from tgt_amd.pcqm import TGT_Multi
from tgt_amd.training.configs import tgt_at_24l
from tgt_amd.training.step import Trainer, StepConfig, preprocess_batch
from tgt_amd.training.synthetic import make_batch, batch_seed
from tgt_amd.training.gemm_tuning import enable_gemm_tuning

enable_gemm_tuning(online=True)
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = TGT_Multi(**tgt_at_24l()).to(dev).train()
cfg = StepConfig(mixed_precision='bf16')
tr = Trainer(model, cfg)
pool = [{k: v.to(dev) for k, v in make_batch(256, 32, batch_seed(s, 0)).items()} for s in range(2)]
gen = torch.Generator(device=dev)
gen.manual_seed(1)
for i in range(3):
    tr.training_step(preprocess_batch(pool[i % 2], dev, cfg, training=True, generator=gen))
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    tr.training_step(preprocess_batch(pool[1], dev, cfg, training=True, generator=gen))
    torch.cuda.synchronize()
want = ('aten::copy_', 'aten::sum', 'aten::clone', 'aten::contiguous', 'aten::mul', 'aten::add', 'aten::add_', 'aten::to', 'aten::_to_copy')
groups = collections.Counter()
times = collections.Counter()
for e in prof.events():
    if e.name in want and e.device_time_total > 0:
        st = [s for s in (e.stack or []) if '/root/repo' in s or 'tgt_amd' in s]
        key = (e.name, str(e.input_shapes)[:70], ' <- '.join(s.split('/')[-1][:60] for s in st[:3]))
        groups[key] += 1
        times[key] += e.device_time_total
for k, n in sorted(groups.items(), key=lambda kv: -times[kv[0]])[:40]:
    print(f'{n:5d} x {times[k] / n:7.1f}us = {times[k] / 1e3:6.2f}ms  {k[0]:14s} {k[1]:70s} {k[2]}')
