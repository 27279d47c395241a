"""which allocations make the caching allocator call hipMalloc in steady state (debug for device_allocs_in_timed_region)"""
import os, sys, collections
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
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
pool = [{k: v.to(dev) for k, v in make_batch(256, 32, batch_seed(s, 0)).items()} for s in range(4)]
gen = torch.Generator(device=dev); gen.manual_seed(1)
def step(i):
    tr.training_step(preprocess_batch(pool[i % 4], dev, cfg, training=True, generator=gen))
for i in range(8):
    step(i)
torch.cuda.synchronize()
torch.cuda.memory._record_memory_history(max_entries=200000)
s0 = torch.cuda.memory_stats()
for i in range(8, 20):
    step(i)
torch.cuda.synchronize()
s1 = torch.cuda.memory_stats()
snap = torch.cuda.memory._snapshot()
print('device allocs in 12 steps:', s1['num_device_alloc'] - s0['num_device_alloc'], 'reserved GB', s1['reserved_bytes.all.current'] / 1e9,
      'inactive split GB', s1.get('inactive_split_bytes.all.current', 0) / 1e9)
ev = [e for tr_ in snap['device_traces'] for e in tr_]
segs = [e for e in ev if e['action'] == 'segment_alloc']
print(len(segs), 'segment_alloc events; sizes MB:', collections.Counter(round(e['size'] / 2**20) for e in segs).most_common(10))
for e in segs[:6]:
    fr = [f"{f['filename'].split('/')[-1]}:{f['line']}:{f['name']}" for f in e.get('frames', []) if 'tgt_amd' in f['filename'] or 'bench' in f['filename']][:6]
    print(round(e['size'] / 2**20), 'MB stream', e.get('stream'), fr)
