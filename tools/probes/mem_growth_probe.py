"""does the caching allocator's reserve keep growing with the steps? (the 0-2 device allocations per step on the node side stream)"""
import os, sys
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
last = 0
for i in range(401):
    tr.training_step(preprocess_batch(pool[i % 4], dev, cfg, training=True, generator=gen))
    if i % 50 == 0:
        torch.cuda.synchronize()
        s = torch.cuda.memory_stats()
        print(i, 'reserved GB', round(s['reserved_bytes.all.current'] / 1e9, 2), 'device allocs', s['num_device_alloc'], '(+%d)' % (s['num_device_alloc'] - last),
              'inactive split GB', round(s['inactive_split_bytes.all.current'] / 1e9, 2), flush=True)
        last = s['num_device_alloc']
