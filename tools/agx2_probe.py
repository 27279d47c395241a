"""Training-step time of the TGT-Agx2 configuration (12 shared layers x2, triplet aggregate), B=256 N=32 bf16:
a secondary configuration (BASELINE configs[0]/[4] use it for CPU / two-stage inference).  python tools/agx2_probe.py"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgt_amd import ops
from tgt_amd.pcqm import TGT_Multi
from tgt_amd.training import configs, gemm_tuning
from tgt_amd.training.step import Trainer, StepConfig, preprocess_batch
from tgt_amd.training.synthetic import make_batch

gemm_tuning.enable_gemm_tuning(online=True)
cfg = StepConfig(num_dist_bins=512, mixed_precision='bf16')
torch.manual_seed(0)
model = TGT_Multi(**configs.tgt_agx2_12x2(num_dist_bins=512)).cuda().train()
tr = Trainer(model, cfg)
pool = [{k: v.cuda() for k, v in make_batch(256, 32, seed=s).items()} for s in range(2)]
gen = torch.Generator(device='cuda'); gen.manual_seed(1)
for i in range(4):
    tr.training_step(preprocess_batch(pool[i % 2], 'cuda', cfg, training=True, generator=gen))
prof = ops.profile_kernels(True)
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 8
for i in range(K):
    _, loss = tr.training_step(preprocess_batch(pool[i % 2], 'cuda', cfg, training=True, generator=gen))
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
ops.profile_kernels(False)
t = {k: round(sum(v) / len(v), 4) for k, v in ops.kernel_times_ms(prof).items()}
print(f'agx2 step {dt*1e3:.1f} ms  {256/dt:.0f} graphs/s  loss {float(loss):.4f}', t)
