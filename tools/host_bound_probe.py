import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from tgt_amd.pcqm import TGT_Multi
from tgt_amd.training.configs import tgt_at_24l
from tgt_amd.training.step import Trainer, StepConfig, preprocess_batch
from tgt_amd.training.synthetic import make_batch, batch_seed
from tgt_amd.training.gemm_tuning import enable_gemm_tuning
enable_gemm_tuning(online=True)
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = TGT_Multi(**tgt_at_24l()).to(dev).train()
cfg = StepConfig()
tr = Trainer(model, cfg)
pool = [{k: v.to(dev) for k, v in make_batch(256, 32, batch_seed(s)).items()} for s in range(2)]
def step(i):
    return tr.training_step(preprocess_batch(pool[i % 2], dev, cfg, training=True))
for i in range(3): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter(); host = []
for i in range(5):
    h0 = time.perf_counter(); step(i); host.append(time.perf_counter() - h0)
torch.cuda.synchronize()
print('wall ms/step', (time.perf_counter() - t0) / 5 * 1e3, 'host enqueue ms/step', [round(h * 1e3, 1) for h in host])
