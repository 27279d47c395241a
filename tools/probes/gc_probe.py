"""are the occasional 30-40 ms step stalls Python garbage collections? (host enqueue runs ~20 ms ahead of the GPU: a long collection
starves the stream)"""
import gc, os, sys, time
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
events, t0 = [], [0.0]
def cb(phase, info):
    if phase == 'start':
        t0[0] = time.perf_counter()
    else:
        events.append((cur[0], info['generation'], round((time.perf_counter() - t0[0]) * 1e3, 2), info['collected']))
gc.callbacks.append(cb)
cur = [0]
def run(n, label):
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    host = []
    marks[0].record()
    for i in range(n):
        cur[0] = i
        h0 = time.perf_counter()
        tr.training_step(preprocess_batch(pool[i % 4], dev, cfg, training=True, generator=gen))
        host.append((time.perf_counter() - h0) * 1e3)
        marks[i + 1].record()
    torch.cuda.synchronize()
    ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(n)]
    slow = [(i, round(ms[i], 1), round(host[i], 1)) for i in range(n) if ms[i] > sorted(ms)[n // 2] + 5]
    print(label, 'median', round(sorted(ms)[n // 2], 2), 'mean', round(sum(ms) / n, 2), 'host median', round(sorted(host)[n // 2], 1), 'slow steps (i, gpu ms, host ms)', slow)
    print('   gc events (step, generation, ms, collected):', [e for e in events if e[2] > 2.0])
    events.clear()
run(60, 'settle')
run(60, 'automatic gc')
gc.collect(); gc.freeze(); gc.disable()
run(60, 'gc frozen + disabled')
