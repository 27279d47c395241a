"""What stands between the training step and a hipGraph capture?  Warm up eagerly, then try to capture ONE Trainer.training_step
(static batch; the dropout seeds of the capture are baked in, so this is a feasibility / timing probe, not a training mode) and
replay it.  Prints the first error of the capture, or eager vs replay time per step at a small and at the BASELINE batch."""
import os, sys, time, traceback
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
B = int(os.environ.get('PROBE_B', '8'))
torch.manual_seed(0)
model = TGT_Multi(**tgt_at_24l()).to(dev).train()
cfg = StepConfig(mixed_precision='bf16')
tr = Trainer(model, cfg)
raw = {k: v.to(dev) for k, v in make_batch(B, 32, batch_seed(0, 0)).items()}
gen = torch.Generator(device=dev); gen.manual_seed(1)
batch = preprocess_batch(raw, dev, cfg, training=True, generator=gen)


def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(6):
    tr.training_step(batch)
print(f'eager: {timed(lambda: tr.training_step(batch), 10):.2f} ms per step at B = {B}')
g = torch.cuda.CUDAGraph()
try:
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        tr.training_step(batch)                      # (side streams / pools created on this stream before the capture)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g, stream=s):
        tr.training_step(batch)
    print(f'captured; replay: {timed(g.replay, 10):.2f} ms per step')
except Exception as e:
    print('capture failed:', type(e).__name__, str(e)[:600])
    tb = traceback.extract_tb(e.__traceback__)
    for fr in tb[-6:]:
        print('   ', fr.filename.replace(R + '/', ''), fr.lineno, fr.name, '|', (fr.line or '')[:120])
