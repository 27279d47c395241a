"""Forward-only and forward+backward time of the BASELINE model with the node side stream on/off
(same process, HIP events).  python tools/stream_probe.py"""
import os
import sys
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
model = TGT_Multi(**configs.tgt_at_24l()).cuda()
tr = Trainer(model, cfg)
model.train()
batch = preprocess_batch(make_batch(256, 32, seed=1234), 'cuda', cfg, training=True)


def timeit(fn, it=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it


def fwd():
    with torch.no_grad(), tr.autocast():
        model(batch)


def fwdbwd():
    tr.compute_gradients(batch)


for rep in range(2):
    for en in (False, True):
        ops.side_stream.enabled = en
        print(f'side_stream={en}: fwd {timeit(fwd):.2f} ms   fwd+bwd {timeit(fwdbwd):.2f} ms', flush=True)
