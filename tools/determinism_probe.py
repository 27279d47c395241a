"""Bitwise run-to-run check of the BASELINE-size training step (dropouts ON, fixed seeds):
python tools/determinism_probe.py  ->  prints per-run loss and a hash of the flat gradient."""
import hashlib
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgt_amd.pcqm import TGT_Multi
from tgt_amd.training import configs, gemm_tuning
from tgt_amd.training.step import Trainer, preprocess_batch
from tgt_amd.training.synthetic import make_batch

gemm_tuning.enable_gemm_tuning(online=True)
from tgt_amd.training.step import StepConfig
mk, cfg = configs.tgt_at_24l(), StepConfig(num_dist_bins=512, mixed_precision='bf16')
layers = int(os.environ.get('LAYERS', 24))
mk = dict(mk, model_height=layers)


def run():
    torch.manual_seed(0)
    from tgt_amd import ops
    ops.reset_random_pools()
    model = TGT_Multi(**mk).cuda()
    tr = Trainer(model, cfg)
    model.train()
    out = []
    for step in range(int(os.environ.get('STEPS', 2))):
        torch.manual_seed(100 + step)
        batch = preprocess_batch(make_batch(256, 32, seed=1234 + step), 'cuda', cfg, training=True)
        loss = tr.compute_gradients(batch)[1]
        torch.cuda.synchronize()
        h = hashlib.sha1(tr.flat.grad.cpu().numpy().tobytes()).hexdigest()[:12]
        out.append((float(loss), h))
        tr.global_step += 1
        tr.apply_gradients()
    return out


for i in range(3):
    print(os.environ.get('TGT_NODE_STREAM', '1'), run()[-2:], flush=True)
