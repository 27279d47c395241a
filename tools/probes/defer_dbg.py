"""which parameter gradients differ between the deferred and the immediate closing sums (debug)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import torch
import golden_util as gu
from tgt_amd import ops
from tgt_amd.pcqm import TGT_Multi
from tgt_amd.training.step import Trainer, StepConfig, preprocess_batch
from tgt_amd.training.synthetic import make_batch
ops._EDGE_MIN_ROWS = 1
ops._SPLIT_MIN_ROWS = 1
kwargs = dict(gu.FULL_AT_CFG, model_height=3)
cfg = StepConfig(num_dist_bins=512, mixed_precision='bf16', coords_noise=0.0, lr_warmup_steps=10, lr_total_steps=100, bucket_mbytes=8)
grads = []
for deferred in (True, False, True):
    ops._DEFER_SUMS = deferred
    m1 = gu.fill_params(TGT_Multi(**kwargs), seed=3).cuda().eval()
    with Trainer(m1, cfg) as tr:
        batch = preprocess_batch(make_batch(3, 7, seed=41, ragged=True), 'cuda', cfg, add_noise=False)
        tr.compute_gradients(batch)
        torch.cuda.synchronize()
        grads.append({n: tr.flat.grad_views[i].clone() for i, (n, p) in enumerate(m1.named_parameters())})
for a, b, tag in ((grads[0], grads[1], 'deferred vs immediate'), (grads[0], grads[2], 'deferred vs deferred')):
    bad = [(n, float((a[n] - b[n]).abs().max()), float(b[n].abs().max())) for n in a if not torch.equal(a[n], b[n])]
    print(tag, len(bad), 'of', len(a), 'differ')
    for x in bad[:40]:
        print('   ', x)
