"""on which stream does autograd accumulate each parameter's gradient, and on which was it produced? (debug for the
'AccumulateGrad stream mismatch' warning)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
import golden_util as gu
from tgt_amd import ops
from tgt_amd.pcqm import TGT_Multi
from tgt_amd.training.step import Trainer, StepConfig, preprocess_batch
from tgt_amd.training.synthetic import make_batch
kwargs = dict(gu.FULL_AT_CFG, model_height=3)
cfg = StepConfig(num_dist_bins=512, mixed_precision='bf16', coords_noise=0.0)
m = gu.fill_params(TGT_Multi(**kwargs), seed=3).cuda().train()
tr = Trainer(m, cfg)
batch = preprocess_batch(make_batch(64, 32, seed=41), 'cuda', cfg, add_noise=False)
tr.training_step(batch)
acc_stream, names = {}, {id(p): n for n, p in m.named_parameters()}
main = torch.cuda.current_stream().cuda_stream
handles = []
for n, p in m.named_parameters():
    node = p.view_as(p).grad_fn.next_functions[0][0]
    def pre(grad_inputs, n=n):
        acc_stream[n] = torch.cuda.current_stream().cuda_stream
    handles.append((node, node.register_prehook(pre)))
tr.training_step(batch)
torch.cuda.synchronize()
side = {k: v.cuda_stream for k, v in ops._side_streams.items()}
print('main', main, 'side', side)
on_side = sorted(n for n, s in acc_stream.items() if s != main)
print(len(on_side), 'parameters accumulate on a non-main stream, e.g.', on_side[:12])
on_main_node = sorted(n for n, s in acc_stream.items() if s == main and ('node_ffn' in n or 'lin_O_h' in n or 'mha_ln_h' in n or 'lin_QKV.' in n))
print(len(on_main_node), 'node-channel parameters accumulate on main, e.g.', on_main_node[:12])
