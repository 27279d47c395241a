"""bucketed (hook-driven) gradient path with the node side stream: does autograd still report an AccumulateGrad stream mismatch?"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch, torch.distributed as dist
import golden_util as gu
from tgt_amd.pcqm import TGT_Multi
from tgt_amd.training.step import Trainer, StepConfig, preprocess_batch
from tgt_amd.training.synthetic import make_batch
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29577')
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
cfg = StepConfig(num_dist_bins=512, mixed_precision='bf16', coords_noise=0.0, bucket_mbytes=8)
m = gu.fill_params(TGT_Multi(**dict(gu.FULL_AT_CFG, model_height=4)), seed=3).cuda().train()
tr = Trainer(m, cfg, force_distributed=True)
b = preprocess_batch(make_batch(32, 32, seed=4), 'cuda', cfg, add_noise=False)
for i in range(4):
    out, loss = tr.training_step(b)
torch.cuda.synchronize()
print('buckets', len(tr.buckets), 'order', tr.bucket_order[:6], 'comm_exposed_ms', [round(v, 3) for v in tr.comm_exposed_ms()], 'loss', float(loss))
dist.destroy_process_group()
