"""cProfile of the host side of the training step (where do the ~70 ms of enqueue time per step go?)
python tools/host_profile.py [N lines]"""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
pool = [{k: v.to(dev) for k, v in make_batch(int(os.environ.get('PROBE_B', '256')), 32, batch_seed(s)).items()} for s in range(2)]
def step(i):
    return tr.training_step(preprocess_batch(pool[i % 2], dev, cfg, training=True))
for i in range(12): step(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
# (single-threaded autograd so that the Python backward functions run in THIS thread and show up in the profile)
with torch.autograd.set_multithreading_enabled(False):
    pr.enable()
    for i in range(3): step(i)
    pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 45)
st.sort_stats('cumtime').print_stats(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
