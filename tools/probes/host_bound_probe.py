"""is the step ever waiting for the HOST?  A marker event is recorded on the main stream at every layer boundary (forward pre-hook,
backward pre-hook); when the host arrives at the next boundary it asks whether the previous marker has already completed
(`query()`): True = the GPU had drained everything the host had queued up to that marker, i.e. it is (about to be) idle.
Prints, per phase, how many of the boundaries found the GPU caught up, the host's enqueue time per step and the GPU step time."""
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
if os.environ.get('PROBE_BIND') == '1':
    from tgt_amd.training.affinity import bind_to_gpu_numa
    print('affinity', bind_to_gpu_numa(0))
torch.manual_seed(0)
model = TGT_Multi(**tgt_at_24l()).to(dev).train()
cfg = StepConfig(mixed_precision='bf16')
tr = Trainer(model, cfg)
pool = [{k: v.to(dev) for k, v in make_batch(int(os.environ.get('PROBE_B', '256')), 32, batch_seed(s, 0)).items()} for s in range(4)]
gen = torch.Generator(device=dev); gen.manual_seed(1)

layers = [m for m in model.modules() if type(m).__name__ == 'TGT_Layer']
caught = {'fwd': [0, 0], 'bwd': [0, 0]}
last = [None]
lead = {'fwd': [], 'bwd': []}
on = [False]
nreg = [0]


def mark(phase):
    def hook(*_):
        if not on[0]:
            return None
        prev = last[0]
        if prev is not None:
            caught[phase][1] += 1
            if prev.query():
                caught[phase][0] += 1
        ev = torch.cuda.Event()
        ev.record()
        last[0] = ev
        return None
    return hook


for m in layers:
    m.register_forward_pre_hook(mark('fwd'))
    def fwd_hook(mod, args, out, _bw=mark('bwd')):      # the layer returns a Graph: hang the backward marker on its edge tensor
        if on[0]:
            for key in ('h', 'e'):
                t = out.get(key) if isinstance(out, dict) else None
                if torch.is_tensor(t) and t.requires_grad:
                    t.register_hook(lambda g: _bw())
                    nreg[0] += 1
                    break
        return None
    m.register_forward_hook(fwd_hook)


def run(n):
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    host = []
    cpu0, thr0 = time.process_time(), time.thread_time()
    marks[0].record()
    for i in range(n):
        h0 = time.perf_counter()
        tr.training_step(preprocess_batch(pool[i % 4], dev, cfg, training=True, generator=gen))
        host.append((time.perf_counter() - h0) * 1e3)
        marks[i + 1].record()
    cpu, thr = (time.process_time() - cpu0) * 1e3 / n, (time.thread_time() - thr0) * 1e3 / n
    torch.cuda.synchronize()
    print(f'   CPU time per step: process {cpu:.1f} ms (all threads), main thread {thr:.1f} ms')
    ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(n))
    return ms[n // 2], sorted(host)[n // 2]


run(50)
gc.collect(); gc.freeze()
on[0] = True
g, h = run(30)
print('backward markers registered', nreg[0])
print(f'with markers: GPU step median {g:.2f} ms, host enqueue median {h:.2f} ms')
for k, (c, n) in caught.items():
    print(f'  {k}: GPU had caught up with the host at {c} of {n} layer boundaries ({100.0 * c / max(1, n):.1f} %)')
on[0] = False
g, h = run(30)
print(f'without markers: GPU step median {g:.2f} ms, host enqueue median {h:.2f} ms')
