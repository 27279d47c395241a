"""Which library GEMMs one training step of the bench workload issues, and what each costs against the bytes it has
to move: torch.mm / addmm / bmm / matmul are wrapped, every call is bracketed by device synchronisation and HIP events
(so the numbers are per-call times in the step's own memory context, not overlapped), and the calls are grouped by
(op, operand shapes, strides' transposition pattern).
    python tools/gemm_shapes.py [--steps 2]  ->  one table (stdout)"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

LOG = collections.OrderedDict()
ON = [False]


def _lay(t):
    if t.dim() < 2:
        return 'v'
    return 'n' if t.stride(-1) == 1 else ('t' if t.stride(-2) == 1 else 's')


def _wrap(name, fn, nmat):
    def f(*a, **k):
        if not ON[0] or not (torch.is_tensor(a[0]) and a[0].is_cuda):
            return fn(*a, **k)
        mats = [x for x in a if torch.is_tensor(x)]
        key = (name, tuple((tuple(x.shape), _lay(x), str(x.dtype).replace('torch.', '')) for x in mats),
               str(k.get('out_dtype', '')).replace('torch.', ''))
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = fn(*a, **k)
        e.record()
        torch.cuda.synchronize()
        nbytes = sum(x.numel() * x.element_size() for x in mats) + (out.numel() * out.element_size() if 'out' not in k else
                                                                    k['out'].numel() * k['out'].element_size())
        rec = LOG.setdefault(key, [0, 0.0, nbytes])
        rec[0] += 1
        rec[1] += s.elapsed_time(e) * 1e3
        return out
    return f


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--nodes', type=int, default=32)
    args = ap.parse_args()
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
    pool = [{k: v.to(dev) for k, v in make_batch(args.batch, args.nodes, batch_seed(s, 0)).items()} for s in range(2)]
    for name, nmat in (('mm', 2), ('addmm', 3), ('bmm', 2), ('matmul', 2)):
        setattr(torch, name, _wrap(name, getattr(torch, name), nmat))
    torch.Tensor.__matmul__ = _wrap('matmul', torch.Tensor.__matmul__, 2)
    torch.nn.functional.linear = _wrap('F.linear', torch.nn.functional.linear, 3)
    for i in range(3):
        tr.training_step(preprocess_batch(pool[i % 2], dev, cfg))
    ON[0] = True
    for i in range(args.steps):
        tr.training_step(preprocess_batch(pool[i % 2], dev, cfg))
    ON[0] = False
    tot = 0.0
    print(f'{"calls/step":>10} {"us/call":>8} {"ms/step":>8} {"MB":>7} {"TB/s":>5}  op operands')
    for key, (n, us, nb) in sorted(LOG.items(), key=lambda kv: -kv[1][1]):
        per = us / n
        tot += us / args.steps
        ops_ = ' '.join(f'{list(s)}{l}:{d}' for s, l, d in key[1])
        print(f'{n / args.steps:10.1f} {per:8.1f} {us / args.steps / 1e3:8.2f} {nb / 1e6:7.0f} {nb / per / 1e6:5.2f}  {key[0]} {ops_} {key[2]}')
    print(f'total {tot / 1e3:.1f} ms/step of library GEMM calls (synchronous timing)')


if __name__ == '__main__':
    main()
