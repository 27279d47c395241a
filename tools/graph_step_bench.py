#!/usr/bin/env python
"""Eager Trainer.training_step vs its hipGraph replay (tgt_amd/training/graphed.py), TGT-At 24L, synthetic batches, dropouts on:

    python tools/graph_step_bench.py --batch 8 --nodes 32 [--steps 30]

prints one JSON line per mode.  The replay pays off where the host is the bottleneck (small batches / few nodes); at the
BASELINE batch the GPU is (bench.py stays eager)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--nodes', type=int, default=32)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--precision', default='bf16')
    a = ap.parse_args()
    from tgt_amd.pcqm import TGT_Multi
    from tgt_amd.training.configs import tgt_at_24l
    from tgt_amd.training.step import Trainer, StepConfig, preprocess_batch
    from tgt_amd.training.graphed import GraphedTrainingStep
    from tgt_amd.training.synthetic import make_batch, batch_seed
    from tgt_amd.training.gemm_tuning import enable_gemm_tuning
    from tgt_amd.training.affinity import bind_to_gpu_numa
    enable_gemm_tuning(online=True)
    dev = torch.device('cuda', 0)
    bind_to_gpu_numa(0)
    cfg = StepConfig(mixed_precision=None if a.precision == 'fp32' else a.precision)
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    pool = [preprocess_batch({k: v.to(dev) for k, v in make_batch(a.batch, a.nodes, batch_seed(s, 0)).items()}, dev, cfg,
                             training=True, generator=gen) for s in range(4)]

    def run(graphed):
        torch.manual_seed(0)
        model = TGT_Multi(**tgt_at_24l()).to(dev).train()
        with Trainer(model, cfg) as tr:
            if graphed:
                gs = GraphedTrainingStep(tr, pool[0], warmup=6)
                step = lambda i: gs.step(pool[i % 4])
            else:
                gs = None
                step = lambda i: tr.training_step(pool[i % 4])
                for i in range(8):
                    step(i)
            for i in range(4):
                step(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(a.steps):
                out = step(i)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / a.steps
            loss = float(out[1])
            if gs is not None:
                gs.close()
        print(json.dumps({'mode': 'hipGraph replay' if graphed else 'eager', 'batch': a.batch, 'nodes': a.nodes,
                          'ms_per_step': round(dt * 1e3, 3), 'graphs_per_s': round(a.batch / dt, 1), 'loss': round(loss, 5),
                          'precision': a.precision, 'steps': a.steps}), flush=True)
        del model
        torch.cuda.empty_cache()

    run(False)
    run(True)


if __name__ == '__main__':
    main()
