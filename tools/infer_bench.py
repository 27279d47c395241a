#!/usr/bin/env python
"""Secondary benchmark lines: forward-only graphs/s of the two inference configurations of BASELINE.json
on one MI355X.  One JSON line per configuration.

  cfg1  TGT-Agx2 12x2 distance predictor (256 bins), PCQM4Mv2-like val mini-batch of 8 ragged graphs (N <= 32), eval
        forward -- the reference's CPU-runnable case, here on the GPU in fp32 (and bf16 autocast)
  cfg5  TGT-Agx2 two-stage end-to-end inference, fp16 autocast, batch 512, N = 32: S stochastic forwards (dropout ON,
        predict_in_train) of TGT_Distance -> argmax bins of the symmetrised softmax -> bins2dist on the device ->
        S stochastic forwards of TGT_Gap, one per bins sample (tgt_amd/pcqm/predict.py::two_stage_predict)

python tools/infer_bench.py [--samples 4] [--batch 512] [--steps 3]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgt_amd import ops                                              # noqa: E402
from tgt_amd.pcqm import TGT_Distance, TGT_Gap, predict as pp        # noqa: E402
from tgt_amd.training import configs, gemm_tuning                   # noqa: E402
from tgt_amd.training.synthetic import make_batch                    # noqa: E402


def device_batch(graphs, nodes, seed, ragged):
    b = {k: v.cuda() for k, v in make_batch(graphs, nodes, seed, ragged=ragged).items()}
    nm = b['node_mask']
    b['edge_mask'] = nm.unsqueeze(-1) * nm.unsqueeze(-2)
    c = b['dft_coords']
    b['dist_input'] = torch.norm(c.unsqueeze(-2) - c.unsqueeze(-3), dim=-1)
    return b


def timed(fn, warmup, steps):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--samples', type=int, default=4, help='MC samples per stage of cfg5 (the shipped configs use 50)')
    ap.add_argument('--batch', type=int, default=512)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--only', default='')
    a = ap.parse_args()
    gemm_tuning.enable_gemm_tuning(online=True)
    torch.manual_seed(0)

    if not a.only or 'cfg1' in a.only:
        kw = configs.tgt_agx2_12x2(num_dist_bins=256, embed_3d_type='none')       # coords_input: none (tgt_agx2_dp_nordkit.yaml)
        model = TGT_Distance(**kw).cuda().eval()
        batch = device_batch(8, 32, 11, ragged=True)
        for prec in ('fp32', 'bf16'):
            ctx = torch.autocast('cuda', dtype=torch.bfloat16) if prec == 'bf16' else torch.autocast('cuda', enabled=False)

            def fwd():
                with torch.no_grad(), ctx:
                    model(batch)
            dt = timed(fwd, 3, 20)
            line = dict(metric='graphs/sec forward, TGT-Agx2 12x2 dist-predictor, 8-graph ragged val mini-batch (cfg 1)',
                        value=round(8 / dt, 1), unit='graphs/s', ms_per_forward=round(dt * 1e3, 3), dtype=prec, n_gpus=1,
                        data='synthetic', launch='eager')
            print(json.dumps(line), flush=True)
            from tgt_amd.pcqm.graphed import GraphedForward
            gf = GraphedForward(model, batch, autocast_dtype=torch.bfloat16 if prec == 'bf16' else None)
            with torch.no_grad(), ctx:
                want = model(batch)
            same = bool(torch.equal(gf(batch), want))
            dt = timed(lambda: gf(batch), 3, 50)
            line.update(value=round(8 / dt, 1), ms_per_forward=round(dt * 1e3, 3), launch='hipGraph replay', bit_identical_to_eager=same)
            print(json.dumps(line), flush=True)
        del model

    if not a.only or 'cfg5' in a.only:
        S, B = a.samples, a.batch
        dk = configs.tgt_agx2_12x2(num_dist_bins=256, embed_3d_type='none')
        gk = configs.tgt_agx2_12x2(num_dist_bins=256, embed_3d_type='gaussian')
        gk.pop('num_dist_bins')
        dist_model = TGT_Distance(**dk).cuda().train()                                # predict_in_train: dropout ON
        gap_model = TGT_Gap(**gk).cuda().train()
        batch = device_batch(B, 32, 12, ragged=False)
        batch['num_nodes'] = batch['node_mask'].sum(-1).long()
        # (events only around the attention kernels, one launch in three: an event pair around every launch slows the host AND the queue,
        #  DESIGN.md 5.2)
        prof = ops.profile_kernels(True, only=('tgt_triplet_aggregate_fwd', 'tgt_node_attention_fwd', 'tgt_node_attention_fwd(logits)'), stride=3)

        def run():
            return pp.two_stage_predict(dist_model, gap_model, batch, S, 256, 8, autocast_dtype=torch.float16)
        dt = timed(run, 1, a.steps)
        ops.profile_kernels(False)
        bins, gap = run()
        kt = {k: round(sum(v) / len(v), 4) for k, v in ops.kernel_times_ms(prof).items()}
        print(json.dumps(dict(metric='graphs/sec end-to-end two-stage inference, TGT-Agx2 12x2 (dist+gap), fp16, batch 512 (cfg 5)',
                              value=round(B / dt, 1), unit='graphs/s', ms_per_batch=round(dt * 1e3, 2), dtype='fp16',
                              mc_samples_per_stage=S, stochastic_forwards_per_batch=2 * S,
                              graph_forwards_per_s=round(2 * S * B / dt, 1), n_gpus=1, data='synthetic',
                              gap_finite=bool(torch.isfinite(gap).all()), kernel_ms=kt)), flush=True)


if __name__ == '__main__':
    main()
