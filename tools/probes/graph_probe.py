#!/usr/bin/env python
"""Would a captured (hipGraph) step remove the dispatch gaps?  The TGT-At 24L forward at the BASELINE batch (256 graphs, N=32,
bf16, train mode with the dropout pattern frozen for the capture) eager against its hipGraph replay -- same kernels, same
stream order, ~900 launches.  Prints one JSON line."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tgt_amd import ops  # noqa: E402,F401
from tgt_amd.pcqm import TGT_Multi  # noqa: E402
from tgt_amd.pcqm.graphed import GraphedForward  # noqa: E402
from tgt_amd.training.configs import tgt_at_24l  # noqa: E402
from tgt_amd.training.gemm_tuning import enable_gemm_tuning  # noqa: E402
from tgt_amd.training.step import StepConfig, preprocess_batch  # noqa: E402
from tgt_amd.training.synthetic import make_batch, batch_seed  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    enable_gemm_tuning(online=True)
    torch.manual_seed(0)
    model = TGT_Multi(**tgt_at_24l()).to(dev).train()
    cfg = StepConfig(mixed_precision='bf16')
    raw = {k: v.to(dev) for k, v in make_batch(256, 32, batch_seed(0, 0)).items()}
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    batch = preprocess_batch(raw, dev, cfg, training=True, generator=gen)
    batch = {k: v for k, v in batch.items() if torch.is_tensor(v)}

    def eager():
        with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
            return model(batch)

    def timeit(fn, iters=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3

    t_eager = timeit(eager)
    g = GraphedForward(model, batch, autocast_dtype=torch.bfloat16, allow_frozen_dropout=True)
    t_graph = timeit(lambda: g.graph.replay())
    # host cost of enqueueing the eager forward (no device wait)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eager()
    t_host = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    print(json.dumps(dict(forward_eager_ms=round(t_eager, 3), forward_hipgraph_replay_ms=round(t_graph, 3),
                          eager_host_enqueue_ms=round(t_host, 3), batch=256, nodes=32, dtype='bf16')))


if __name__ == '__main__':
    main()
