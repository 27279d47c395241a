"""hipGraph replay of an inference forward.

A small-batch forward of a 24-layer TGT model is launch-bound: ~2400 kernels of a few microseconds each behind a Python
dispatch of 5-10 us per launch (BASELINE config 1: 8 graphs).  `GraphedForward` captures ONE forward of a task model
(eval mode or train mode -- the dropout kernels draw their seeds on the host at capture time, so a replayed train-mode
forward repeats the captured dropout pattern: use it for eval-mode forwards) into a HIP graph through
torch.cuda.CUDAGraph and replays it on new inputs of the same shapes: one graph launch instead of thousands of kernel
launches.  The kernels are the same libtgt_hip.so kernels on the same stream order; outputs are bit-identical to the eager
forward (tests/test_hip_predict.py).  The reference has no counterpart (its forward is eager PyTorch); this is the MI355X
form of its `prediction_loop` inner call for small batches (lib/training/training.py:700-722).
"""
import torch


def _stochastic(model):
    """does any sub-module carry a positive dropout / drop-path rate?"""
    for m in model.modules():
        for name in ('p', 'drop_prob', 'drop_path', 'source_dropout', 'act_dropout', 'attention_dropout', 'triplet_dropout'):
            v = getattr(m, name, 0)
            if isinstance(v, (int, float)) and not isinstance(v, bool) and v > 0:
                return True
    return False


class GraphedForward:
    def __init__(self, model, example_batch, autocast_dtype=None, warmup=3, allow_frozen_dropout=False):
        if model.training and not allow_frozen_dropout and _stochastic(model):
            # the dropout kernels take host-drawn seeds: a captured train-mode forward replays ONE drop pattern, so S
            # "Monte-Carlo samples" through it would be S copies of the same sample (prediction_loop(predict_in_train=True))
            raise RuntimeError('GraphedForward: the model is in train mode with dropout / drop_path > 0 -- a replay would repeat '
                               'the captured dropout pattern.  Capture in eval mode, or pass allow_frozen_dropout=True.')
        self.model, self.autocast_dtype = model, autocast_dtype
        self.static_in = {k: v.clone() for k, v in example_batch.items()}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):                     # allocator, TunableOp decisions, lazy attribute memos: before capture
                self._forward()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_out = self._forward()

    def _forward(self):
        ctx = (torch.autocast('cuda', dtype=self.autocast_dtype) if self.autocast_dtype is not None
               else torch.autocast('cuda', enabled=False))
        with torch.no_grad(), ctx:
            return self.model(self.static_in)

    def __call__(self, batch):
        """the model's output for `batch` (same keys, shapes and dtypes as the example); the returned tensors are the graph's
        static outputs: clone them if they must survive the next call"""
        if batch.keys() != self.static_in.keys():          # a missing key would silently replay the example's data
            raise RuntimeError(f'GraphedForward: batch keys {sorted(batch)} differ from the captured {sorted(self.static_in)}')
        for k, v in batch.items():
            dst = self.static_in[k]
            if dst.shape != v.shape or dst.dtype != v.dtype:
                raise RuntimeError(f'GraphedForward: input {k} is {tuple(v.shape)} {v.dtype}, captured {tuple(dst.shape)} {dst.dtype}')
            dst.copy_(v, non_blocking=True)
        self.graph.replay()
        return self.static_out
