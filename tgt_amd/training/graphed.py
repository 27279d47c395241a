"""The training step as a hipGraph replay -- for the shapes where the HOST is the bottleneck.

The eager step costs the host ~65 ms of enqueue time whatever the batch (2100 launches from Python); at the BASELINE batch
(256 graphs of 32 nodes) the GPU needs 80 ms and hides it, at 64 graphs it needs 28 ms and at 8 graphs 16 ms -- the eager step
still takes 58-61 ms there (tools/probes/graph_step_probe.py).  `GraphedTrainingStep` captures ONE `Trainer.training_step`
(reference lib/training/training.py:439-470: forward, loss, backward, clipping, optimizer) on fixed-shape batches and replays
it; what a replay must not take from the capture is made a DEVICE value first:

  * dropout patterns: the kernels mix a device step counter into their seeds (ops.set_seed_counter, tgt_set_seed_counter); the
    graph starts with `counter += 1`.  DropPath factors and source-dropout masks are drawn by torch inside the graph (its
    generator is capture-aware).  Attention dropout inside the triplet / node attention kernels is not covered: refused.
  * the learning rate: Adam reads it from the optimizer's control block (Trainer.set_device_lr), refreshed before every replay;
    the bias corrections already use the device-side count of applied steps.
  * the batch: copied into the static tensors the capture saw.

Single rank by default; `allow_distributed=True` captures the bucketed RCCL all-reduces with the step (their hooks run at capture
time, the collectives become graph nodes): exercised with the nccl backend at world size 1 only.  Same kernels, same arguments: in graph-safe mode an eager step
and a replay give bit-identical parameters (tests/test_hip_trainer.py::test_graphed_training_step_equals_eager).
"""
import warnings
import weakref

import torch

from .. import ops
from .step import Trainer


def _leave_graph_mode(trainer_ref, counter=None):
    """back to eager: host seeds, pooled DropPath draws, the learning rate as a kernel argument.  Shared by close(), by the
    failure path of a capture and by the finalizer of an owner that was dropped without close() -- a Trainer left in graph-safe /
    device-lr mode by accident would silently repeat dropout patterns and keep a stale learning rate.
    `counter` is the ownership token: the mode is only given back when the registered seed counter is still THIS
    owner's -- an old owner that is collected (or closed) after a newer one took the mode over must not switch it off under it."""
    if counter is not None and ops._seed_counter[0] is not counter:
        return
    ops.set_seed_counter(None)
    ops.graph_safe_rng(False)
    tr = trainer_ref() if isinstance(trainer_ref, weakref.ReferenceType) else trainer_ref
    if tr is not None:
        tr.device_lr = False


def _attention_dropout(model):
    """largest attention-dropout probability inside the model's triplet / node attention modules"""
    worst = 0.0
    for m in model.modules():
        for name in ('attention_dropout', 'attn_dropout', 'triplet_dropout'):
            v = getattr(m, name, None)
            if isinstance(v, (int, float)):
                worst = max(worst, float(v))
    return worst


class GraphedTrainingStep:
    """step(batch) -> (outputs, loss) like Trainer.training_step, as a graph replay.

    trainer: a single-rank Trainer on the GPU; example_batch: a preprocessed batch (preprocess_batch) of the shapes every later
    batch will have; warmup: eager steps on it before the capture (allocator, GEMM tuning, autograd hooks) -- they are REAL
    optimizer steps.  The outputs / loss returned by step() are the capture's static tensors (overwritten by the next replay)."""

    def __init__(self, trainer, example_batch, warmup=3, update_losses=False, allow_distributed=False, counter=None):
        if not isinstance(trainer, Trainer):
            raise RuntimeError('GraphedTrainingStep: a tgt_amd Trainer is required')
        if trainer.distributed and not allow_distributed:
            # The bucketed RCCL all-reduces ARE capturable (the hooks run at capture time and their collectives become graph nodes:
            # verified with the nccl backend at world size 1, tests/test_hip_trainer.py) -- but no multi-GPU box has replayed
            # such a graph yet, and every rank must capture and replay in lock-step.  Opt in explicitly.
            raise RuntimeError('GraphedTrainingStep: a single-rank Trainer is required (allow_distributed=True captures the '
                               'bucketed all-reduce as well: untested beyond one rank)')
        dev = trainer.flat.param.device
        if dev.type != 'cuda':
            raise RuntimeError('GraphedTrainingStep needs the GPU')
        if trainer.model.training and _attention_dropout(trainer.model) > 0:
            raise RuntimeError('GraphedTrainingStep: attention dropout inside the attention kernels is seeded from the host; '
                               'a captured step would repeat its pattern (use the eager Trainer)')
        self.trainer, self.update_losses = trainer, update_losses
        self.static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in example_batch.items()}
        # counter given: this graph is one of several on the trainer (GraphedStepCache owns the mode and the counter); warmup may
        # then be 0 -- the trainer has already run eager steps of this shape in graph-safe mode
        self._owns_mode = counter is None
        self.counter = torch.zeros(1, dtype=torch.int64, device=dev) if counter is None else counter
        self._closed = False
        self._finalizer = None
        if self._owns_mode:
            ops.graph_safe_rng(True)
            ops.set_seed_counter(self.counter)
            trainer.set_device_lr(True)
            warmup = max(1, warmup)
            # an owner dropped without close() must not leave the process in graph-safe mode
            self._finalizer = weakref.finalize(self, _leave_graph_mode, weakref.ref(trainer), self.counter)
        try:
            self.stream = torch.cuda.Stream(dev)
            self.stream.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(self.stream):
                for _ in range(warmup):
                    self._eager()
            torch.cuda.current_stream(dev).wait_stream(self.stream)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            # With a process group alive, RCCL's watchdog THREAD polls events while this thread captures; under the default
            # ('global') capture mode such a call from another thread is an error that surfaces in that thread -- which answers it
            # with std::terminate (one abort at destroy_process_group in ~10 runs of the GPU suite, round 6).  'thread_local' is the
            # mode torch's own DDP-under-graphs path uses: only THIS thread's unsafe calls invalidate the capture.
            mode = 'thread_local' if trainer.distributed else 'global'
            with torch.cuda.graph(self.graph, stream=self.stream, capture_error_mode=mode):
                self.out = self._body()
        except BaseException:
            # warm-up or capture failed: give the mode back before the error travels on (the Trainer stays usable eagerly)
            self._closed = True
            self.graph = self.static = self.out = None
            if self._finalizer is not None:
                self._finalizer()
            raise
        self.replays = 0

    # what the graph holds: the counter bump and the trainer's own step on the static batch
    def _body(self):
        tr = self.trainer
        self.counter.add_(1)
        outputs, loss = tr.compute_gradients(self.static)
        tr.apply_gradients()
        if self.update_losses:
            tr.update_losses(loss, self.static)
        return outputs, loss

    def _eager(self):
        tr = self.trainer
        tr.global_step += 1
        tr.write_device_lr()
        return self._body()

    def step(self, batch):
        if self._closed:
            raise RuntimeError('GraphedTrainingStep is closed')
        tr = self.trainer
        for k, v in batch.items():
            if torch.is_tensor(v):
                dst = self.static[k]
                if dst.shape != v.shape or dst.dtype != v.dtype:
                    raise RuntimeError(f'GraphedTrainingStep: batch[{k!r}] is {tuple(v.shape)} {v.dtype}, the capture saw '
                                       f'{tuple(dst.shape)} {dst.dtype} (one graph per shape)')
                dst.copy_(v, non_blocking=True)
        tr.global_step += 1
        tr.write_device_lr()
        self.graph.replay()
        self.replays += 1
        return self.out

    def close(self):
        """back to eager: host seeds, pooled DropPath draws, the learning rate as an argument"""
        if not self._closed:
            self._closed = True
            if self._finalizer is not None:
                self._finalizer()            # runs _leave_graph_mode once; a later garbage collection does nothing
            self.graph = None
            self.static = self.out = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


def eager_graph_safe_step(trainer, counter, batch, update_losses=False):
    """one eager Trainer step in graph-safe mode -- exactly what a replay of a graph captured in that mode computes"""
    trainer.global_step += 1
    trainer.write_device_lr()
    counter.add_(1)
    outputs, loss = trainer.compute_gradients(batch)
    trainer.apply_gradients()
    if update_losses:
        trainer.update_losses(loss, batch)
    return outputs, loss


class GraphedStepCache:
    """One captured step per batch SHAPE (datasets whose padded sizes vary: PCQM batches are padded to their largest molecule,
    reference lib/data/...padded_collate), at most `max_graphs` of them (least recently used goes first), all on one Trainer and one
    dropout counter.  step(batch) is always exactly ONE optimizer step: the first `warmup` batches of a shape run eagerly (in
    graph-safe mode: the same arithmetic a replay performs), the next one is captured and replayed.  Bucket the padding (e.g. to
    multiples of 8 nodes) to keep the number of shapes small -- every graph owns the activations of its shape."""

    def __init__(self, trainer, warmup=2, max_graphs=4, update_losses=False, allow_distributed=False):
        if not isinstance(trainer, Trainer):
            raise RuntimeError('GraphedStepCache: a tgt_amd Trainer is required')
        if trainer.distributed and not allow_distributed:
            raise RuntimeError('GraphedStepCache: a single-rank Trainer is required (see GraphedTrainingStep)')
        dev = trainer.flat.param.device
        if trainer.model.training and _attention_dropout(trainer.model) > 0:
            raise RuntimeError('GraphedStepCache: attention dropout inside the attention kernels is seeded from the host')
        self.trainer, self.warmup, self.max_graphs = trainer, max(1, warmup), max(1, max_graphs)
        self.update_losses, self.allow_distributed = update_losses, allow_distributed
        self.counter = torch.zeros(1, dtype=torch.int64, device=dev)
        ops.graph_safe_rng(True)
        ops.set_seed_counter(self.counter)
        trainer.set_device_lr(True)
        self._finalizer = weakref.finalize(self, _leave_graph_mode, weakref.ref(trainer), self.counter)
        self.graphs, self.seen = {}, {}            # shape key -> GraphedTrainingStep (insertion order = recency) / eager steps so far
        self.captures = self.evictions = self.replays = self.eager_steps = 0
        self.thrash_window = 8 * self.max_graphs   # steps over which captures are compared with replays
        self._recent = []                          # 1 = this step captured a graph, 0 = it replayed one
        self.eager_fallback = False                # set once captures dominate: every later step runs eagerly (graph-safe mode)
        self._closed = False

    @staticmethod
    def _key(batch):
        """tensors by shape / dtype, anything else by VALUE: a captured graph bakes a Python-valued batch entry into its launch
        arguments, so a changed value must select (or capture) another graph, never replay a stale one"""
        key = []
        for k, v in sorted(batch.items()):
            if torch.is_tensor(v):
                key.append((k, tuple(v.shape), str(v.dtype)))
            else:
                try:
                    hash(v)
                    key.append((k, 'py', v))
                except TypeError:
                    key.append((k, 'py', repr(v)))
        return tuple(key)

    def step(self, batch):
        if self._closed:
            raise RuntimeError('GraphedStepCache is closed')
        if self.eager_fallback:
            self.eager_steps += 1
            return eager_graph_safe_step(self.trainer, self.counter, batch, self.update_losses)
        key = self._key(batch)
        gs = self.graphs.pop(key, None)
        if gs is None:
            n = self.seen.get(key, 0)
            if n < self.warmup:
                if len(self.seen) >= 64 * self.max_graphs and key not in self.seen:
                    self.seen.pop(next(iter(self.seen)))          # (bounded: the oldest shape starts its warm-up over)
                self.seen[key] = n + 1
                self.eager_steps += 1
                return eager_graph_safe_step(self.trainer, self.counter, batch, self.update_losses)
            # Thrash guard: with more live shapes than max_graphs every step evicts a graph and captures another one -- a capture
            # costs a multiple of an eager step and a private activation pool.  When captures outnumber replays over the last
            # `thrash_window` graph steps, stop capturing: eager steps in graph-safe mode compute exactly what a replay would.
            self._recent.append(1)
            del self._recent[:-self.thrash_window]
            if len(self._recent) >= self.thrash_window and 2 * sum(self._recent) > len(self._recent):
                warnings.warn(f'GraphedStepCache: {sum(self._recent)} captures in the last {len(self._recent)} graph steps with '
                              f'max_graphs={self.max_graphs}: more live batch shapes than graphs -- falling back to eager steps '
                              '(bucket the padding or raise max_graphs)', RuntimeWarning, stacklevel=2)
                self.eager_fallback = True
                for g_ in self.graphs.values():
                    g_.close()
                self.graphs.clear()
                self.eager_steps += 1
                return eager_graph_safe_step(self.trainer, self.counter, batch, self.update_losses)
            while len(self.graphs) >= self.max_graphs:
                old_key = next(iter(self.graphs))
                self.graphs.pop(old_key).close()
                self.evictions += 1
            gs = GraphedTrainingStep(self.trainer, batch, warmup=0, update_losses=self.update_losses,
                                     allow_distributed=self.allow_distributed, counter=self.counter)
            self.captures += 1
        else:
            self._recent.append(0)
            del self._recent[:-self.thrash_window]
        self.graphs[key] = gs                      # most recently used last
        self.replays += 1
        return gs.step(batch)

    def close(self):
        if not self._closed:
            self._closed = True
            for gs in self.graphs.values():
                gs.close()
            self.graphs.clear()
            self._finalizer()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
