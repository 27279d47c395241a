"""The training step the metric is defined on, MI355X-first.

Semantics follow the reference's hot loop
(lib/training/training.py:439-470 `training_step`, :410-418 + pretrain/scheme.py:60-88
`preprocess_batch` / `calculate_loss`, training_mixins.py:292-317 LR schedule,
training.py:152 DDP gradient averaging), re-designed for one process per GPU:

 * parameters, gradients and Adam moments live in FLAT float32 buffers;
   `p.data` / `p.grad` are views, so zeroing grads is one memset, the
   optimizer is one HIP kernel launch (tgt_adam_step) instead of 883 tensors,
   and the data-parallel exchange is an RCCL all-reduce of contiguous buckets
   launched from autograd hooks while the backward of earlier layers is still
   running (xGMI is point-to-point: few large messages, not 883 small ones).
 * no per-step host sync: the loss stays on the device; `.item()` only when
   the caller logs.
"""
import contextlib
import math
import weakref
from dataclasses import dataclass, field

import torch
import torch.distributed as dist
import torch.nn.functional as F

from .. import ops


@dataclass
class StepConfig:
    # optimizer / schedule (reference defaults: tgt_at_tp.yaml, training_mixins.py:276-288)
    max_lr: float = 2e-3
    min_lr: float = 1e-6
    lr_warmup_steps: int = 15000
    lr_total_steps: int = 300000
    cosine_halfwave: bool = False
    betas: tuple = (0.9, 0.999)
    eps: float = 1e-8
    weight_decay: float = 0.0
    # pretrain scheme (reference pretrain/scheme.py:14-27, tgt_at_tp.yaml)
    coords_noise: float = 0.2
    coords_noise_smooth: float = 1.0
    num_dist_bins: int = 512
    range_dist_bins: float = 8
    dist_loss_weight: float = 0.1
    # precision: None (fp32) | 'bf16' | 'fp16'
    mixed_precision: str = 'bf16'
    bucket_mbytes: int = 64
    # gradient exchange dtype: None = the float32 flat buffer as it is (reference DDP semantics, bit-exact parity);
    # 'bf16' = each bucket crosses the links as bfloat16 (half the xGMI bytes, SURVEY 5.8 / 8(e): an option, off by default)
    grad_comm_dtype: str = None
    # collective per bucket: 'all_reduce' (what torch DDP issues, reference training.py:152), or 'reduce_scatter': reduce-scatter +
    # all-gather of the bucket -- on the fully connected 8-GPU xGMI mesh each rank then owns 1/world of the sum and the two halves
    # use all seven links at once instead of a ring's one; same sums, another summation order
    grad_exchange: str = 'all_reduce'
    # with grad_exchange='reduce_scatter': ZeRO-1 style sharding of the optimizer step.  Every rank keeps only ITS 1/world shard of each
    # bucket's reduced gradient (no all-gather of gradients), runs tgt_adam_step on that shard of the flat buffers (the 3.1 GB Adam
    # sweep becomes 3.1 GB / world per rank) and the UPDATED float32 parameters are all-gathered; the 16-bit shadow is refreshed from
    # them locally.  Wire bytes equal the replicated reduce_scatter mode's (4 B reduce-scatter + 4 B all-gather per parameter), the
    # moments exist only on their owner: optimizer_state_dict() is then a COLLECTIVE (every rank calls it).  Needs: no dynamic loss
    # scale, no gradient clipping (their decisions need the whole gradient), no weight-shared stack.  Opt-in; never run on hardware
    # with more than one GPU.
    shard_optimizer: bool = False
    # gradient clipping (reference training.py:446-462; None = off, as in the shipped YAMLs)
    clip_grad_value: float = None
    clip_grad_norm: float = None
    # fp16 loss scaling: torch.cuda.amp.GradScaler() defaults (reference training.py:427-431)
    init_scale: float = 65536.0
    growth_factor: float = 2.0
    backoff_factor: float = 0.5
    growth_interval: int = 2000


def lr_at(step, cfg):
    """reference lib/training/training_mixins.py:292-317"""
    if step <= cfg.lr_warmup_steps:
        return cfg.min_lr + (cfg.max_lr - cfg.min_lr) * (step / cfg.lr_warmup_steps)
    t = (step - cfg.lr_warmup_steps) / (cfg.lr_total_steps - cfg.lr_warmup_steps)
    if cfg.cosine_halfwave:
        return float(cfg.min_lr + (cfg.max_lr - cfg.min_lr) * math.cos(0.5 * math.pi * t))
    return float(cfg.min_lr + (cfg.max_lr - cfg.min_lr) * (1 + math.cos(math.pi * t)) * 0.5)


def coords2dist(coords):
    return torch.norm(coords.unsqueeze(-2) - coords.unsqueeze(-3), dim=-1)


def preprocess_batch(batch, device, cfg, training=True, generator=None, add_noise=True):
    """host batch -> device, + edge_mask, + noised distance input
    (reference training.py:410-418, pretrain/scheme.py:60-76, commons.py:10-16).  The coordinate
    noise is added in training AND validation, as the reference's scheme does; add_noise=False is for
    tests that need the un-noised distances."""
    b = {k: v.to(device, non_blocking=True) for k, v in batch.items()}
    nm = b['node_mask']
    em = nm.unsqueeze(-1) * nm.unsqueeze(-2)
    b['edge_mask'] = em
    coords = b['dft_coords']
    if add_noise and cfg.coords_noise > 0:
        noise = torch.randn(coords.shape, dtype=coords.dtype, device=coords.device, generator=generator)
        noise = noise * cfg.coords_noise
        dm = coords2dist(coords) + (1 - em.float()) * 1e9
        noise = torch.softmax(-dm / cfg.coords_noise_smooth, dim=-1) @ noise
        coords = coords + noise
    b['dist_input'] = coords2dist(coords)
    return b


_HIP_XENT = True       # settled on (tests patch this); False: ATen's cross entropy on an fp32 image of the logits


def binned_distance_loss(logits, dist_target, edge_mask, num_bins, range_bins):
    """reference lib/training_schemes/pcqm/commons.py:19-48"""
    bsz = logits.size(0)
    bins = (dist_target * ((num_bins - 1) / range_bins)).long().clamp(0, num_bins - 1)
    if logits.is_cuda and _HIP_XENT:          # (268 MB of bf16 logits at the BASELINE batch: no fp32 image, no separate softmax passes)
        xent = ops.cross_entropy_rows(logits.reshape(-1, num_bins), bins.reshape(-1)).view(bsz, -1)
    else:
        xent = F.cross_entropy(logits.reshape(-1, num_bins), bins.reshape(-1), reduction='none').view(bsz, -1)
    m = edge_mask.to(xent.dtype).view(bsz, -1)
    return (xent * m).sum() / (m.sum() + 1e-9)


def pretrain_loss(outputs, batch, cfg):
    """L1(gap) + w * binned-distance x-ent (reference pretrain/scheme.py:78-88);
    the target is float64, so the loss is too (quirk Q9)."""
    gap, logits = outputs
    prim = F.l1_loss(gap, batch['target'])
    dl = binned_distance_loss(logits, coords2dist(batch['dft_coords']), batch['edge_mask'],
                              cfg.num_dist_bins, cfg.range_dist_bins)
    return prim + cfg.dist_loss_weight * dl


class FlatState:
    """Flat float32 parameter / gradient / Adam-moment buffers with per-tensor views."""

    def __init__(self, model, align=64):
        self.params = [p for p in model.parameters() if p.requires_grad]
        seen, uniq = set(), []
        for p in self.params:
            if id(p) not in seen:
                seen.add(id(p))
                uniq.append(p)
        self.params = uniq
        for p in self.params:              # shadows of an earlier Trainer on this model would go stale silently
            if hasattr(p, '_lp'):
                del p._lp
        dev = self.params[0].device
        self.offsets, off = [], 0
        for p in self.params:
            assert p.dtype == torch.float32, 'parameters stay float32 (mixed precision casts per op)'
            self.offsets.append(off)
            off += -(-p.numel() // align) * align
        self.numel = off
        self.param = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad_views = []
        for p, o in zip(self.params, self.offsets):
            self.param[o:o + p.numel()].view_as(p).copy_(p.data)
            p.data = self.param[o:o + p.numel()].view_as(p)
            self.grad_views.append(self.grad[o:o + p.numel()].view_as(p))
            p.grad = None
        self.shadow = None
        self._registered = []
        if dev.type == 'cuda':
            # kernels that end a weight gradient inside a Trainer's backward write it into these slices (ops._grad_dst)
            ops.register_flat_grads(self.params, self.grad_views)
            self._registered = [p.data_ptr() for p in self.params]
            weakref.finalize(self, ops.unregister_flat_grads, list(self._registered))      # (the registry holds views of self.grad)

    def release(self):
        ops.unregister_flat_grads(self._registered)
        self._registered = []

    def views(self, flat):
        """per-parameter views of a flat buffer laid out like self.param"""
        return [flat[o:o + p.numel()].view_as(p) for p, o in zip(self.params, self.offsets)]

    def make_shadow(self, dtype):
        """16-bit copy of all parameters (views hung on each parameter as `_lp`); the Adam
        kernel keeps it current, so the step has no per-tensor weight casts."""
        self.shadow = self.param.to(dtype)
        for p, o in zip(self.params, self.offsets):
            p._lp = self.shadow[o:o + p.numel()].view_as(p)

    def drop_shadow(self):
        self.shadow = None
        for p in self.params:
            if hasattr(p, '_lp'):
                del p._lp

    def clear_grads(self):
        for p in self.params:
            p.grad = None

    def collect_grads(self, indices=None):
        """autograd's per-tensor gradients -> the flat buffer (one multi-tensor copy instead of
        883 accumulate kernels); parameters without a gradient contribute zeros."""
        idx = range(len(self.params)) if indices is None else indices
        if self.param.is_cuda:
            ops.flush_deferred()                           # the closing sums of the backward so far (ops.DeferredSums), per stream
            ops.wait_side_streams(self.param.device)       # node-FFN gradients come from a second stream
        src, dst = [], []
        for i in idx:
            g = self.params[i].grad
            if g is None:
                self.grad_views[i].zero_()
            elif g.data_ptr() != self.grad_views[i].data_ptr():     # (else: the kernel that ended this gradient wrote it in place, ops._grad_dst)
                src.append(g if g.dtype == torch.float32 else g.float())
                dst.append(self.grad_views[i])
        if src:
            torch._foreach_copy_(dst, src)


def _parameter_reused(root, params):
    """True when a parameter enters the autograd graph under `root` through more than one edge, at least one of them from one of
    this package's own autograd Functions.  autograd then SUMS the incoming gradients in the input buffer of the parameter's
    AccumulateGrad node on a stream of its choosing: neither a gradient produced on the forked parameter-gradient stream nor a
    registered-but-not-yet-computed closing sum (ops.trainer_backward) may take part in that.  (Library nodes produce their
    gradients on the stream autograd knows about: the two look-ups of the 3-D kernel's per-edge-type tables are fine.)  One walk
    over the graph of the first step."""
    ids = {id(p) for p in params}
    uses, own, seen, stack = {}, set(), set(), [root.grad_fn]
    while stack:
        fn = stack.pop()
        if fn is None or fn in seen:
            continue
        seen.add(fn)                      # (the node objects themselves: they stay alive, one Python wrapper per graph node)
        custom = isinstance(fn, torch.autograd.function.BackwardCFunction)
        for nxt, _ in fn.next_functions:
            if nxt is None:
                continue
            v = getattr(nxt, 'variable', None)
            if v is not None and id(v) in ids:
                uses[id(v)] = uses.get(id(v), 0) + 1
                if custom:
                    own.add(id(v))
            else:
                stack.append(nxt)
    return any(n > 1 and k in own for k, n in uses.items())


class Trainer:
    """The reference's per-step sequence (lib/training/training.py:439-470 + the loop body of
    :520-532) on flat buffers.  Everything the reference decides on the host per step -- GradScaler's
    found_inf / skip / scale update, the clipping coefficient, the running loss -- is decided on the
    device in a 16-float control block (`self.ctl`, ops.CTL_*), so the step never synchronises."""

    def __init__(self, model, cfg=None, loss_fn=pretrain_loss, process_group=None, force_distributed=False):
        self.model, self.cfg, self.loss_fn = model, (cfg or StepConfig()), loss_fn
        self.flat = FlatState(model)
        self.global_step = 0
        self._applied_steps = 0            # optimizer steps applied (host count; the fp16 path counts on the device)
        # force_distributed: run the bucketed all-reduce path even with one rank (tests)
        self.distributed = dist.is_available() and dist.is_initialized() and \
            (dist.get_world_size(process_group) > 1 or force_distributed)
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if self.distributed else 1
        self._handles = []
        self._armed = False
        self.buckets = None                # [start, end, n_params, first_param_index] per bucket (distributed runs: _setup_buckets)
        self._hooks_registered = False
        self.bucket_order = []             # bucket indices in the order their all-reduce was launched during the last backward
        self._comm_events = []             # (before, after) event pairs around the waits for the gradient exchange, one per step
        dev = self.flat.param.device
        self.dynamic_scale = self.cfg.mixed_precision == 'fp16'
        # the device control block drives the step whenever a per-step decision exists
        self.use_ctl = dev.type == 'cuda' and (self.dynamic_scale or bool(self.cfg.clip_grad_norm))
        self.ctl = torch.zeros(ops.CTL_SIZE, dtype=torch.float32, device=dev)
        self.ctl[ops.CTL_SCALE] = self.cfg.init_scale if self.dynamic_scale else 1.0
        self.ctl[ops.CTL_COEF] = 1.0
        self.sharded = False               # StepConfig.shard_optimizer in effect (set by _setup_buckets)
        self.rank = dist.get_rank(process_group) if self.distributed else 0
        if self.distributed:
            # replicas start from rank 0's parameters (DDP's broadcast at wrap, training.py:152)
            dist.broadcast(self.flat.param, src=0, group=self.pg)
            self._setup_buckets()
        if self.cfg.shard_optimizer and self.distributed and not self.sharded:
            raise RuntimeError('StepConfig.shard_optimizer needs grad_exchange="reduce_scatter", bucket sizes divisible by the world size, no '
                               'weight-shared stack, no dynamic loss scale (fp16) and no gradient clipping')
        self._wt = None
        self._shared = getattr(model, 'layer_multiplier', 1) > 1 or getattr(getattr(model, 'encoder', None), 'layer_multiplier', 1) > 1
        self._reuse_checked = False        # the first backward walks its graph once for parameters that enter it more than once
        self.device_lr = False             # True: Adam takes the learning rate from ctl[CTL_LR] (set_device_lr)
        if self.cfg.mixed_precision in ('bf16', 'fp16') and self.flat.param.is_cuda:
            self.flat.make_shadow(torch.bfloat16 if self.cfg.mixed_precision == 'bf16' else torch.float16)
            # W^T of the shadowed weights for the data-gradient kernels, refreshed in one launch after every optimizer step
            self._wt = ops.WeightTransposes(self.flat.shadow)
        # parameters changed behind the trainer's back (the reference loads pretrained weights AFTER the
        # trainer exists, tgt_training.py:174-189): the 16-bit shadows must follow
        # (a weak reference: the hook must not keep a dropped trainer -- and its four flat buffers -- alive)
        wself = weakref.ref(self)

        def _post_load(m, keys):
            t = wself()
            if t is not None and not t._closed:
                t.refresh_shadow()
        self._hooks = [model.register_load_state_dict_post_hook(_post_load)]
        # this trainer orders its own gradient collection after both streams; the node side stream is
        # only safe under it (torch DDP's reducer does not know about it): see ops.side_stream.  close()
        # (or garbage collection of the trainer) gives the ownership back.
        ops.side_stream.owner_present(True)
        self._closed = False
        self._finalizer = weakref.finalize(self, Trainer._release, self._hooks, self._wt, list(self.flat._registered))

    @staticmethod
    def _release(hooks, wt=None, grad_ptrs=()):
        for h in hooks:
            h.remove()
        hooks.clear()
        if wt is not None:
            wt.close()
        ops.unregister_flat_grads(grad_ptrs)        # (the registry holds views of this trainer's gradient buffer)
        ops.side_stream.owner_present(False)

    def close(self):
        """Give up the step: the node side stream loses this owner (with none left, ops.side_stream runs its blocks on the
        current stream again, which is what torch DDP's reducer needs), the load_state_dict hook is removed and the
        parameters' 16-bit shadows are dropped.  The parameters keep living in the flat buffer.  Idempotent."""
        if not self._closed:
            self._closed = True
            self._finalizer()               # runs _release once; a later garbage collection does nothing
            self.flat.drop_shadow()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    @property
    def loss_scale(self):
        """current loss scale (a host read of the device value: a sync; for logging / checkpoints)"""
        return float(self.ctl[ops.CTL_SCALE])

    # ---- data-parallel gradient exchange ---------------------------------
    def _setup_buckets(self):
        f, limit = self.flat, self.cfg.bucket_mbytes * (1 << 20) // 4
        shared = getattr(self.model, 'layer_multiplier', 1) > 1 or \
            getattr(getattr(self.model, 'encoder', None), 'layer_multiplier', 1) > 1
        self.buckets = []          # [start, end, n_params, first_param_index]
        self._bucket_of = {}
        if shared:                 # weight-shared repeats accumulate several times: reduce after backward
            self.buckets = None
            return
        start, count, first = 0, 0, 0
        ends = [o + (-(-p.numel() // 64) * 64) for p, o in zip(f.params, f.offsets)]
        # Buckets are contiguous slices of the flat buffer (= parameter order) and a bucket's all-reduce starts when its LAST gradient
        # arrives.  Two rules keep the part of the exchange that cannot hide behind the backward small (round 5; only the LAST
        # bucket(s) to close are exposed):
        #  * a bucket never spans two top-level modules: `model.parameters()` lists the encoder, then the input embedding, then the
        #    heads -- the embedding's gradients arrive at the very END of the backward, the heads' and the top layers' at its start,
        #    and one bucket holding all three would keep the top layers' 60 MB waiting for the embedding (observed launch order
        #    5, 4, 3, 2, 1, 0, 6 with seven buckets: two full buckets exposed instead of one);
        #  * the first bucket (the bottom layers: the last gradients of the encoder) is a quarter of the others.
        names = {id(p): n for n, p in self.model.named_parameters()}
        group = [names.get(id(p), '').split('.', 1)[0] for p in f.params]
        for i, (p, o) in enumerate(zip(f.params, f.offsets)):
            self._bucket_of[id(p)] = len(self.buckets)
            count += 1
            lim = limit // 4 if not self.buckets else limit
            last = i == len(f.params) - 1
            # (a module change closes the bucket only once it holds >= 1 MB: the heads' few small tensors share one)
            if ends[i] - start >= lim or last or (group[i + 1] != group[i] and ends[i] - start >= (1 << 18)):
                self.buckets.append([start, ends[i], count, first])
                start, count, first = ends[i], 0, i + 1
        self._pending = [b[2] for b in self.buckets]
        cfg = self.cfg
        self.sharded = bool(cfg.shard_optimizer and cfg.grad_exchange == 'reduce_scatter' and cfg.grad_comm_dtype is None and
                            self.world > 1 and all((b[1] - b[0]) % self.world == 0 for b in self.buckets) and
                            cfg.mixed_precision != 'fp16' and not cfg.clip_grad_norm and not cfg.clip_grad_value)
        # The hooks are registered AFTER the first forward (compute_gradients), not here: a post-accumulate hook keeps the
        # parameter's AccumulateGrad node alive for good, and that node runs on the stream that was current when it was CREATED.
        # Created here, every node would sit on the default stream while the node channel's gradients are produced on the side
        # stream -- torch's "AccumulateGrad node's stream does not match" case: one cross-stream wait per node-channel parameter in
        # every backward.  The first forward creates each node on the stream that uses its parameter; the hook then pins THAT one.
        self._hooks_registered = False

    def _grad_ready(self, p):
        if not self._armed:
            return
        k = self._bucket_of[id(p)]
        self._pending[k] -= 1
        if self._pending[k] == 0:
            self._reduce_bucket(k)

    def _all_reduce_async(self, g):
        """all-reduce (sum) of a slice of the flat gradient; returns something with .wait()"""
        assert self.cfg.grad_comm_dtype in (None, 'bf16'), self.cfg.grad_comm_dtype
        assert self.cfg.grad_exchange in ('all_reduce', 'reduce_scatter'), self.cfg.grad_exchange
        wire = g.to(torch.bfloat16) if self.cfg.grad_comm_dtype == 'bf16' else g
        pg, world = self.pg, self.world
        if self.cfg.grad_exchange == 'reduce_scatter' and world > 1 and wire.numel() % world == 0:
            # the sum in two halves: every rank reduces its 1/world shard of the bucket (reduce-scatter), then the reduced
            # shards are gathered back over the bucket
            n = wire.numel() // world
            shard = torch.empty(n, dtype=wire.dtype, device=wire.device)
            h1 = dist.reduce_scatter_tensor(shard, wire, group=pg, async_op=True)
            if self.sharded:
                # sharded optimizer: the reduced shard goes back into ITS slice of the flat gradient and that is all -- the rest of the
                # bucket keeps this rank's local (unreduced) gradients, which nothing reads: Adam runs on the owned slices only
                class _One:
                    def wait(_self):
                        h1.wait()
                        g[self.rank * n:(self.rank + 1) * n].copy_(shard)
                return _One()

            class _Two:
                def wait(_self):
                    h1.wait()
                    dist.all_gather_into_tensor(wire, shard, group=pg)
                    if wire is not g:
                        g.copy_(wire)
            return _Two()
        h = dist.all_reduce(wire, group=pg, async_op=True)
        if wire is g:
            return h

        class _Done:
            def wait(_self):
                h.wait()
                g.copy_(wire)
        return _Done()

    def _reduce_bucket(self, k):
        s, e, n, first = self.buckets[k]
        self.bucket_order.append(k)
        self.flat.collect_grads(range(first, first + n))
        self._handles.append(self._all_reduce_async(self.flat.grad[s:e]))

    def _finish_reduce(self):
        if not self.distributed:
            self.flat.collect_grads()
            return
        if self.buckets is None:
            self.flat.collect_grads()
            self._all_reduce_async(self.flat.grad).wait()
            return
        # parameters that received no gradient this step never fire their hook
        for k, left in enumerate(self._pending):
            if left > 0:
                self._reduce_bucket(k)
        # What of the exchange is NOT hidden behind the backward: the time the step's stream spends blocked on the collectives'
        # completion (an event before and after the waits; nothing else is queued in between).  Read later, off the step's path
        # (comm_exposed_ms): no host synchronisation here.
        # (not under a hipGraph capture: an event recorded while capturing cannot be passed to elapsed_time, and replays would
        #  add no pairs -- a captured step reports no exposed-communication time)
        timed = self.flat.param.is_cuda and len(self._comm_events) < 4096 and not torch.cuda.is_current_stream_capturing()
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for h in self._handles:
            h.wait()
        if timed:
            e1.record()
            self._comm_events.append((e0, e1))
        self._handles.clear()
        self._pending = [b[2] for b in self.buckets]

    def comm_exposed_ms(self, reset=True):
        """per-step milliseconds the step's stream waited for the gradient exchange after the backward had ended (the part of the
        all-reduce that the backward did not hide), for the steps since the last call; synchronises (call it when logging)"""
        if not self._comm_events:
            return []
        torch.cuda.synchronize(self.flat.param.device)
        out = [a.elapsed_time(b) for a, b in self._comm_events]
        if reset:
            self._comm_events.clear()
        return out

    def refresh_shadow(self):
        """the 16-bit parameter shadows <- the float32 parameters.  Runs by itself after
        model.load_state_dict (hook); call it after any other out-of-band parameter change."""
        if self.flat.shadow is not None:
            self.flat.shadow.copy_(self.flat.param)
            if self._wt is not None:
                self._wt.refresh()

    # ---- one step ----------------------------------------------------------
    def autocast(self):
        mp = self.cfg.mixed_precision
        if mp is None:
            return torch.autocast('cuda', enabled=False)
        return torch.autocast('cuda', dtype=torch.bfloat16 if mp == 'bf16' else torch.float16)

    def compute_gradients(self, batch):
        """zero grads -> autocast forward + loss -> backward (+ overlapped RCCL
        all-reduce of gradient buckets).  Leaves SUMMED (and, in fp16, still loss-scaled) gradients
        in flat.grad."""
        cfg, f = self.cfg, self.flat
        f.clear_grads()
        self._armed = True
        self.bucket_order = []
        ops.begin_step()                   # (graph-safe randomness: the per-call seed positions start over; a no-op otherwise)
        with self.autocast():
            outputs = self.model(batch)
            loss = self.loss_fn(outputs, batch, cfg)
        # GradScaler.scale(loss): the scale is a device scalar, so a changed scale costs no sync
        # parameter gradients may come from a forked stream (collect_grads joins it) -- unless a parameter takes part in the graph
        # more than once (weight-shared stacks): autograd then ADDS into .grad on the current stream, unordered with the fork
        if self.buckets is not None and not self._hooks_registered:
            for p in f.params:
                p.register_post_accumulate_grad_hook(self._grad_ready)
            self._hooks_registered = True
        if not self._shared and not self._reuse_checked:
            # any OTHER parameter reuse (tied weights, a module applied twice): found in the graph of the first step
            self._shared = _parameter_reused(loss, f.params)
            self._reuse_checked = True
        with (ops.trainer_backward() if not self._shared else contextlib.nullcontext()):
            (loss * self.ctl[ops.CTL_SCALE].to(loss.dtype) if self.dynamic_scale else loss).backward()
        self._armed = False
        self._finish_reduce()
        if self.flat.param.is_cuda:
            ops.release_stream_keepalive(self.flat.param.device)      # both streams joined: cross-stream tensors may go back to their pools
        # Hand back values, not graph roots.  A caller that keeps last step's loss keeps that step's autograd nodes alive through
        # it -- including every parameter's AccumulateGrad node, which the NEXT forward then re-uses with the stream it was created
        # on; with the node channel on its own stream that is torch's "AccumulateGrad node's stream does not match" situation: an
        # extra cross-stream wait per parameter in every backward, and a capture hazard.  Detached, the nodes die with the backward
        # and each forward creates them on the stream that uses the parameter.
        loss = loss.detach()
        outputs = tuple(o.detach() if torch.is_tensor(o) else o for o in outputs) if isinstance(outputs, (tuple, list)) else \
            (outputs.detach() if torch.is_tensor(outputs) else outputs)
        return outputs, loss

    def apply_gradients(self):
        """[GradScaler.unscale_ + found_inf] -> clip_grad_value_ -> clip_grad_norm_ -> Adam ->
        [GradScaler.update], the order of training.py:451-469, as (at most) two passes over the flat
        gradient and one over the optimizer state; averages over ranks via the gradient multiplier."""
        cfg, f = self.cfg, self.flat
        # device_lr (a captured step, training/graphed.py): the kernel reads the rate from the control block, which the owner of
        # the graph refreshes before every replay
        lr = -1.0 if self.device_lr else lr_at(self.global_step, cfg)
        if self.sharded:
            return self._apply_gradients_sharded(lr)
        if self.use_ctl:
            ops.grad_scaler_step_(f.grad, self.ctl, self.world, cfg.clip_grad_value, cfg.clip_grad_norm, self.dynamic_scale,
                                  cfg.growth_factor, cfg.backoff_factor, cfg.growth_interval)
            ops.adam_step_(f.param, f.grad, f.exp_avg, f.exp_avg_sq, 0, lr, betas=cfg.betas, eps=cfg.eps,
                           weight_decay=cfg.weight_decay, shadow=f.shadow, clip_value=cfg.clip_grad_value, ctl=self.ctl)
        else:
            self._applied_steps += 1
            ops.adam_step_(f.param, f.grad, f.exp_avg, f.exp_avg_sq, self._applied_steps, lr,
                           betas=cfg.betas, eps=cfg.eps, weight_decay=cfg.weight_decay, grad_scale=1.0 / self.world,
                           shadow=f.shadow, clip_value=cfg.clip_grad_value)
        if self._wt is not None:
            self._wt.refresh()              # (the shadow just changed: every registered W^T follows, one launch)

    def owned_slices(self):
        """[(start, end)] of the flat buffers this rank owns under the sharded optimizer: its 1/world piece of every bucket"""
        out = []
        for s0, e0, _, _ in self.buckets:
            n = (e0 - s0) // self.world
            out.append((s0 + self.rank * n, s0 + (self.rank + 1) * n))
        return out

    def _apply_gradients_sharded(self, lr):
        """Adam on the owned slices (their gradients are the reduced sums, see _all_reduce_async), then the updated float32 parameters
        of every bucket are all-gathered in place and the 16-bit shadow follows locally."""
        cfg, f = self.cfg, self.flat
        if self.use_ctl or self.device_lr:
            raise RuntimeError('sharded optimizer: the device control block (dynamic loss scale / clipping / captured steps) is not supported')
        self._applied_steps += 1
        for a, b in self.owned_slices():
            ops.adam_step_(f.param[a:b], f.grad[a:b], f.exp_avg[a:b], f.exp_avg_sq[a:b], self._applied_steps, lr,
                           betas=cfg.betas, eps=cfg.eps, weight_decay=cfg.weight_decay, grad_scale=1.0 / self.world)
        handles = []
        for (s0, e0, _, _), (a, b) in zip(self.buckets, self.owned_slices()):
            # in place: rank r's input is its own slice of the output
            handles.append(dist.all_gather_into_tensor(f.param[s0:e0], f.param[a:b], group=self.pg, async_op=True))
        for h in handles:
            h.wait()
        self._moments_consolidated = False
        self.refresh_shadow()                # (shadow <- parameters, one cast pass; the registered W^T follow)

    def consolidate_optimizer_state(self):
        """COLLECTIVE (sharded optimizer only): all-gather the Adam moments so that every rank holds all of them -- call it on every
        rank before optimizer_state_dict() / state_dict(); the next optimizer step un-consolidates them again."""
        if not self.sharded:
            return
        f = self.flat
        for (s0, e0, _, _), (a, b) in zip(self.buckets, self.owned_slices()):
            dist.all_gather_into_tensor(f.exp_avg[s0:e0], f.exp_avg[a:b].clone(), group=self.pg)
            dist.all_gather_into_tensor(f.exp_avg_sq[s0:e0], f.exp_avg_sq[a:b].clone(), group=self.pg)
        self._moments_consolidated = True

    def set_device_lr(self, on=True):
        """the learning rate as a device value (ctl[CTL_LR]) instead of a kernel argument; needs the control-block path of the
        optimizer, which is switched on with it"""
        self.device_lr = bool(on)
        if on:
            if not self.use_ctl:           # the applied-step count moves to the device with it
                self.ctl[ops.CTL_STEPS] = float(self._applied_steps)
            self.use_ctl = True
            self.write_device_lr()

    def write_device_lr(self):
        self.ctl[ops.CTL_LR] = lr_at(max(self.global_step, 1), self.cfg)

    def training_step(self, batch):
        """batch: device tensors incl. edge_mask / dist_input (see preprocess_batch).
        Returns (outputs, loss) like the reference's training_step (training.py:439-470)."""
        self.global_step += 1
        if self.device_lr:
            # graph-safe mode (training/graphed.py owns this trainer) and somebody steps it EAGERLY -- e.g. the odd-shaped last
            # batch of an epoch that the one-graph-per-shape owner refused: do what a replay does around the captured body, so
            # that the step neither runs on a stale learning rate nor repeats the previous step's dropout patterns
            self.write_device_lr()
            ops.bump_seed_counter()
        outputs, loss = self.compute_gradients(batch)
        self.apply_gradients()
        return outputs, loss

    # ---- running loss (reference tgt_training.py:137-171), sync-free -------
    def initialize_losses(self):
        self.ctl[ops.CTL_LOSS:ops.CTL_LOSS_LO + 1] = 0          # loss, samples, NaN streak, low part of the loss sum
        self.ctl[ops.CTL_SAMPLES_LO] = 0

    def update_losses(self, loss, batch):
        """total_loss += loss * samples, total_samples += samples (summed over ranks; a NaN step loss is
        skipped under mixed precision unless 10 came in a row).  The reference pays two scalar
        all-reduces and two `.item()` per step for this; here it is one 8-byte all-reduce between two
        one-thread kernels, and the host reads the mean only when it logs (mean_loss)."""
        samples = float(batch['num_nodes'].shape[0])
        mixed = self.cfg.mixed_precision is not None
        loss = loss.detach()
        if loss.dtype not in (torch.float32, torch.float64):
            loss = loss.float()
        if not self.distributed:
            ops.loss_accumulate_(loss, samples, self.ctl, mixed, 3)
            return
        ops.loss_accumulate_(loss, samples, self.ctl, mixed, 1)
        dist.all_reduce(self.ctl[ops.CTL_PAIR:ops.CTL_PAIR + 2], group=self.pg)
        ops.loss_accumulate_(None, samples, self.ctl, mixed, 2)

    def mean_loss(self):
        """running mean loss per sample since initialize_losses (host read: synchronises)"""
        c = self.ctl.tolist()
        return (c[ops.CTL_LOSS] + c[ops.CTL_LOSS_LO]) / (c[ops.CTL_SAMPLES] + c[ops.CTL_SAMPLES_LO] + 1e-12)

    def step_stats(self):
        """host copy of the control block's counters (synchronises; for logging / tests)"""
        c = self.ctl.tolist()
        applied = int(c[ops.CTL_STEPS]) if self.use_ctl else self._applied_steps
        return dict(loss_scale=c[ops.CTL_SCALE], growth_tracker=int(c[ops.CTL_TRACKER]), found_inf=bool(c[ops.CTL_FOUND_INF]),
                    applied_steps=applied, skipped_steps=int(c[ops.CTL_SKIPPED]), grad_norm=c[ops.CTL_NORM],
                    clip_coef=c[ops.CTL_COEF])

    # ---- checkpoint / resume (reference training.py:290-360: training_state.pt, optimizer_state.pt,
    #      grad_scaler_state.pt; model_state.pt is model.state_dict()) ---------
    def optimizer_state_dict(self):
        """torch.optim.Adam-format state ({'state': {i: {step, exp_avg, exp_avg_sq}}, 'param_groups': [...]}),
        parameter order = model.parameters(): loads into torch.optim.Adam(model.parameters())"""
        f = self.flat
        steps = self.step_stats()['applied_steps']
        if self.sharded and steps > 0 and not getattr(self, '_moments_consolidated', False):
            raise RuntimeError('sharded optimizer: the Adam moments live on their owner ranks -- call consolidate_optimizer_state() on EVERY rank '
                               '(a collective) before optimizer_state_dict() / state_dict()')
        state = {i: dict(step=torch.tensor(float(steps)), exp_avg=m.detach().clone().cpu(), exp_avg_sq=v.detach().clone().cpu())
                 for i, (m, v) in enumerate(zip(f.views(f.exp_avg), f.views(f.exp_avg_sq)))} if steps > 0 else {}
        group = dict(lr=lr_at(self.global_step, self.cfg), betas=tuple(self.cfg.betas), eps=self.cfg.eps,
                     weight_decay=self.cfg.weight_decay, amsgrad=False, maximize=False, foreach=None, capturable=False,
                     differentiable=False, fused=None, params=list(range(len(f.params))))
        return dict(state=state, param_groups=[group])

    def load_optimizer_state_dict(self, sd):
        """accepts torch.optim.Adam's layout (per-parameter 'step') and apex FusedAdam's (the step lives in
        param_groups[0]['step'], reference training.py:339)"""
        f = self.flat
        st = sd.get('state', {})
        steps = 0
        if st:
            if len(st) != len(f.params):
                raise ValueError(f'optimizer state has {len(st)} entries, the model {len(f.params)} parameters')
            for i, (m, v) in enumerate(zip(f.views(f.exp_avg), f.views(f.exp_avg_sq))):
                e = st[i] if i in st else st[str(i)]
                m.copy_(e['exp_avg'])
                v.copy_(e['exp_avg_sq'])
                if 'step' in e:
                    steps = int(float(e['step']))
        else:
            f.exp_avg.zero_()
            f.exp_avg_sq.zero_()
        if steps == 0 and sd.get('param_groups') and 'step' in sd['param_groups'][0]:
            steps = int(sd['param_groups'][0]['step'])
        self._applied_steps = steps
        self.ctl[ops.CTL_STEPS] = float(steps)

    def grad_scaler_state_dict(self):
        """torch.cuda.amp.GradScaler.state_dict() layout"""
        s = self.step_stats()
        return dict(scale=s['loss_scale'], growth_factor=self.cfg.growth_factor, backoff_factor=self.cfg.backoff_factor,
                    growth_interval=self.cfg.growth_interval, _growth_tracker=s['growth_tracker'])

    def load_grad_scaler_state_dict(self, sd):
        if not sd:
            return
        self.ctl[ops.CTL_SCALE] = float(sd['scale'])
        self.ctl[ops.CTL_TRACKER] = float(sd.get('_growth_tracker', 0))
        for k in ('growth_factor', 'backoff_factor', 'growth_interval'):
            if k in sd:
                setattr(self.cfg, k, type(getattr(self.cfg, k))(sd[k]))

    def state_dict(self):
        """everything a run needs to resume besides model.state_dict()"""
        return dict(training_state=dict(global_step=self.global_step), optimizer=self.optimizer_state_dict(),
                    grad_scaler=self.grad_scaler_state_dict() if self.dynamic_scale else {})

    def load_state_dict(self, sd):
        self.global_step = int(sd.get('training_state', {}).get('global_step', 0))
        self.load_optimizer_state_dict(sd.get('optimizer', {}))
        self.load_grad_scaler_state_dict(sd.get('grad_scaler', {}))
        self.refresh_shadow()
