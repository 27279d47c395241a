"""Data-parallel path on CPU: world_size-2 gloo.  The exchange logic of
tgt_amd.training.step.Trainer (flat gradient buffer, bucketed all-reduce fired
from autograd hooks, rank-0 parameter broadcast) is backend-agnostic; here it
drives the oracle model (CPU) because the HIP kernels need a GPU.  Property
checked: gradients of a 2-rank run on disjoint half-batches,
divided by world size, equal the 1-rank gradients on the concatenated batch
when the loss is a per-graph mean."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import golden_util as gu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gap_l1(outputs, batch, cfg):
    return torch.nn.functional.l1_loss(outputs, batch['target'])


def _make(kwargs, seed):
    from oracle import modules as om
    return gu.fill_params(om.TGT_Gap(**kwargs), seed=seed).train()


def _batch(B, N, seed):
    return gu.model_batch(dict(B=B, N=N, num_nodes=[N] * B), seed)


def _slice(batch, lo, hi):
    return {k: v[lo:hi] for k, v in batch.items()}


def _worker(rank, world, port, kwargs, bucket_mb, result, comm=None, exchange='all_reduce'):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from tgt_amd.training.step import Trainer, StepConfig
    torch.set_num_threads(2)
    model = _make(kwargs, seed=11 + rank)            # ranks start DIFFERENT: broadcast must fix it
    tr = Trainer(model, StepConfig(mixed_precision=None, bucket_mbytes=bucket_mb, grad_comm_dtype=comm, grad_exchange=exchange),
                 loss_fn=_gap_l1)
    full = _batch(4, 6, seed=5)
    part = _slice(full, 2 * rank, 2 * rank + 2)
    tr.global_step += 1
    tr.compute_gradients(part)
    if rank == 0:
        result['grad'] = (tr.flat.grad / world).clone()
        result['param'] = tr.flat.param.clone()
        result['nbuckets'] = -1 if tr.buckets is None else len(tr.buckets)
        result['bucket_order'] = list(tr.bucket_order)
        result['layer_of_bucket'] = [] if tr.buckets is None else [
            [n for n, p in model.named_parameters() if p.requires_grad][b[3]] for b in tr.buckets]
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('variant', ['bucketed', 'shared_weights'])
def test_two_rank_gradients_equal_single_rank(variant):
    kwargs = dict(gu.MODEL_CASES['gap_at_tiny'][1])
    kwargs['embed_3d_type'] = 'none'
    if variant == 'shared_weights':
        kwargs['layer_multiplier'] = 2                 # weight-shared repeats: single post-backward all-reduce
    bucket_mb = 0 if variant == 'bucketed' else 64     # 0 MB -> every parameter closes its own bucket
    mgr = mp.Manager()
    result = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), kwargs, bucket_mb, result), nprocs=2, join=True)

    from tgt_amd.training.step import Trainer, StepConfig
    model = _make(kwargs, seed=11)                     # rank 0's initial parameters
    tr = Trainer(model, StepConfig(mixed_precision=None), loss_fn=_gap_l1)
    assert torch.equal(result['param'], tr.flat.param)
    tr.compute_gradients(_batch(4, 6, seed=5))
    ref = tr.flat.grad
    err = (result['grad'] - ref).abs().max() / ref.abs().max()
    assert err < 1e-5, err
    if variant == 'bucketed':
        assert result['nbuckets'] > 10
        # the exchange overlaps the backward because buckets close -- and their all-reduce is launched -- in the order the
        # backward produces gradients: the LAST layers' buckets first.  Every bucket once; the encoder layers strictly from the
        # top of the stack down (buckets are laid out in parameter = forward order, so that is descending bucket index).
        order = result['bucket_order']
        assert sorted(order) == list(range(result['nbuckets']))
        layer = []
        for k in order:
            name = result['layer_of_bucket'][k]
            if '.TGT_layers.' in name:
                layer.append(int(name.split('.TGT_layers.')[1].split('.')[0]))
        assert len(set(layer)) >= 3 and layer == sorted(layer, reverse=True), layer
    else:
        assert result['nbuckets'] == -1


def test_bf16_gradient_exchange_option():
    """grad_comm_dtype='bf16' (half the bytes on the links): same gradients up to bfloat16 rounding of
    each rank's contribution"""
    kwargs = dict(gu.MODEL_CASES['gap_at_tiny'][1])
    kwargs['embed_3d_type'] = 'none'
    mgr = mp.Manager()
    result = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), kwargs, 0, result, 'bf16'), nprocs=2, join=True)
    from tgt_amd.training.step import Trainer, StepConfig
    tr = Trainer(_make(kwargs, seed=11), StepConfig(mixed_precision=None), loss_fn=_gap_l1)
    tr.compute_gradients(_batch(4, 6, seed=5))
    ref = tr.flat.grad
    err = (result['grad'] - ref).norm() / ref.norm()
    assert 0 < err < 1e-2, err          # not bit-identical (it IS compressed), within bfloat16 rounding


@pytest.mark.parametrize('variant', ['bucketed', 'shared_weights', 'bf16_wire'])
def test_reduce_scatter_all_gather_exchange_option(variant):
    """grad_exchange='reduce_scatter' (reduce-scatter + all-gather per bucket, the form that uses all xGMI
    links of the mesh at once): the same averaged gradients as the single-rank run, bucketed, for weight-shared models (one
    exchange after backward) and with the bfloat16 wire format."""
    kwargs = dict(gu.MODEL_CASES['gap_at_tiny'][1])
    kwargs['embed_3d_type'] = 'none'
    if variant == 'shared_weights':
        kwargs['layer_multiplier'] = 2
    mgr = mp.Manager()
    result = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), kwargs, 64 if variant == 'shared_weights' else 0, result,
                            'bf16' if variant == 'bf16_wire' else None, 'reduce_scatter'), nprocs=2, join=True)
    from tgt_amd.training.step import Trainer, StepConfig
    tr = Trainer(_make(kwargs, seed=11), StepConfig(mixed_precision=None), loss_fn=_gap_l1)
    assert torch.equal(result['param'], tr.flat.param)
    tr.compute_gradients(_batch(4, 6, seed=5))
    ref = tr.flat.grad
    if variant == 'bf16_wire':
        err = (result['grad'] - ref).norm() / ref.norm()
        assert 0 < err < 1e-2, err
    else:
        err = (result['grad'] - ref).abs().max() / ref.abs().max()
        assert err < 1e-5, err


def test_flat_state_views_alias_parameters():
    from tgt_amd.training.step import FlatState
    m = torch.nn.Sequential(torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))
    before = [p.detach().clone() for p in m.parameters()]
    f = FlatState(m)
    for p, b in zip(m.parameters(), before):
        assert torch.equal(p, b)
    f.param.mul_(2)
    for p, b in zip(m.parameters(), before):
        assert torch.equal(p, 2 * b)
    m(torch.ones(1, 5)).sum().backward()
    f.collect_grads()
    for p, v in zip(f.params, f.grad_views):
        assert torch.equal(p.grad, v)
    assert f.grad.abs().sum() > 0
    f.clear_grads()
    assert all(p.grad is None for p in m.parameters())


def test_lr_schedule_matches_reference_formula():
    from tgt_amd.training.step import lr_at, StepConfig
    cfg = StepConfig(max_lr=2e-3, min_lr=1e-6, lr_warmup_steps=100, lr_total_steps=1000)
    assert abs(lr_at(0, cfg) - 1e-6) < 1e-12
    assert abs(lr_at(100, cfg) - 2e-3) < 1e-12
    assert abs(lr_at(1000, cfg) - 1e-6) < 1e-9
    assert abs(lr_at(550, cfg) - (1e-6 + (2e-3 - 1e-6) * 0.5)) < 1e-9


def _torch_adam_(param, grad, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0,
                 shadow=None, clip_value=0.0, ctl=None):
    """TEST-SIDE stand-in for ops.adam_step_ (the HIP kernel has no CPU form and the product has no fallback): torch.optim.Adam's
    update on flat buffers, elementwise -- so a slice of the buffers updates exactly like the same elements inside the whole."""
    g = grad * grad_scale
    exp_avg.mul_(betas[0]).add_(g, alpha=1 - betas[0])
    exp_avg_sq.mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
    bc1, bc2 = 1 - betas[0] ** step, 1 - betas[1] ** step
    param.addcdiv_(exp_avg, (exp_avg_sq / bc2).sqrt_().add_(eps), value=-lr / bc1)


def _sharded_worker(rank, world, port, kwargs, shard, result):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from tgt_amd import ops
    from tgt_amd.training.step import Trainer, StepConfig
    ops.adam_step_ = _torch_adam_                    # (this process only)
    torch.set_num_threads(2)
    tr = Trainer(_make(kwargs, seed=11 + rank), StepConfig(mixed_precision=None, bucket_mbytes=0, grad_exchange='reduce_scatter',
                                                         shard_optimizer=shard, lr_warmup_steps=2, lr_total_steps=50), loss_fn=_gap_l1)
    assert tr.sharded == shard
    owned = tr.owned_slices() if shard else None
    for step in range(3):
        full = _batch(4, 6, seed=5 + step)
        tr.global_step += 1
        tr.compute_gradients(_slice(full, 2 * rank, 2 * rank + 2))
        if step == 0:
            result[f'grad{rank}'] = tr.flat.grad.clone()
        tr.apply_gradients()
    if shard:
        tr.consolidate_optimizer_state()
    result[f'param{rank}'] = tr.flat.param.clone()
    result[f'm{rank}'] = tr.flat.exp_avg.clone()
    result[f'v{rank}'] = tr.flat.exp_avg_sq.clone()
    if rank == 0:
        result['owned'] = owned
        result['buckets'] = [tuple(b[:2]) for b in tr.buckets]
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_optimizer_bookkeeping_on_gloo():
    """StepConfig.shard_optimizer, world 2 on CPU: the reduce-scatter leaves the reduced gradient on each rank's owned slices (the
    two ranks' slices partition every bucket), and three steps of (owned-slice update + all-gather of the parameters) end with the
    parameters and -- after the collective consolidate -- the moments of the REPLICATED reduce_scatter run, bit for bit, on both
    ranks.  (The update itself is a test-side torch restatement here; the HIP kernel's turn is tests/test_hip_trainer.py.)"""
    kwargs = dict(gu.MODEL_CASES['gap_at_tiny'][1])
    kwargs['embed_3d_type'] = 'none'
    mgr = mp.Manager()
    rep, sh = mgr.dict(), mgr.dict()
    mp.spawn(_sharded_worker, args=(2, _free_port(), kwargs, False, rep), nprocs=2, join=True)
    mp.spawn(_sharded_worker, args=(2, _free_port(), kwargs, True, sh), nprocs=2, join=True)
    owned0, buckets = sh['owned'], sh['buckets']
    assert len(buckets) > 10
    for (s0, e0), (a, b) in zip(buckets, owned0):
        assert a == s0 and b == s0 + (e0 - s0) // 2          # rank 0 owns the first half of every bucket
        # reduced (summed) gradient on the owned halves: rank 0's first half, rank 1's second half
        assert torch.equal(sh['grad0'][a:b], rep['grad0'][a:b])
        assert torch.equal(sh['grad1'][b:e0], rep['grad0'][b:e0])
    for r in (0, 1):
        assert torch.equal(sh[f'param{r}'], rep['param0']) and torch.equal(rep[f'param{r}'], rep['param0'])
        assert torch.equal(sh[f'm{r}'], rep['m0']) and torch.equal(sh[f'v{r}'], rep['v0'])
