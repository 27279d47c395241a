"""The optimizer side of the step on the GPU (csrc/optimizer.hip through the C ABI) against what the reference
runs on the host: torch.amp.GradScaler + nn.utils.clip_grad_* + Adam (lib/training/training.py:439-470) and
update_losses (lib/training_schemes/pcqm/tgt_training.py:141-171)."""
import math

import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _pair(precision, **cfg_kw):
    from tgt_amd.pcqm import TGT_Multi
    from tgt_amd.training.step import Trainer, StepConfig
    kwargs = gu.MODEL_CASES['multi_at_tiny'][1]
    cfg = StepConfig(num_dist_bins=24, mixed_precision=precision, coords_noise=0.0, lr_warmup_steps=10, lr_total_steps=100,
                     **cfg_kw)
    m1 = gu.fill_params(TGT_Multi(**kwargs), seed=3).cuda().eval()       # dropouts off: both sides see the same function
    m2 = gu.fill_params(TGT_Multi(**kwargs), seed=3).cuda().eval()
    return m1, m2, Trainer(m1, cfg), cfg


def _batch(cfg, step):
    from tgt_amd.training.step import preprocess_batch
    from tgt_amd.training.synthetic import make_batch
    return preprocess_batch(make_batch(3, 7, seed=40 + step, ragged=True), 'cuda', cfg, add_noise=False)


def _params(m):
    return torch.cat([p.detach().reshape(-1) for p in m.parameters()])


def test_fp16_trainer_follows_grad_scaler_semantics_with_forced_overflow():
    """init_scale 2^40 overflows the fp16 backward: both sides must skip the same steps, halve the scale in
    step, never advance Adam's step count on a skip, and (growth_interval=2) grow again afterwards"""
    from tgt_amd.training.step import pretrain_loss, lr_at
    kw = dict(init_scale=2.0 ** 40, growth_interval=2)
    m1, m2, tr, cfg = _pair('fp16', **kw)
    opt = torch.optim.Adam(m2.parameters(), lr=1.0)
    scaler = torch.amp.GradScaler('cuda', init_scale=kw['init_scale'], growth_interval=2)
    skipped_ref = 0
    scales = []
    for step in range(1, 41):
        batch = _batch(cfg, step % 4)
        tr.training_step(batch)
        for g in opt.param_groups:
            g['lr'] = lr_at(step, cfg)
        opt.zero_grad(set_to_none=True)
        with torch.autocast('cuda', dtype=torch.float16):
            loss2 = pretrain_loss(m2(batch), batch, cfg)
        scaler.scale(loss2).backward()
        before = scaler.get_scale()
        scaler.step(opt)
        scaler.update()
        skipped_ref += scaler.get_scale() < before
        st = tr.step_stats()
        scales.append(st['loss_scale'])
        assert st['loss_scale'] == scaler.get_scale(), (step, st, scaler.get_scale())
        assert st['skipped_steps'] == skipped_ref and st['applied_steps'] == step - skipped_ref
    assert skipped_ref >= 5, 'the forced overflow did not happen'           # 2^40 must back off many times
    assert max(scales[-10:]) > min(scales), 'the scale never grew back'
    some = next(iter(opt.state.values()))
    assert float(some['step']) == tr.step_stats()['applied_steps']
    assert rel(_params(m1), _params(m2)) < 2e-3
    assert torch.equal(tr.flat.shadow.float(), tr.flat.param.to(torch.float16).float())       # shadow followed every applied step


@pytest.mark.parametrize('clip', [dict(clip_grad_value=1e-3), dict(clip_grad_norm=0.05),
                                  dict(clip_grad_value=2e-3, clip_grad_norm=0.02)])
def test_gradient_clipping_matches_torch(clip):
    from tgt_amd.training.step import pretrain_loss, lr_at
    m1, m2, tr, cfg = _pair(None, **clip)
    opt = torch.optim.Adam(m2.parameters(), lr=1.0)
    for step in range(1, 4):
        batch = _batch(cfg, step)
        tr.training_step(batch)
        for g in opt.param_groups:
            g['lr'] = lr_at(step, cfg)
        opt.zero_grad(set_to_none=True)
        pretrain_loss(m2(batch), batch, cfg).backward()
        if cfg.clip_grad_value is not None:
            torch.nn.utils.clip_grad_value_(m2.parameters(), cfg.clip_grad_value)
        if cfg.clip_grad_norm is not None:
            norm = torch.nn.utils.clip_grad_norm_(m2.parameters(), cfg.clip_grad_norm)
            st = tr.step_stats()
            assert abs(st['grad_norm'] - float(norm)) < 1e-4 * float(norm)
            assert st['clip_coef'] < 1.0, 'pick a max_norm that actually clips'
        opt.step()
    assert rel(_params(m1), _params(m2)) < 1e-4
    if cfg.clip_grad_value is not None:              # the value clip was active (Adam is invariant to the norm clip's scaling)
        assert float(tr.flat.grad.abs().max()) > cfg.clip_grad_value


def test_update_losses_accumulates_like_the_reference():
    m1, _, tr, cfg = _pair('bf16')
    tr.initialize_losses()
    total, samples = 0.0, 0.0
    for step in range(1, 4):
        batch = _batch(cfg, step)
        _, loss = tr.training_step(batch)
        tr.update_losses(loss, batch)
        n = float(batch['num_nodes'].shape[0])
        total, samples = total + float(loss) * n, samples + n
    assert abs(tr.mean_loss() - total / samples) < 1e-5 * abs(total / samples)
    # NaN rule under mixed precision: skipped ... unless more than 10 arrive in a row
    nan = torch.full((), float('nan'), device='cuda', dtype=torch.float64)
    batch = _batch(cfg, 1)
    for _ in range(10):
        tr.update_losses(nan, batch)
    assert abs(tr.mean_loss() - total / samples) < 1e-5 * abs(total / samples)
    tr.update_losses(nan, batch)
    assert math.isnan(tr.mean_loss())
    tr.initialize_losses()
    assert tr.mean_loss() == 0.0


def test_resume_from_state_dict_continues_bit_identically():
    """state_dict -> new Trainer on a new model copy -> same next step, and torch.optim.Adam resumes from it too"""
    from tgt_amd.pcqm import TGT_Multi
    from tgt_amd.training.step import Trainer, pretrain_loss, lr_at
    m1, m2, tr, cfg = _pair(None)
    for step in range(1, 4):
        tr.training_step(_batch(cfg, step))
    sd, msd = tr.state_dict(), {k: v.clone() for k, v in m1.state_dict().items()}
    m3 = TGT_Multi(**gu.MODEL_CASES['multi_at_tiny'][1]).cuda().eval()
    tr3 = Trainer(m3, cfg)
    m3.load_state_dict(msd)
    tr3.load_state_dict(sd)
    m2.load_state_dict(msd)
    opt = torch.optim.Adam(m2.parameters(), lr=1.0)
    opt.load_state_dict(sd['optimizer'])
    batch = _batch(cfg, 9)
    tr.training_step(batch)
    tr3.training_step(batch)
    assert tr3.global_step == tr.global_step == 4
    assert torch.equal(_params(m1), _params(m3))
    for g in opt.param_groups:
        g['lr'] = lr_at(4, cfg)
    opt.zero_grad(set_to_none=True)
    pretrain_loss(m2(batch), batch, cfg).backward()
    opt.step()
    assert rel(_params(m1), _params(m2)) < 1e-5


def test_shadow_follows_load_state_dict():
    """the reference loads pretrained weights AFTER the trainer exists (tgt_training.py:174-189): the 16-bit
    parameter shadows the kernels read must follow without a manual refresh"""
    m1, m2, tr, cfg = _pair('bf16')
    sd = {k: v + 0.25 for k, v in m2.state_dict().items()}
    m1.load_state_dict(sd)
    assert torch.equal(tr.flat.shadow, tr.flat.param.to(torch.bfloat16))
    for p in tr.flat.params:
        assert torch.equal(p._lp, p.detach().to(torch.bfloat16))


@pytest.mark.parametrize('precision', ['bf16', 'fp16'])
def test_cached_weight_transposes_follow_every_optimizer_step(precision, monkeypatch):
    """ops.WeightTransposes: W^T of the shadowed weights for the data-gradient kernels is refreshed in one launch after each
    optimizer step (and after load_state_dict); five steps with the cache equal five steps with a fresh transpose per launch,
    bit for bit, and every cached W^T equals the transpose of its shadow at the end."""
    from tgt_amd import ops
    monkeypatch.setattr(ops, '_EDGE_MIN_ROWS', 1)             # the tiny model's edge Linears on the kernels that ask for W^T
    from tgt_amd.pcqm import TGT_Multi
    from tgt_amd.training.step import Trainer, StepConfig
    kwargs = dict(gu.FULL_AT_CFG, model_height=2)             # BASELINE widths (the kernels take K in {64,128,256}), two layers
    cfg = StepConfig(num_dist_bins=512, mixed_precision=precision, coords_noise=0.0, lr_warmup_steps=10, lr_total_steps=100)
    runs = []
    for cached in (True, False):
        monkeypatch.setattr(ops, '_WT_CACHE', cached)
        m1 = gu.fill_params(TGT_Multi(**kwargs), seed=3).cuda().eval()
        tr = Trainer(m1, cfg)
        for step in range(1, 4):
            tr.training_step(_batch(cfg, step))
        m1.load_state_dict({k: v * 1.01 for k, v in m1.state_dict().items()})      # out-of-band change: the hook refreshes
        for step in range(4, 6):
            tr.training_step(_batch(cfg, step))
        runs.append(_params(m1).clone())
        if cached:
            assert len(tr._wt.entries) > 0
            lo = tr.flat.shadow.data_ptr()
            for ptr, (t, shape) in tr._wt.entries.items():
                off = (ptr - lo) // 2
                w = tr.flat.shadow[off:off + shape[0] * shape[1]].view(shape)
                assert torch.equal(t, w.t())
        tr.close()
        assert tr._wt not in ops.WeightTransposes.live
    assert torch.equal(runs[0], runs[1])


def test_forked_parameter_gradient_stream_changes_nothing(monkeypatch):
    """ops._wgrad_fork: inside the Trainer's backward the weight gradients of the edge Linears, their bias column sums and the
    closing sums of the LayerNorm / bias partials run on a third stream that only the gradient collection joins.  Same kernels,
    same order of summation: three steps with the fork equal three steps without it, bit for bit -- and outside a Trainer's
    backward (torch.autograd.grad here) nothing is forked."""
    from tgt_amd import ops
    from tgt_amd.pcqm import TGT_Multi
    from tgt_amd.training.step import Trainer, StepConfig
    monkeypatch.setattr(ops, '_EDGE_MIN_ROWS', 1)
    monkeypatch.setattr(ops, '_SPLIT_MIN_ROWS', 1)
    kwargs = dict(gu.FULL_AT_CFG, model_height=2)
    cfg = StepConfig(num_dist_bins=512, mixed_precision='bf16', coords_noise=0.0, lr_warmup_steps=10, lr_total_steps=100)
    runs = []
    for forked in (True, False):
        monkeypatch.setattr(ops, '_WGRAD_STREAM', forked)
        m1 = gu.fill_params(TGT_Multi(**kwargs), seed=3).cuda().eval()
        with Trainer(m1, cfg) as tr:
            for step in range(1, 4):
                tr.training_step(_batch(cfg, step))
            runs.append(_params(m1).clone())
            if forked:
                assert len(ops._wgrad_streams) > 0
                x = torch.randn(300, 256, device='cuda', dtype=torch.bfloat16, requires_grad=True)
                w = torch.randn(256, 256, device='cuda', requires_grad=True)
                before = ops._trainer_backward[0]
                with torch.autocast('cuda', dtype=torch.bfloat16):
                    y = ops.linear(x, w)
                dw, = torch.autograd.grad(y.float().sum(), w)             # not the Trainer's backward: computed on this stream
                assert before == 0 and torch.isfinite(dw).all()
    assert torch.equal(runs[0], runs[1])


def test_deferred_closing_sums_change_nothing(monkeypatch):
    """ops.flush_deferred (round 4, the backward's launch diet): inside the Trainer's backward the closing sums of the split-M
    weight gradients and of the column-sum / LayerNorm partials are only registered and run as ONE tgt_sum_many launch per <= 64
    of them (+ the kernels that read them), flushed before the gradient collection reads anything.  tgt_sum_many does per item what
    tgt_sum_planes does: three steps with the deferral equal three steps without it, bit for bit -- with and without the bucketed
    gradient path (hooks read gradients in the middle of the backward), node side stream on."""
    import torch.distributed as dist
    from tgt_amd import ops
    from tgt_amd.pcqm import TGT_Multi
    from tgt_amd.training.step import Trainer, StepConfig
    monkeypatch.setattr(ops, '_EDGE_MIN_ROWS', 1)
    monkeypatch.setattr(ops, '_SPLIT_MIN_ROWS', 1)
    kwargs = dict(gu.FULL_AT_CFG, model_height=3)
    cfg = StepConfig(num_dist_bins=512, mixed_precision='bf16', coords_noise=0.0, lr_warmup_steps=10, lr_total_steps=100, bucket_mbytes=8)
    own = not dist.is_initialized()
    if own:
        dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{_free_port()}', rank=0, world_size=1)
    try:
        for bucketed in (False, True):
            runs = []
            for deferred in (True, False):
                monkeypatch.setattr(ops, '_DEFER_SUMS', deferred)
                ops._deferred_stats[:] = [0, 0]
                m1 = gu.fill_params(TGT_Multi(**kwargs), seed=3).cuda().eval()
                with Trainer(m1, cfg, force_distributed=bucketed) as tr:
                    for step in range(1, 4):
                        tr.training_step(_batch(cfg, step))
                    runs.append(_params(m1).clone())
                assert not any(q[0] or q[1] for q in ops._deferred.values())          # nothing left registered
                if deferred:
                    # the closing sums of every backward were registered (the tiny batch has no split-M weight gradients: the
                    # column-sum / LayerNorm partials only) and ran in fewer launches than there were sums
                    assert ops._deferred_stats[0] >= 3 * 3 * 3 and ops._deferred_stats[1] < ops._deferred_stats[0]
                else:
                    assert ops._deferred_stats == [0, 0]
            assert torch.equal(runs[0], runs[1]), bucketed
    finally:
        if own:
            dist.destroy_process_group()
    # outside a Trainer's backward nothing is deferred: the result is there when the call returns
    part = torch.randn(16, 256, 64, device='cuda')
    out = ops.sum_planes(part, torch.empty(256, 64, device='cuda'))
    assert torch.allclose(out, part.sum(0), atol=1e-4) and ops._deferred_stats == [0, 0]


def test_flat_gradient_destinations_change_nothing(monkeypatch):
    """ops._grad_dst (round 4): inside the Trainer's backward the kernels that end a weight gradient write it into the parameter's
    slice of the flat gradient buffer; AccumulateGrad adopts the alias and the gradient collection skips the copy.  Same kernels,
    same order: three steps with the destinations equal three steps without, bit for bit, with and without the bucketed path
    (whose hooks collect in the middle of the backward); and most of the gradient bytes are in place when the backward ends."""
    import torch.distributed as dist
    from tgt_amd import ops
    from tgt_amd.pcqm import TGT_Multi
    from tgt_amd.training.step import Trainer, StepConfig
    monkeypatch.setattr(ops, '_EDGE_MIN_ROWS', 1)
    monkeypatch.setattr(ops, '_SPLIT_MIN_ROWS', 1)
    kwargs = dict(gu.FULL_AT_CFG, model_height=3)
    cfg = StepConfig(num_dist_bins=512, mixed_precision='bf16', coords_noise=0.0, lr_warmup_steps=10, lr_total_steps=100, bucket_mbytes=8)
    own = not dist.is_initialized()
    if own:
        dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{_free_port()}', rank=0, world_size=1)
    try:
        for bucketed in (False, True):
            runs = []
            for in_place in (True, False):
                monkeypatch.setattr(ops, '_FLAT_GRAD_ON', in_place)
                m1 = gu.fill_params(TGT_Multi(**kwargs), seed=3).cuda().eval()
                with Trainer(m1, cfg, force_distributed=bucketed) as tr:
                    for step in range(1, 4):
                        tr.training_step(_batch(cfg, step))
                    placed = sum(p.numel() for p, v in zip(tr.flat.params, tr.flat.grad_views)
                                 if p.grad is not None and p.grad.data_ptr() == v.data_ptr())
                    total = sum(p.numel() for p in tr.flat.params if p.grad is not None)
                    runs.append(_params(m1).clone())
                    mine = [p.data_ptr() for p in tr.flat.params]
                    assert all(q in ops._FLAT_GRAD for q in mine)
                assert not any(q in ops._FLAT_GRAD for q in mine)          # a closed trainer takes its slices out of the registry
                if in_place:
                    assert placed > 0.5 * total, (placed, total)          # the large matrices; biases / LayerNorm vectors still travel
                else:
                    assert placed == 0
            assert torch.equal(runs[0], runs[1]), bucketed
    finally:
        if own:
            dist.destroy_process_group()
    # outside a Trainer's backward gradients are ordinary tensors
    m2 = gu.fill_params(TGT_Multi(**kwargs), seed=3).cuda().eval()
    with Trainer(m2, cfg) as tr:
        assert ops._grad_dst(tr.flat.params[0].data_ptr(), tr.flat.params[0].shape, torch.float32) is None


def test_graphed_training_step_equals_eager():
    """training/graphed.py: Trainer.training_step captured in a hipGraph and replayed.  What a replay must not inherit from the
    capture is a device value -- the dropout step counter the kernels mix into their seeds (tgt_set_seed_counter), torch's
    capture-aware generator for the DropPath / source-dropout draws, the learning rate in the optimizer's control block -- so in
    graph-safe mode an eager step and a replay launch the same kernels with the same arguments: after three warm-up steps and
    three more steps on fresh batches the parameters are bit-identical, dropouts ON; and successive replays do not repeat their
    drop patterns."""
    from tgt_amd import ops
    from tgt_amd.pcqm import TGT_Multi
    from tgt_amd.training.step import Trainer, StepConfig
    from tgt_amd.training.graphed import GraphedTrainingStep
    _graphed_equals_eager(bucketed=False)


def test_graphed_training_step_fp16_loss_scaling_and_fp32():
    """the device-side GradScaler (found_inf / skip / scale update in the control block) replays with the step; fp32 likewise"""
    _graphed_equals_eager(bucketed=False, precision='fp16')
    _graphed_equals_eager(bucketed=False, precision=None)


def test_graphed_training_step_captures_the_bucketed_all_reduce():
    """the same equality with the distributed path on (nccl backend, world size 1, small buckets): the gradient hooks run at capture
    time, their RCCL all-reduces on the high-priority stream become nodes of the graph"""
    import torch.distributed as dist
    own = not dist.is_initialized()
    if own:
        dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{_free_port()}', rank=0, world_size=1)
    try:
        _graphed_equals_eager(bucketed=True)
    finally:
        if own:
            dist.destroy_process_group()


def _graphed_equals_eager(bucketed, precision='bf16'):
    from tgt_amd import ops
    from tgt_amd.pcqm import TGT_Multi
    from tgt_amd.training.step import Trainer, StepConfig
    from tgt_amd.training.graphed import GraphedTrainingStep
    kwargs = dict(gu.FULL_AT_CFG, model_height=2, source_dropout=0.3, drop_path=0.2, node_act_dropout=0.1, edge_act_dropout=0.1)
    cfg = StepConfig(num_dist_bins=512, mixed_precision=precision, coords_noise=0.0, lr_warmup_steps=4, lr_total_steps=100, bucket_mbytes=8)
    batches = [_batch(cfg, s) for s in range(4)]
    runs, losses = [], []
    try:
        for graphed in (False, True):
            torch.manual_seed(77)
            ops.reset_random_pools()
            m = gu.fill_params(TGT_Multi(**kwargs), seed=3).cuda().train()
            with Trainer(m, cfg, force_distributed=bucketed) as tr:
                assert tr.distributed == bucketed
                if graphed:
                    if bucketed:
                        import pytest
                        with pytest.raises(RuntimeError, match='single-rank'):
                            GraphedTrainingStep(tr, batches[0], warmup=1)
                    with GraphedTrainingStep(tr, batches[0], warmup=3, allow_distributed=bucketed) as gs:
                        for b in batches[1:]:
                            out, loss = gs.step(b)
                            losses.append(float(loss))
                        # the same batch twice more: a new counter value, a new DropPath draw -> another loss
                        again = [float(gs.step(batches[1])[1]) for _ in range(2)]
                        assert gs.replays == 5 and int(gs.counter) == 3 + 5
                else:
                    ctr = torch.zeros(1, dtype=torch.int64, device='cuda')
                    ops.graph_safe_rng(True)
                    ops.set_seed_counter(ctr)
                    tr.set_device_lr(True)
                    for b in [batches[0]] * 3 + batches[1:] + [batches[1]] * 2:
                        tr.global_step += 1
                        tr.write_device_lr()
                        ctr.add_(1)
                        tr.compute_gradients(b)
                        tr.apply_gradients()
                    ops.set_seed_counter(None)
                    ops.graph_safe_rng(False)
                runs.append(_params(m).clone())
                steps = tr.global_step
            assert steps == 8
    finally:
        ops.set_seed_counter(None)
        ops.graph_safe_rng(False)
    assert torch.isfinite(runs[0]).all()
    assert torch.equal(runs[0], runs[1])
    assert again[0] != again[1] and losses[0] != again[0]          # replays of one batch differ: the drop patterns moved on


def test_graphed_training_step_refuses_what_it_cannot_replay():
    """one graph per batch shape; attention dropout inside the attention kernels is host-seeded: both are errors, not silent
    repetitions; close() hands the process back to host seeds and pooled draws"""
    import pytest
    from tgt_amd import ops
    from tgt_amd.pcqm import TGT_Multi
    from tgt_amd.training.step import Trainer, StepConfig, preprocess_batch
    from tgt_amd.training.synthetic import make_batch
    from tgt_amd.training.graphed import GraphedTrainingStep
    cfg = StepConfig(num_dist_bins=512, mixed_precision='bf16', coords_noise=0.0, lr_warmup_steps=4, lr_total_steps=100)
    kwargs = dict(gu.FULL_AT_CFG, model_height=2, node_act_dropout=0.1)
    m = gu.fill_params(TGT_Multi(**kwargs), seed=3).cuda().train()
    with Trainer(m, cfg) as tr:
        with GraphedTrainingStep(tr, _batch(cfg, 0), warmup=1) as gs:
            assert ops._GRAPH_SAFE[0] and ops._seed_counter[0] is gs.counter and tr.device_lr
            other = preprocess_batch(make_batch(2, 7, seed=1, ragged=True), 'cuda', cfg, add_noise=False)      # another batch size
            with pytest.raises(RuntimeError, match='one graph per shape'):
                gs.step(other)
            gs.step(_batch(cfg, 1))
        assert not ops._GRAPH_SAFE[0] and ops._seed_counter[0] is None and not tr.device_lr
        with pytest.raises(RuntimeError, match='closed'):
            gs.step(_batch(cfg, 1))
        tr.training_step(_batch(cfg, 2))                     # the eager trainer goes on (learning rate as an argument again)
    m2 = gu.fill_params(TGT_Multi(**dict(kwargs, triplet_dropout=0.1)), seed=3).cuda().train()
    with Trainer(m2, cfg) as tr2:
        with pytest.raises(RuntimeError, match='attention dropout'):
            GraphedTrainingStep(tr2, _batch(cfg, 0), warmup=1)
    assert not ops._GRAPH_SAFE[0]


def test_graph_mode_is_given_back_when_capture_fails_or_the_owner_is_dropped(monkeypatch):
    """GraphedTrainingStep / GraphedStepCache switch process-global state (graph-safe randomness, the device
    seed counter) and Trainer state (device learning rate).  A capture that raises, or an owner that is garbage-collected without
    close(), must hand all of it back -- a Trainer silently left in that mode repeats dropout patterns on a stale learning rate."""
    import gc
    import pytest
    from tgt_amd import ops
    from tgt_amd.pcqm import TGT_Multi
    from tgt_amd.training.step import Trainer, StepConfig
    from tgt_amd.training import graphed
    cfg = StepConfig(num_dist_bins=512, mixed_precision='bf16', coords_noise=0.0, lr_warmup_steps=4, lr_total_steps=100)
    kwargs = dict(gu.FULL_AT_CFG, model_height=2, node_act_dropout=0.1)
    m = gu.fill_params(TGT_Multi(**kwargs), seed=3).cuda().train()
    with Trainer(m, cfg) as tr:
        calls = [0]
        real = graphed.GraphedTrainingStep._body

        def failing(self):
            calls[0] += 1
            if calls[0] == 2:                         # the warm-up step passes, the capture raises
                raise RuntimeError('boom inside the capture')
            return real(self)
        monkeypatch.setattr(graphed.GraphedTrainingStep, '_body', failing)
        with pytest.raises(RuntimeError, match='boom'):
            graphed.GraphedTrainingStep(tr, _batch(cfg, 0), warmup=1)
        assert not ops._GRAPH_SAFE[0] and ops._seed_counter[0] is None and not tr.device_lr
        monkeypatch.setattr(graphed.GraphedTrainingStep, '_body', real)
        torch.cuda.synchronize()
        tr.training_step(_batch(cfg, 1))              # the eager trainer goes on
        gs = graphed.GraphedTrainingStep(tr, _batch(cfg, 0), warmup=1)
        assert ops._GRAPH_SAFE[0] and tr.device_lr
        del gs                                        # dropped without close()
        gc.collect()
        assert not ops._GRAPH_SAFE[0] and ops._seed_counter[0] is None and not tr.device_lr
        cache = graphed.GraphedStepCache(tr, warmup=1, max_graphs=1)
        assert ops._GRAPH_SAFE[0]
        del cache
        gc.collect()
        assert not ops._GRAPH_SAFE[0] and not tr.device_lr
        # an un-closed owner REBOUND by a new one -- the old object's finalizer runs after the new capture finished and
        # must not switch the mode off under the live owner (ownership token = the registered seed counter)
        gs = graphed.GraphedTrainingStep(tr, _batch(cfg, 0), warmup=1)
        gs = graphed.GraphedTrainingStep(tr, _batch(cfg, 0), warmup=1)     # noqa: F841  (the first owner is dropped here)
        gc.collect()
        assert ops._GRAPH_SAFE[0] and tr.device_lr and ops._seed_counter[0] is gs.counter
        gs.close()
        assert not ops._GRAPH_SAFE[0] and ops._seed_counter[0] is None and not tr.device_lr


def test_eager_step_on_a_graph_owned_trainer_advances_rate_and_counter():
    """ADVICE r4 (medium), second half: Trainer.training_step called directly while a graph owns the trainer (the odd-shaped
    last batch the one-graph-per-shape owner refuses) does what a replay does around the body -- refreshes the device learning
    rate and bumps the dropout counter -- and equals eager_graph_safe_step bit for bit."""
    from tgt_amd import ops
    from tgt_amd.pcqm import TGT_Multi
    from tgt_amd.training.step import Trainer, StepConfig, lr_at
    from tgt_amd.training.graphed import GraphedTrainingStep, eager_graph_safe_step
    kwargs = dict(gu.FULL_AT_CFG, model_height=2, source_dropout=0.3, drop_path=0.2, node_act_dropout=0.1, edge_act_dropout=0.1)
    cfg = StepConfig(num_dist_bins=512, mixed_precision='bf16', coords_noise=0.0, lr_warmup_steps=4, lr_total_steps=100)
    runs = []
    for direct in (True, False):
        torch.manual_seed(79)
        ops.reset_random_pools()
        m = gu.fill_params(TGT_Multi(**kwargs), seed=3).cuda().train()
        with Trainer(m, cfg) as tr:
            with GraphedTrainingStep(tr, _batch(cfg, 0), warmup=1) as gs:
                gs.step(_batch(cfg, 1))
                c0 = int(gs.counter)
                if direct:
                    tr.training_step(_batch(cfg, 2))
                else:
                    eager_graph_safe_step(tr, gs.counter, _batch(cfg, 2))
                assert int(gs.counter) == c0 + 1
                assert abs(float(tr.ctl[ops.CTL_LR]) - lr_at(tr.global_step, cfg)) < 1e-12 + 1e-6 * lr_at(tr.global_step, cfg)
                gs.step(_batch(cfg, 3))
            torch.cuda.synchronize()
            runs.append(tr.flat.param.clone())
    assert torch.equal(runs[0], runs[1])


def test_graphed_step_cache_stops_capturing_when_it_thrashes():
    """more live batch shapes than max_graphs would evict and re-capture on every step; the cache notices that
    captures outnumber replays, warns once and runs the remaining steps eagerly in graph-safe mode (same arithmetic).  A
    Python-valued batch entry is part of the key."""
    import pytest
    from tgt_amd import ops
    from tgt_amd.pcqm import TGT_Multi
    from tgt_amd.training.step import Trainer, StepConfig, preprocess_batch
    from tgt_amd.training.synthetic import make_batch
    from tgt_amd.training.graphed import GraphedStepCache
    kwargs = dict(gu.FULL_AT_CFG, model_height=2, node_act_dropout=0.1)
    cfg = StepConfig(num_dist_bins=512, mixed_precision='bf16', coords_noise=0.0, lr_warmup_steps=4, lr_total_steps=100)
    shapes = [(2, 5), (2, 6), (2, 7)]
    m = gu.fill_params(TGT_Multi(**kwargs), seed=3).cuda().train()
    with Trainer(m, cfg) as tr:
        with GraphedStepCache(tr, warmup=1, max_graphs=1) as cache:
            assert cache._key({'a': torch.zeros(2), 'flag': 1}) != cache._key({'a': torch.zeros(2), 'flag': 2})
            cache.thrash_window = 6
            with pytest.warns(RuntimeWarning, match='falling back to eager'):
                for i in range(24):
                    b = preprocess_batch(make_batch(*shapes[i % 3], seed=60 + i, ragged=True), 'cuda', cfg, add_noise=False)
                    cache.step(b)
            assert cache.eager_fallback and not cache.graphs and cache.captures <= 8
            assert int(cache.counter) == 24              # every step() was exactly one optimizer step
    assert not ops._GRAPH_SAFE[0]


def test_graphed_step_cache_one_graph_per_shape():
    """GraphedStepCache: batches of two shapes, interleaved; every step() is exactly one optimizer step (the first `warmup` of a
    shape eagerly in graph-safe mode, then a captured replay), one dropout counter for all graphs, least recently used evicted:
    the parameters equal those of the same batch sequence stepped eagerly in graph-safe mode, bit for bit"""
    from tgt_amd import ops
    from tgt_amd.pcqm import TGT_Multi
    from tgt_amd.training.step import Trainer, StepConfig, preprocess_batch
    from tgt_amd.training.synthetic import make_batch
    from tgt_amd.training.graphed import GraphedStepCache, eager_graph_safe_step
    kwargs = dict(gu.FULL_AT_CFG, model_height=2, source_dropout=0.3, drop_path=0.2, node_act_dropout=0.1, edge_act_dropout=0.1)
    cfg = StepConfig(num_dist_bins=512, mixed_precision='bf16', coords_noise=0.0, lr_warmup_steps=4, lr_total_steps=100)
    shapes = [(3, 7), (2, 9), (4, 5)]
    seq = [0, 1, 0, 0, 1, 1, 2, 0, 2, 2, 1, 0]            # with max_graphs = 2 the third shape evicts the least recently used one
    batches = [preprocess_batch(make_batch(*shapes[k], seed=60 + i, ragged=True), 'cuda', cfg, add_noise=False) for i, k in enumerate(seq)]
    runs = []
    try:
        for cached in (False, True):
            torch.manual_seed(78)
            ops.reset_random_pools()
            m = gu.fill_params(TGT_Multi(**kwargs), seed=3).cuda().train()
            with Trainer(m, cfg) as tr:
                if cached:
                    with GraphedStepCache(tr, warmup=1, max_graphs=2) as cache:
                        for b in batches:
                            cache.step(b)
                        assert cache.captures >= 3 and cache.evictions >= 1 and len(cache.graphs) <= 2
                        assert int(cache.counter) == len(seq)
                else:
                    ctr = torch.zeros(1, dtype=torch.int64, device='cuda')
                    ops.graph_safe_rng(True)
                    ops.set_seed_counter(ctr)
                    tr.set_device_lr(True)
                    for b in batches:
                        eager_graph_safe_step(tr, ctr, b)
                    ops.set_seed_counter(None)
                    ops.graph_safe_rng(False)
                assert tr.global_step == len(seq)
                runs.append(_params(m).clone())
    finally:
        ops.set_seed_counter(None)
        ops.graph_safe_rng(False)
    assert torch.isfinite(runs[0]).all() and torch.equal(runs[0], runs[1])


# ---- world-size 2 on the GPU: two ranks share cuda:0 and exchange over gloo (device tensors staged through the host by
# the backend).  RCCL refuses two ranks on one device and the test boxes have one GPU, so this is the closest a 1-GPU box gets
# to the N > 1 path: autograd hooks -> bucket gather behind BOTH streams -> asynchronous all-reduce -> one-launch Adam, on
# the HIP kernels, with the node side stream on.
def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _world2_worker(rank, world, port, out_dir, side_stream, backend='gloo', exchange='all_reduce', shard=False, tag=''):
    import os
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')        # (dmabuf IPC: what the host driver supports)
    if backend == 'nccl':                                           # RCCL: one device per rank, as bench.py / the reference run
        torch.cuda.set_device(rank)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', rank))
    else:
        torch.cuda.set_device(0)
        dist.init_process_group('gloo', rank=rank, world_size=world)
    from tgt_amd import ops
    from tgt_amd.pcqm import TGT_Multi
    from tgt_amd.training.step import Trainer, StepConfig, preprocess_batch
    from tgt_amd.training.synthetic import make_batch
    ops.side_stream.enabled = side_stream
    kwargs = dict(gu.MODEL_CASES['multi_at_tiny'][1])
    kwargs.update(model_height=4)
    cfg = StepConfig(num_dist_bins=24, mixed_precision='bf16', coords_noise=0.0, bucket_mbytes=0.05,
                     lr_warmup_steps=10, lr_total_steps=100, grad_exchange=exchange, shard_optimizer=shard)
    model = gu.fill_params(TGT_Multi(**kwargs), seed=5 + rank).cuda().eval()       # ranks start DIFFERENT: the broadcast fixes it
    tr = Trainer(model, cfg)
    assert tr.distributed and tr.world == 2 and tr.buckets is not None and len(tr.buckets) > 4 and tr.sharded == shard
    grads = []
    for step in range(3):
        full = make_batch(8, 12, seed=90 + step, ragged=False)
        part = {k: v[4 * rank:4 * rank + 4] for k, v in full.items()}
        batch = preprocess_batch(part, 'cuda', cfg, add_noise=False)
        tr.global_step += 1
        tr.compute_gradients(batch)
        grads.append((tr.flat.grad / world).cpu())
        tr.apply_gradients()
    torch.cuda.synchronize()
    extra = {}
    if shard:
        import pytest as _pt
        with _pt.raises(RuntimeError, match='consolidate_optimizer_state'):
            tr.optimizer_state_dict()
        tr.consolidate_optimizer_state()                            # (a collective: both ranks)
    extra = {'exp_avg': tr.flat.exp_avg.cpu(), 'exp_avg_sq': tr.flat.exp_avg_sq.cpu(), 'shadow': tr.flat.shadow.float().cpu(),
             'opt_steps': tr.optimizer_state_dict()['state'][0]['step']}
    torch.save({'grads': grads, 'param': tr.flat.param.cpu(), **extra}, os.path.join(out_dir, f'rank{rank}{tag}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def _check_world2_against_single_rank(tmp_path, grads_replicated=True):
    from tgt_amd.pcqm import TGT_Multi
    from tgt_amd.training.step import Trainer, StepConfig, preprocess_batch
    from tgt_amd.training.synthetic import make_batch
    r0 = torch.load(tmp_path / 'rank0.pt')
    r1 = torch.load(tmp_path / 'rank1.pt')
    assert torch.equal(r0['param'], r1['param'])                # replicas stay bit-identical
    if grads_replicated:                                        # (sharded optimizer: a rank holds the reduced gradient on its slices only)
        for g0, g1 in zip(r0['grads'], r1['grads']):
            assert torch.equal(g0, g1)

    kwargs = dict(gu.MODEL_CASES['multi_at_tiny'][1])
    kwargs.update(model_height=4)
    cfg = StepConfig(num_dist_bins=24, mixed_precision='bf16', coords_noise=0.0, lr_warmup_steps=10, lr_total_steps=100)
    model = gu.fill_params(TGT_Multi(**kwargs), seed=5).cuda().eval()
    tr = Trainer(model, cfg)
    for step in range(3):
        batch = preprocess_batch(make_batch(8, 12, seed=90 + step, ragged=False), 'cuda', cfg, add_noise=False)
        tr.global_step += 1
        tr.compute_gradients(batch)
        # the loss is a mean over the rank's graphs / pairs: halves average to the whole when the halves weigh the same
        if grads_replicated:
            assert rel(r0['grads'][step], tr.flat.grad) < 2e-2, (step, rel(r0['grads'][step], tr.flat.grad))
        tr.apply_gradients()
    assert rel(r0['param'], tr.flat.param) < 1e-3


@pytest.mark.parametrize('side_stream', [True, False])
def test_world2_on_one_gpu_matches_single_rank(tmp_path, side_stream):
    """two ranks on disjoint half-batches == one rank on the whole batch (per-graph-mean losses), and both ranks
    hold the same parameters after three optimizer steps.  Two ranks SHARE cuda:0 and exchange device tensors over gloo (RCCL
    refuses two ranks on one device); the RCCL form of the same test follows."""
    import torch.multiprocessing as mp
    mp.spawn(_world2_worker, args=(2, _free_port(), str(tmp_path), side_stream), nprocs=2, join=True)
    _check_world2_against_single_rank(tmp_path)


def test_sharded_optimizer_equals_the_replicated_step_bit_for_bit(tmp_path):
    """StepConfig.shard_optimizer: reduce-scatter only, Adam on each rank's 1/world slice of every bucket, all-gather
    of the updated parameters.  Two ranks sharing cuda:0 over gloo, three steps: parameters, the 16-bit shadow and (after the
    collective consolidate_optimizer_state) both Adam moments equal the replicated reduce_scatter run BIT FOR BIT on both ranks."""
    import torch.multiprocessing as mp
    mp.spawn(_world2_worker, args=(2, _free_port(), str(tmp_path), True, 'gloo', 'reduce_scatter', False, '_rep'), nprocs=2, join=True)
    mp.spawn(_world2_worker, args=(2, _free_port(), str(tmp_path), True, 'gloo', 'reduce_scatter', True, '_shard'), nprocs=2, join=True)
    rep = torch.load(tmp_path / 'rank0_rep.pt')
    for r in (0, 1):
        sh = torch.load(tmp_path / f'rank{r}_shard.pt')
        for k in ('param', 'shadow', 'exp_avg', 'exp_avg_sq'):
            assert torch.equal(sh[k], rep[k]), (r, k)
        assert float(sh['opt_steps']) == float(rep['opt_steps']) == 3.0


@pytest.mark.parametrize('exchange', ['all_reduce', 'reduce_scatter', 'reduce_scatter_sharded'])
def test_world2_over_rccl_matches_single_rank(tmp_path, exchange):
    """BASELINE config 3's exchange on real links: two ranks, one MI355X each, backend 'nccl' (= RCCL over xGMI) -- bucketed
    all-reduce from the gradient hooks (and the reduce-scatter + all-gather option), rank-0 broadcast, packed loss all-reduce.
    Skips on a box with one GPU (the 1-GPU test boxes); runs wherever torch sees two."""
    if torch.cuda.device_count() < 2:
        pytest.skip(f'needs 2 GPUs for two RCCL ranks, this box has {torch.cuda.device_count()}')
    import torch.multiprocessing as mp
    shard = exchange.endswith('_sharded')
    mp.spawn(_world2_worker, args=(2, _free_port(), str(tmp_path), True, 'nccl', exchange.replace('_sharded', ''), shard), nprocs=2, join=True)
    _check_world2_against_single_rank(tmp_path, grads_replicated=not shard)
