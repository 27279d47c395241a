"""Host logic of tgt_amd.training.step.Trainer that needs no GPU: checkpoint/resume state in the formats the
reference saves (lib/training/training.py:290-360: optimizer_state.pt = optimizer.state_dict(),
grad_scaler_state.pt = GradScaler.state_dict(), training_state.pt = {global_step, ...})."""
import copy

import torch

from tgt_amd.training.step import Trainer, StepConfig, FlatState


def _model(seed=0):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.GELU(), torch.nn.Linear(7, 3))


def _loss(outputs, batch, cfg):
    return (outputs - batch['y']).pow(2).mean()


def test_optimizer_state_dict_loads_into_torch_adam_and_back():
    m = _model()
    tr = Trainer(m, StepConfig(mixed_precision=None), loss_fn=_loss)
    g = torch.Generator().manual_seed(1)
    tr.flat.exp_avg.copy_(torch.randn(tr.flat.numel, generator=g))
    tr.flat.exp_avg_sq.copy_(torch.rand(tr.flat.numel, generator=g))
    tr._applied_steps, tr.global_step = 7, 9
    sd = tr.state_dict()
    assert sd['training_state'] == {'global_step': 9} and sd['grad_scaler'] == {}
    # torch.optim.Adam takes it as is (parameter order = model.parameters())
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    opt.load_state_dict(copy.deepcopy(sd['optimizer']))
    for i, p in enumerate(m.parameters()):
        st = opt.state[p]
        assert float(st['step']) == 7
        assert torch.equal(st['exp_avg'], tr.flat.views(tr.flat.exp_avg)[i])
        assert torch.equal(st['exp_avg_sq'], tr.flat.views(tr.flat.exp_avg_sq)[i])
    # ... and torch.optim.Adam's own state_dict loads back into a fresh Trainer
    tr2 = Trainer(_model(1), StepConfig(mixed_precision=None), loss_fn=_loss)
    tr2.load_state_dict(dict(training_state=dict(global_step=9), optimizer=opt.state_dict()))
    assert tr2.global_step == 9 and tr2._applied_steps == 7 and tr2.step_stats()['applied_steps'] == 7
    for a, b in zip(tr.flat.views(tr.flat.exp_avg) + tr.flat.views(tr.flat.exp_avg_sq),
                    tr2.flat.views(tr2.flat.exp_avg) + tr2.flat.views(tr2.flat.exp_avg_sq)):
        assert torch.equal(a, b)


def test_apex_style_group_step_is_understood():
    tr = Trainer(_model(), StepConfig(mixed_precision=None), loss_fn=_loss)
    n = len(tr.flat.params)
    sd = dict(state={i: dict(exp_avg=torch.zeros_like(p), exp_avg_sq=torch.ones_like(p)) for i, p in enumerate(tr.flat.params)},
              param_groups=[dict(step=123, params=list(range(n)))])
    tr.load_optimizer_state_dict(sd)
    assert tr._applied_steps == 123
    assert float(tr.flat.exp_avg_sq.sum()) == sum(p.numel() for p in tr.flat.params)


def test_grad_scaler_state_dict_layout_matches_torch():
    tr = Trainer(_model(), StepConfig(mixed_precision='fp16', growth_interval=50), loss_fn=_loss)
    sd = tr.grad_scaler_state_dict()
    ref = torch.amp.GradScaler('cpu', enabled=True, growth_interval=50).state_dict()
    assert set(sd) == set(ref), (sd, ref)
    assert sd['scale'] == 65536.0 and sd['growth_interval'] == 50 and sd['_growth_tracker'] == 0
    tr.load_grad_scaler_state_dict(dict(scale=1024.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=7, _growth_tracker=3))
    st = tr.step_stats()
    assert st['loss_scale'] == 1024.0 and st['growth_tracker'] == 3 and tr.cfg.growth_interval == 7


def test_new_flat_state_drops_stale_shadows():
    m = _model()
    for p in m.parameters():
        p._lp = p.detach().to(torch.bfloat16)          # what an earlier mixed-precision Trainer left behind
    FlatState(m)
    assert not any(hasattr(p, '_lp') for p in m.parameters())


def test_trainer_close_releases_the_side_stream_and_the_hook():
    """A dropped or closed Trainer must give the node side stream back (ops.side_stream is only safe while a Trainer orders the
    gradient collection after both streams; torch DDP's reducer does not) and must not stay alive through its
    load_state_dict hook."""
    import gc
    import weakref
    from tgt_amd import ops
    base = ops.side_stream._owners
    m = _model()
    hooks0 = len(m._load_state_dict_post_hooks)
    tr = Trainer(m, StepConfig(mixed_precision=None), loss_fn=_loss)
    assert ops.side_stream._owners == base + 1 and len(m._load_state_dict_post_hooks) == hooks0 + 1
    tr.close()
    tr.close()                                       # idempotent
    assert ops.side_stream._owners == base and len(m._load_state_dict_post_hooks) == hooks0
    m.load_state_dict(m.state_dict())                # no stale hook fires
    # garbage collection alone does the same, and the hook does not keep the trainer (and its flat buffers) alive
    tr2 = Trainer(m, StepConfig(mixed_precision=None), loss_fn=_loss)
    ref = weakref.ref(tr2)
    assert ops.side_stream._owners == base + 1
    del tr2
    gc.collect()
    assert ref() is None
    assert ops.side_stream._owners == base and len(m._load_state_dict_post_hooks) == hooks0
    with Trainer(m, StepConfig(mixed_precision=None), loss_fn=_loss):
        assert ops.side_stream._owners == base + 1
    assert ops.side_stream._owners == base
