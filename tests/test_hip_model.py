"""Module- and model-level parity of the HIP-backed mirror (tgt_amd.tgt,
tgt_amd.pcqm) against the oracle and the committed golden vectors, on the GPU."""
import os

import numpy as np
import pytest
import torch

import golden_util as gu
import parity_log
from oracle import core, modules as om

pytestmark = pytest.mark.gpu


# fp32 model-level bar (24 layers, forward AND parameter gradients, against the reference's own fp32 numbers): measured maximum
# 1.05e-6 over 40 comparisons (profiles/parity_errors.json, `fp32-model`; GEMM selection pinned by tests/conftest.py); stated 5x that.
# (Rounds 1-5 ran these with 2e-3 / 5e-3: a 1000x regression of an fp32 24L gradient would have passed.)
FP32_MODEL_TOL = 5e-6


def tag(mode):
    """files the comparisons that follow under `<mode>-model` in the parity log (TGT_PARITY_LOG, tests/parity_log.py)"""
    parity_log.Tol.last = f'{mode}-model'


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return parity_log.record(float((a - b).norm() / (b.norm() + 1e-30)))


def load_gold(name):
    z = np.load(os.path.join(gu.GOLDEN_DIR, name + '.npz'))
    return {k.split('::')[0]: z[k] for k in z.files if k.endswith('::full')}


def product_class(name):
    from tgt_amd.tgt.layers import blocks as L, triplet as T
    from tgt_amd import pcqm
    for mod in (L, T, pcqm):
        if hasattr(mod, name):
            return getattr(mod, name)
    raise KeyError(name)


SMALL_OPS = [k for k, v in gu.OP_CASES.items() if v[2] is gu.SMALL]


@pytest.mark.parametrize('name', SMALL_OPS)
def test_module_matches_reference_golden_fp32(name):
    """HIP-backed module in fp32 vs the reference's own float64 outputs/gradients."""
    cls_name, kwargs, geom = gu.OP_CASES[name]
    i = list(gu.OP_CASES).index(name)
    mod = gu.fill_params(product_class(cls_name)(**kwargs), seed=100 + i).cuda().eval()
    h, e, mask, gh, ge = (t.float().cuda() for t in gu.op_inputs(geom['B'], geom['N'], geom['W'], geom['C'],
                                                               geom['num_nodes'], 100 + i + 1))
    # finfo(float64).min would overflow to -inf in float32: build the mask as the model does, in fp32
    mask = gu.additive_mask(geom['num_nodes'], geom['N'], torch.float32).cuda()
    h.requires_grad_(True)
    e.requires_grad_(True)
    gold = load_gold('op_' + name)
    res = {}
    if cls_name in gu.NODE_OPS:
        ho, eo = mod(h, e, mask)
        loss = 0.
        if ho is not h:
            res['out_h'] = ho
            loss = loss + (ho * gh).sum()
        if eo is not e:
            res['out_e'] = eo
            loss = loss + (eo * ge).sum()
    elif cls_name == 'FFN':
        res['out_e'] = mod(e)
        loss = (res['out_e'] * ge).sum()
    else:
        res['out_e'] = mod(e, mask)
        loss = (res['out_e'] * ge).sum()
    loss.backward()
    if h.grad is not None:
        res['grad_h'] = h.grad
    if e.grad is not None:
        res['grad_e'] = e.grad
    for k, p in mod.named_parameters():
        if p.grad is not None:
            res['pgrad.' + k] = p.grad
    assert set(res) == set(gold)
    for k, t in res.items():
        g = torch.from_numpy(gold[k])
        if g.abs().max() < 1e-12:
            assert t.abs().max() < 1e-4, k
        else:
            assert rel(t, g) < 2e-4, (k, rel(t, g))


@pytest.mark.parametrize('name', list(gu.MODEL_CASES))
@pytest.mark.parametrize('mode', ['fp32', 'bf16'])
def test_task_model_matches_reference_golden(name, mode):
    from tgt_amd.training.step import pretrain_loss, binned_distance_loss, coords2dist, StepConfig
    cls_name, kwargs, geom = gu.MODEL_CASES[name]
    i = list(gu.MODEL_CASES).index(name)
    model = gu.fill_params(product_class(cls_name)(**kwargs), seed=500 + i).cuda().train()
    batch = {k: v.cuda() for k, v in gu.model_batch(geom, seed=600 + i).items()}
    gold = load_gold('model_' + name)
    cfg = StepConfig(num_dist_bins=kwargs.get('num_dist_bins', 0), range_dist_bins=8, dist_loss_weight=0.1)
    ctx = torch.autocast('cuda', dtype=torch.bfloat16) if mode == 'bf16' else torch.autocast('cuda', enabled=False)
    with ctx:
        out = model(batch)
        if cls_name == 'TGT_Multi':
            res = dict(gap=out[0], logits=out[1])
            loss = pretrain_loss(out, batch, cfg)
        elif cls_name == 'TGT_Distance':
            res = dict(logits=out)
            loss = binned_distance_loss(out, coords2dist(batch['dft_coords']), batch['edge_mask'], cfg.num_dist_bins, 8)
        else:
            res = dict(gap=out)
            loss = torch.nn.functional.l1_loss(out, batch['target'])
    assert loss.dtype == torch.float64 or cls_name == 'TGT_Distance'
    res['loss'] = loss
    loss.backward()
    named = dict(model.named_parameters())
    for k in gu.GRAD_PROBE_KEYS:
        if k in named and named[k].grad is not None:
            res['pgrad.' + k] = named[k].grad
    assert set(res) == set(gold)
    # fp32 HIP path vs the fp64 reference: 3e-4 outputs, 2e-3 gradients.
    # bf16: 2x the REFERENCE's own bf16-autocast drift on this very case, tensor by tensor (tests/golden/bf16_drift.npz,
    # written by tools/make_golden.py from the reference; SURVEY 8c "grads <= 2x"): 1.2e-2 .. 2.9e-2 for the probed
    # gradients and the logits.  `gap` (B numbers) and `loss` (one number) are too few values for a relative L2 to be
    # stable at that level: they keep absolute floors of 3e-2 / 1e-3.
    drift = gu.bf16_drift(name)
    floors = dict(gap=3e-2, loss=1e-3)
    for k, t in res.items():
        g = torch.from_numpy(gold[k])
        if mode == 'fp32':
            tol = 2e-3 if k.startswith('pgrad.') else 3e-4
        else:
            tol = max(2.0 * drift[k], floors.get(k, 0.0))
        assert rel(t, g) < tol, (k, rel(t, g), tol)


def test_full_width_24L_forward_vs_reference_golden():
    """TGT-At 24L at BASELINE widths: eval forward vs the reference's fp32 CPU forward."""
    from tgt_amd.pcqm import TGT_Multi
    geom = dict(B=2, N=12, num_nodes=[12, 9])
    model = gu.fill_params(TGT_Multi(**gu.FULL_AT_CFG), seed=900).cuda().eval()
    batch = {k: v.cuda() for k, v in gu.model_batch(geom, seed=901).items()}
    z = np.load(os.path.join(gu.GOLDEN_DIR, 'model_full_at_24L_fp32.npz'))
    with torch.no_grad():
        gap, logits = model(batch)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            gap16, logits16 = model(batch)
    flat = logits.double().cpu().numpy().reshape(-1)
    idx = gu.sample_index(flat.size)
    ref_s, ref_norm = z['logits::samples'], float(z['logits::norm'])
    assert np.abs(flat[idx] - ref_s).max() <= 2e-3 * np.abs(ref_s).max()
    assert abs(np.linalg.norm(flat) - ref_norm) <= 1e-3 * ref_norm
    assert np.abs(gap.double().cpu().numpy() - z['gap::full']).max() < 1e-3
    agree = (logits.argmax(-1).cpu().numpy() == z['logits_argmax::full']).mean()
    assert agree > 0.99
    # stated bf16 tolerance: <= 3e-2 rel-L2 on the 24L logits, gap <= 1e-2 (SURVEY §8c anchors)
    f16 = logits16.double().cpu().numpy().reshape(-1)
    assert np.linalg.norm(f16[idx] - ref_s) <= 3e-2 * np.linalg.norm(ref_s)
    assert np.abs(gap16.double().cpu().numpy() - z['gap::full']).max() < 5e-2


def _sampled_ok(t, z, key, tol):
    """t against the golden's samples + norm of `key` (rel-L2 on the samples, relative norm)"""
    flat = t.detach().double().cpu().numpy().reshape(-1)
    if key + '::full' in z.files:
        ref = z[key + '::full'].reshape(-1)
        parity_log.record(float(np.linalg.norm(flat - ref) / max(np.linalg.norm(ref), 1e-30)))
        return np.linalg.norm(flat - ref) <= tol * max(np.linalg.norm(ref), 1e-30), (key, np.linalg.norm(flat - ref), np.linalg.norm(ref))
    ref_s = z[key + '::samples']
    s = flat[gu.sample_index(flat.size)]
    parity_log.record(float(np.linalg.norm(s - ref_s) / np.linalg.norm(ref_s)))
    ok = np.linalg.norm(s - ref_s) <= tol * np.linalg.norm(ref_s) and abs(np.linalg.norm(flat) - float(z[key + '::norm'])) <= tol * float(z[key + '::norm'])
    return ok, (key, np.linalg.norm(s - ref_s) / np.linalg.norm(ref_s))


def test_full_width_agx2_12x2_forward_vs_reference_golden():
    """BASELINE config 1 exactly: TGT-Agx2 12 shared layers x 2 distance predictor at full width on the 8-graph ragged
    mini-batch (N <= 32), eval forward, against the reference's fp32 CPU forward (lib/models/pcqm/distance_predictor.py:9-55);
    fp32, and bf16 / fp16 autocast within the stated 24L tolerance (3e-2 rel-L2 on the logits)."""
    from tgt_amd.pcqm import TGT_Distance
    model = gu.fill_params(TGT_Distance(**gu.FULL_AGX2_CFG), seed=920).cuda().eval()
    batch = {k: v.cuda() for k, v in gu.model_batch(gu.FULL_AGX2_GEOM, seed=921).items()}
    z = np.load(os.path.join(gu.GOLDEN_DIR, 'model_full_agx2_12x2_fp32.npz'))
    with torch.no_grad():
        logits = model(batch)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            l_bf = model(batch)
        with torch.autocast('cuda', dtype=torch.float16):
            l_fp = model(batch)
    tag('fp32')
    ok, info = _sampled_ok(logits, z, 'logits', FP32_MODEL_TOL)
    assert ok, info
    valid = (batch['edge_mask'] > 0).cpu().numpy()
    agree = (logits.argmax(-1).cpu().numpy() == z['logits_argmax::full'])[valid].mean()
    assert agree > 0.99, agree
    for name, l16 in (('bf16', l_bf), ('fp16', l_fp)):
        tag(name)
        ok, info = _sampled_ok(l16, z, 'logits', 3e-2)
        assert ok, (name, info)


@pytest.mark.parametrize('which', ['n32', 'n48', 'n48_b8'])
def test_full_width_24L_n32_vs_reference_golden(which):
    """TGT-At 24L at BASELINE widths AND the benchmarks' node counts (B = 2, one ragged graph; N = 32: the kernels the bench
    line runs -- projection-fused triplet forward, the round-4 backward, the fused edge Linears; N = 48 = BASELINE config 4: the
    16-wide triplet kernels and the lane-per-head node attention) against the REFERENCE's fp32 run -- eval forward, then loss and
    parameter gradients in train mode with every dropout off (fp32 and bf16 autocast)."""
    from tgt_amd.pcqm import TGT_Multi
    from tgt_amd.training.step import pretrain_loss, StepConfig
    name, geom, seeds = {'n32': ('model_full_at_24L_n32_fp32.npz', gu.FULL_AT_N32_GEOM, (930, 931)),
                         'n48': ('model_full_at_24L_n48_fp32.npz', gu.FULL_AT_N48_GEOM, (940, 941)),
                         # BASELINE config 4 as a mini-batch: 8 ragged graphs (17..48 nodes), Gaussian 3-D embedding
                         'n48_b8': ('model_full_at_24L_n48_b8_fp32.npz', gu.FULL_AT_N48_B8_GEOM, (960, 961))}[which]
    z = np.load(os.path.join(gu.GOLDEN_DIR, name))
    batch = {k: v.cuda() for k, v in gu.model_batch(geom, seed=seeds[1]).items()}
    model = gu.fill_params(TGT_Multi(**gu.FULL_AT_CFG), seed=seeds[0]).cuda().eval()
    with torch.no_grad():
        gap, logits = model(batch)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            gap16, logits16 = model(batch)
    tag('fp32')
    ok, info = _sampled_ok(logits, z, 'logits', FP32_MODEL_TOL)
    assert ok, info
    assert np.abs(gap.double().cpu().numpy() - z['gap::full']).max() < 1e-3
    valid = (batch['edge_mask'] > 0).cpu().numpy()
    assert (logits.argmax(-1).cpu().numpy() == z['logits_argmax::full'])[valid].mean() > 0.99
    tag('bf16')
    ok, info = _sampled_ok(logits16, z, 'logits', 3e-2)
    assert ok, info
    assert np.abs(gap16.double().cpu().numpy() - z['gap::full']).max() < 5e-2
    del model
    cfg = StepConfig(num_dist_bins=512, mixed_precision=None)
    loss_ref = float(z['loss::full'])
    drift = gu.bf16_drift('full_at_24L')       # the reference's own bf16-autocast drift (B = 2, N = 12 case): the anchor of the bf16 tolerances
    for mode, tol_loss, tol_grad in (('fp32', 2e-5, FP32_MODEL_TOL), ('bf16', 1e-3, None)):
        model = gu.fill_params(TGT_Multi(**gu.FULL_AT_CFG), seed=seeds[0]).cuda().train()
        ctx = torch.autocast('cuda', dtype=torch.bfloat16) if mode == 'bf16' else torch.autocast('cuda', enabled=False)
        tag(mode)
        with ctx:
            loss = pretrain_loss(model(batch), batch, cfg)
        loss.backward()
        parity_log.record(abs(float(loss.detach()) - loss_ref) / abs(loss_ref))
        assert abs(float(loss.detach()) - loss_ref) < tol_loss * abs(loss_ref), (mode, float(loss.detach()), loss_ref)
        pm = dict(model.named_parameters())
        for k in gu.FULL_GRAD_KEYS:
            tol = tol_grad if tol_grad is not None else 2.0 * drift['pgrad.' + k]
            ok, info = _sampled_ok(pm[k].grad, z, 'pgrad.' + k, tol)
            assert ok, (mode, info, tol)
        del model


def test_full_width_gap_agx2_12x2_vs_reference_golden():
    """BASELINE config 5, second stage, at full width: the TGT-Agx2 12 shared layers x 2 GAP predictor (aggregate triplets,
    Gaussian 3-D embedding) on 8 ragged graphs whose distance input went through the bins format (tgt_bins_to_dist on the device,
    bit-equal to the reference's BinsProcessor.bins2dist), eval forward against the REFERENCE's fp32 CPU forward
    (lib/models/pcqm/gap_predictor.py:48-63): fp32, and fp16 / bf16 autocast (`mixed_precision: true` of the gap_pred YAMLs) within
    a stated multiple of the reference's OWN autocast drift on this very batch (stored with the golden)."""
    from tgt_amd import ops
    from tgt_amd.pcqm import TGT_Gap, predict
    z = np.load(os.path.join(gu.GOLDEN_DIR, 'model_full_gap_agx2_12x2_fp32.npz'))
    cpu = gu.model_batch(gu.FULL_AGX2_GEOM, seed=951)
    batch = {k: v.cuda() for k, v in cpu.items()}
    bins = torch.triu((batch['dist_input'].float() * (255 / 8)).long().clamp(0, 255), 1)
    ref_dist = torch.from_numpy(z['dist_input::full'])
    assert torch.equal(predict.bins2dist(bins, 8 / 255).cpu(), ref_dist)              # the device-side bins2dist (tgt_bins_to_dist), bit for bit
    batch['dist_input'] = ref_dist.cuda()
    model = gu.fill_params(TGT_Gap(**gu.FULL_GAP_AGX2_CFG), seed=950).cuda().eval()
    ref = z['gap::full']
    with torch.no_grad():
        gap = model(batch)
        with torch.autocast('cuda', dtype=torch.float16):
            gap16 = model(batch)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            gapbf = model(batch)
    e32 = np.abs(gap.double().cpu().numpy() - ref).max()
    e16 = np.abs(gap16.double().cpu().numpy() - ref).max()
    ebf = np.abs(gapbf.double().cpu().numpy() - ref).max()
    d16, dbf = float(z['fp16_drift::full']), float(z['bf16_drift::full'])
    print(f'gap agx2 12x2: fp32 {e32:.2e}; fp16 {e16:.2e} (reference drift {d16:.2e}); bf16 {ebf:.2e} (reference drift {dbf:.2e})')
    assert e32 < 2e-4, e32
    assert torch.isfinite(gap16).all() and torch.isfinite(gapbf).all()
    assert e16 < 4 * d16, (e16, d16)
    assert ebf < 2 * dbf, (ebf, dbf)


def test_state_dict_roundtrip_with_oracle():
    """A checkpoint written by the reference-schema model loads strictly."""
    from tgt_amd.pcqm import TGT_Multi
    kwargs = gu.MODEL_CASES['multi_at_tiny'][1]
    src = om.TGT_Multi(**kwargs)
    dst = TGT_Multi(**kwargs)
    dst.load_state_dict(src.state_dict(), strict=True)


def test_trainer_step_runs_and_matches_torch_adam():
    from tgt_amd.pcqm import TGT_Multi
    from tgt_amd.training.step import Trainer, StepConfig, preprocess_batch
    from tgt_amd.training.synthetic import make_batch
    kwargs = gu.MODEL_CASES['multi_at_tiny'][1]
    cfg = StepConfig(num_dist_bins=24, mixed_precision=None, coords_noise=0.0, lr_warmup_steps=10, lr_total_steps=100)
    torch.manual_seed(0)
    m1 = gu.fill_params(TGT_Multi(**kwargs), seed=3).cuda()
    m2 = gu.fill_params(TGT_Multi(**kwargs), seed=3).cuda()
    tr = Trainer(m1, cfg)
    opt = torch.optim.Adam(m2.parameters(), lr=1.0)
    from tgt_amd.training.step import pretrain_loss, lr_at
    for step in range(1, 4):
        batch = preprocess_batch(make_batch(3, 7, seed=40 + step, ragged=True), 'cuda', cfg, add_noise=False)
        _, loss1 = tr.training_step(batch)
        for g in opt.param_groups:
            g['lr'] = lr_at(step, cfg)
        opt.zero_grad(set_to_none=True)
        loss2 = pretrain_loss(m2(batch), batch, cfg)
        loss2.backward()
        opt.step()
        assert abs(float(loss1) - float(loss2)) < 1e-4 * abs(float(loss2))
    p1 = torch.cat([p.detach().reshape(-1) for p in m1.parameters()])
    p2 = torch.cat([p.detach().reshape(-1) for p in m2.parameters()])
    assert rel(p1, p2) < 1e-4


def test_node_side_stream_gives_identical_gradients():
    """the node FFN of every layer runs on a second HIP stream under the edge kernels: the
    losses and all gradients must be bit-identical to the single-stream run (no race), also
    through the bucketed gradient path that reads gradients inside autograd hooks."""
    import torch.distributed as dist
    from tgt_amd import ops
    from tgt_amd.pcqm import TGT_Multi
    from tgt_amd.training.step import Trainer, StepConfig, preprocess_batch
    from tgt_amd.training.synthetic import make_batch
    kwargs = dict(gu.MODEL_CASES['multi_at_tiny'][1])
    kwargs.update(model_height=6)
    cfg = StepConfig(num_dist_bins=24, mixed_precision='bf16', coords_noise=0.0, bucket_mbytes=0.05)
    own_pg = not dist.is_initialized()
    if own_pg:
        dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29517', rank=0, world_size=1)
    try:
        grads = {}
        for enabled in (False, True, True):
            ops.side_stream.enabled = enabled
            model = gu.fill_params(TGT_Multi(**kwargs), seed=5).cuda()
            tr = Trainer(model, cfg, force_distributed=True)
            outs = []
            for step in range(3):
                batch = preprocess_batch(make_batch(16, 24, seed=70 + step, ragged=True), 'cuda', cfg, add_noise=False)
                model.eval()                               # dropouts off: runs are comparable bit for bit
                loss = tr.compute_gradients(batch)[1]
                outs.append((float(loss), tr.flat.grad.clone()))
            torch.cuda.synchronize()
            grads.setdefault(enabled, []).append(outs)
        ref = grads[False][0]
        for run in grads[True]:
            for (l0, g0), (l1, g1) in zip(ref, run):
                assert l0 == l1
                assert torch.equal(g0, g1)
    finally:
        ops.side_stream.enabled = True
        if own_pg:
            dist.destroy_process_group()


def test_cfg4_two_node_tiles_model_step():
    """BASELINE cfg 4 shape class: N up to 48 (two node tiles), Gaussian 3-D embedding,
    ragged batch; fp32 HIP path vs the oracle (outputs, loss, a few gradients)."""
    from tgt_amd.pcqm import TGT_Multi
    from tgt_amd.training.step import pretrain_loss, StepConfig
    kwargs = dict(gu.MODEL_CASES['multi_at_tiny'][1])
    geom = dict(B=3, N=48, num_nodes=[48, 33, 40])
    model = gu.fill_params(TGT_Multi(**kwargs), seed=21).cuda().train()
    ref = gu.fill_params(om.TGT_Multi(**kwargs), seed=21).train()
    cpu = gu.model_batch(geom, seed=22)
    batch = {k: v.cuda() for k, v in cpu.items()}
    cfg = StepConfig(num_dist_bins=kwargs['num_dist_bins'], mixed_precision=None)
    out = model(batch)
    loss = pretrain_loss(out, batch, cfg)
    loss.backward()
    g_ref, l_ref = ref(cpu)
    loss_ref = torch.nn.functional.l1_loss(g_ref, cpu['target']) + 0.1 * core.binned_distance_xent(
        l_ref, core.pairwise_dist(cpu['dft_coords']), cpu['edge_mask'], kwargs['num_dist_bins'], 8)
    loss_ref.backward()
    assert rel(out[0], g_ref) < 1e-3 and rel(out[1], l_ref) < 1e-3
    assert abs(float(loss) - float(loss_ref)) < 1e-4 * abs(float(loss_ref))
    pm, pr = dict(model.named_parameters()), dict(ref.named_parameters())
    for k in gu.GRAD_PROBE_KEYS:
        if k in pm and pr[k].grad is not None:
            assert rel(pm[k].grad, pr[k].grad) < 5e-3, (k, rel(pm[k].grad, pr[k].grad))


def test_cfg5_two_stage_fp16_inference():
    """BASELINE cfg 5 shape class: TGT-Agx2 (shared-weight x2, aggregate) distance predictor ->
    argmax bins -> bins2dist -> gap predictor under fp16 autocast (the reference's
    `mixed_precision: true`), eval mode; vs the fp32 oracle."""
    from tgt_amd.pcqm import TGT_Distance, TGT_Gap
    dk = dict(gu.MODEL_CASES['dist_agx2_tiny'][1])
    gk = dict(gu.MODEL_CASES['gap_at_tiny'][1])
    gk.update(embed_3d_type='gaussian', triplet_type='aggregate', layer_multiplier=2)
    geom = dict(B=4, N=9, num_nodes=[9, 6, 9, 4])
    cpu = gu.model_batch(geom, seed=31)
    batch = {k: v.cuda() for k, v in cpu.items()}

    def run(dist_model, gap_model, b, half):
        with torch.no_grad():
            logits = dist_model(b)
            bins = logits.float().argmax(-1)
            bins = torch.triu(bins, 1)
            dist = core.bins_to_dist(bins, 8 / (dk['num_dist_bins'] - 1)).to(b['dist_input'].dtype)
            b2 = dict(b)
            b2['dist_input'] = dist
            return logits, gap_model(b2)

    d_ref = gu.fill_params(om.TGT_Distance(**dk), seed=32).eval()
    g_ref = gu.fill_params(om.TGT_Gap(**gk), seed=33).eval()
    l_ref, gap_ref = run(d_ref, g_ref, cpu, False)
    d_hip = gu.fill_params(TGT_Distance(**dk), seed=32).cuda().eval()
    g_hip = gu.fill_params(TGT_Gap(**gk), seed=33).cuda().eval()
    hb = batch
    with torch.autocast('cuda', dtype=torch.float16):
        l_hip, gap_hip = run(d_hip, g_hip, hb, True)
    assert l_hip.dtype == torch.float16 and torch.isfinite(l_hip).all() and torch.isfinite(gap_hip).all()
    assert rel(l_hip, l_ref) < 2e-2, rel(l_hip, l_ref)
    # the gap stage sees the hip model's own predicted bins; feed it the oracle's to isolate it
    with torch.no_grad():
        bins = torch.triu(l_ref.argmax(-1), 1)
        b2 = dict(hb)
        b2['dist_input'] = core.bins_to_dist(bins, 8 / (dk['num_dist_bins'] - 1)).cuda()
        with torch.autocast('cuda', dtype=torch.float16):
            gap_iso = g_hip(b2)
    assert (gap_iso.float().cpu() - gap_ref).abs().max() < 3e-2 * gap_ref.abs().max()


def test_trainer_rccl_bucket_path_single_rank():
    """The data-parallel code path on the GPU with RCCL (one rank: all-reduce is the identity):
    autograd hooks -> bucket gather -> async all_reduce -> Adam must give the same parameters as
    the plain path."""
    import torch.distributed as dist
    from tgt_amd.pcqm import TGT_Multi
    from tgt_amd.training.step import Trainer, StepConfig, preprocess_batch
    from tgt_amd.training.synthetic import make_batch
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    created = False
    if not dist.is_initialized():
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
        created = True
    try:
        kwargs = gu.MODEL_CASES['multi_at_tiny'][1]
        cfg = StepConfig(num_dist_bins=24, mixed_precision='bf16', coords_noise=0.0, bucket_mbytes=0,
                         lr_warmup_steps=10, lr_total_steps=100)
        m1 = gu.fill_params(TGT_Multi(**kwargs), seed=3).cuda()
        m2 = gu.fill_params(TGT_Multi(**kwargs), seed=3).cuda()
        t1 = Trainer(m1, cfg, force_distributed=True)
        t2 = Trainer(m2, cfg)
        assert t1.distributed and t1.buckets is not None and len(t1.buckets) > 10 and not t2.distributed
        for step in range(3):
            batch = preprocess_batch(make_batch(3, 7, seed=50 + step, ragged=True), 'cuda', cfg, add_noise=False)
            _, l1 = t1.training_step(batch)
            _, l2 = t2.training_step(batch)
            assert abs(float(l1) - float(l2)) < 1e-6 * abs(float(l2))
        torch.cuda.synchronize()
        assert rel(t1.flat.param, t2.flat.param) < 1e-6
    finally:
        if created:
            dist.destroy_process_group()


def test_full_width_24L_training_gradients_vs_oracle():
    """TGT-At 24L at BASELINE widths: loss and parameter gradients of one training-step-equivalent
    (dropouts off) on the HIP path, fp32 and bf16 autocast, against the oracle in fp32 on the CPU."""
    from tgt_amd.pcqm import TGT_Multi
    from tgt_amd.training.step import pretrain_loss, StepConfig
    geom = dict(B=2, N=12, num_nodes=[12, 9])
    cpu = gu.model_batch(geom, seed=911)
    ref = gu.fill_params(om.TGT_Multi(**gu.FULL_AT_CFG), seed=910).train()
    g_ref, l_ref = ref(cpu)
    loss_ref = torch.nn.functional.l1_loss(g_ref, cpu['target']) + 0.1 * core.binned_distance_xent(
        l_ref, core.pairwise_dist(cpu['dft_coords']), cpu['edge_mask'], 512, 8)
    loss_ref.backward()
    pr = dict(ref.named_parameters())
    keys = gu.FULL_GRAD_KEYS
    drift = gu.bf16_drift('full_at_24L')       # the reference's own bf16-autocast drift on this case (0.3 .. 1.3e-2)
    batch = {k: v.cuda() for k, v in cpu.items()}
    cfg = StepConfig(num_dist_bins=512, mixed_precision=None)
    for mode, tol_loss, tol_grad in (('fp32', 2e-5, FP32_MODEL_TOL), ('bf16', 1e-3, None)):
        model = gu.fill_params(TGT_Multi(**gu.FULL_AT_CFG), seed=910).cuda().train()
        ctx = torch.autocast('cuda', dtype=torch.bfloat16) if mode == 'bf16' else torch.autocast('cuda', enabled=False)
        tag(mode)
        with ctx:
            loss = pretrain_loss(model(batch), batch, cfg)
        loss.backward()
        assert abs(float(loss.detach()) - float(loss_ref.detach())) < tol_loss * abs(float(loss_ref.detach())), mode
        pm = dict(model.named_parameters())
        for k in keys:
            tol = tol_grad if tol_grad is not None else 2.0 * drift['pgrad.' + k]      # bf16: 2x the reference's drift
            assert rel(pm[k].grad, pr[k].grad) < tol, (mode, k, rel(pm[k].grad, pr[k].grad), tol)
        del model


def test_drop_path_folded_into_producers_matches_the_plain_wiring(monkeypatch):
    """Three TGT layers with DropPath on: the factor folded into the node attention's H_hat and the FFN's activation
    (ops.can_prescale -> linear_residual_layer_norm(prescaled=True)) against the plain wiring (scale applied by the residual
    entry, scaled gradient copy in its backward) on the SAME drop masks: outputs and every parameter gradient."""
    from tgt_amd import ops
    from tgt_amd.tgt import Graph, TGT_Encoder
    monkeypatch.setattr(ops, '_EDGE_MIN_ROWS', 1)
    B, N, W, C, H = 6, 8, 256, 256, 64
    kw = dict(node_width=W, edge_width=C, num_heads=H, activation='gelu', scale_degree=True, node_update=True, edge_update=True,
              triplet_heads=16, triplet_type='attention', triplet_dropout=0, node_ffn_multiplier=1., edge_ffn_multiplier=1.,
              source_dropout=0., drop_path=0.4, node_act_dropout=0., edge_act_dropout=0.)
    g = torch.Generator(device='cuda').manual_seed(3)
    h0 = torch.randn(B, N, W, device='cuda', generator=g)
    e0 = torch.randn(B, N, N, C, device='cuda', generator=g)
    mask = gu.additive_mask([8, 5, 8, 3, 8, 6], N, torch.float32).cuda()
    gh, ge = torch.randn(B, N, W, device='cuda', generator=g), torch.randn(B, N, N, C, device='cuda', generator=g)
    runs = []
    for fold in (True, False):
        monkeypatch.setattr(ops, '_PRESCALE', fold)
        layer = gu.fill_params(TGT_Encoder(model_height=3, **kw), seed=9).cuda().train()      # (layers hand their closing residual on)
        torch.manual_seed(123)
        ops.reset_random_pools()
        h, e = h0.clone().requires_grad_(True), e0.clone().requires_grad_(True)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            out = layer(Graph(h=h, e=e, mask=mask))
        ((out.h.float() * gh).sum() + (out.e.float() * ge).sum()).backward()
        torch.cuda.synchronize()
        runs.append((out.h.detach(), out.e.detach(), h.grad, e.grad, {k: p.grad for k, p in layer.named_parameters()}))
    a, b = runs
    for i, name in enumerate(('h', 'e', 'dh', 'de')):
        assert rel(a[i], b[i]) < (2e-2 if i < 2 else 4e-2), (name, rel(a[i], b[i]))
    for k in b[4]:
        assert (a[4][k] is None) == (b[4][k] is None), k
        if b[4][k] is not None and float(b[4][k].abs().max()) > 0:
            assert rel(a[4][k], b[4][k]) < 6e-2, (k, rel(a[4][k], b[4][k]))


# ---------------------------------------------------------------------------------------------------------------
# The BASELINE-size code path at model level.  ops._EDGE_MIN_ROWS / _SPLIT_MIN_ROWS (65536 rows) keep every small model
# test above on the UNFUSED edge path; the benchmark (262144 rows) runs the fused tgt_edge_linear launches, 32 row tiles per
# persistent workgroup.  These tests (a) put the golden 24L cases on the fused path with the workgroup count capped, so
# that each workgroup walks >= 3 tiles, and (b) hold a B = 256 run to the B = 8 slice the oracle pins.
# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture
def fused_edge_path(monkeypatch):
    from tgt_amd import ops, _lib
    monkeypatch.setattr(ops, '_EDGE_MIN_ROWS', 1)
    monkeypatch.setattr(ops, '_SPLIT_MIN_ROWS', 1)
    _lib.lib().tgt_edge_linear_set_grid_cap(2)          # 288 rows = 9 row tiles on 2 workgroups: 5 and 4 tiles each
    yield
    _lib.lib().tgt_edge_linear_set_grid_cap(0)


def test_full_width_24L_on_the_fused_edge_path_vs_reference_golden(fused_edge_path):
    """TGT-At 24L at BASELINE widths, bf16 autocast, every edge Linear on the fused launches of the benchmark
    (lin_O_e / lin_W2 + residual + LayerNorm, lin_W1 + GELU, slice kernels, split projection), >= 4 tiles per workgroup:
    eval forward vs the reference's fp32 CPU forward, and loss + parameter gradients vs the oracle (dropouts off)."""
    from tgt_amd import ops
    from tgt_amd.pcqm import TGT_Multi
    from tgt_amd.training.step import pretrain_loss, StepConfig
    geom = dict(B=2, N=12, num_nodes=[12, 9])
    # forward, against the golden file of test_full_width_24L_forward_vs_reference_golden
    model = gu.fill_params(TGT_Multi(**gu.FULL_AT_CFG), seed=900).cuda().eval()
    batch = {k: v.cuda() for k, v in gu.model_batch(geom, seed=901).items()}
    z = np.load(os.path.join(gu.GOLDEN_DIR, 'model_full_at_24L_fp32.npz'))
    prof = ops.profile_kernels(True)
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
        gap16, logits16 = model(batch)
    torch.cuda.synchronize()
    ops.profile_kernels(False)
    # the fused path DID run: lin_EG, lin_O_e + res + LN, third-arm E/G, lin_W1 + GELU, lin_W2 + res + LN per layer (the last layer closes differently)
    assert len(prof.get('tgt_edge_linear', ())) >= 24 * 5 - 4, {k: len(v) for k, v in prof.items()}
    f16 = logits16.double().cpu().numpy().reshape(-1)
    idx = gu.sample_index(f16.size)
    ref_s = z['logits::samples']
    assert np.linalg.norm(f16[idx] - ref_s) <= 3e-2 * np.linalg.norm(ref_s)
    assert np.abs(gap16.double().cpu().numpy() - z['gap::full']).max() < 5e-2
    del model
    # training-step-equivalent, against the oracle (as test_full_width_24L_training_gradients_vs_oracle, bf16 leg)
    cpu = gu.model_batch(geom, seed=911)
    ref = gu.fill_params(om.TGT_Multi(**gu.FULL_AT_CFG), seed=910).train()
    g_ref, l_ref = ref(cpu)
    loss_ref = torch.nn.functional.l1_loss(g_ref, cpu['target']) + 0.1 * core.binned_distance_xent(
        l_ref, core.pairwise_dist(cpu['dft_coords']), cpu['edge_mask'], 512, 8)
    loss_ref.backward()
    pr = dict(ref.named_parameters())
    drift = gu.bf16_drift('full_at_24L')
    batch = {k: v.cuda() for k, v in cpu.items()}
    cfg = StepConfig(num_dist_bins=512, mixed_precision=None)
    model = gu.fill_params(TGT_Multi(**gu.FULL_AT_CFG), seed=910).cuda().train()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        loss = pretrain_loss(model(batch), batch, cfg)
    loss.backward()
    assert abs(float(loss.detach()) - float(loss_ref.detach())) < 1e-3 * abs(float(loss_ref.detach()))
    pm = dict(model.named_parameters())
    for k in gu.FULL_GRAD_KEYS:
        tol = 2.0 * drift['pgrad.' + k]
        assert rel(pm[k].grad, pr[k].grad) < tol, (k, rel(pm[k].grad, pr[k].grad), tol)


def test_baseline_batch_256_equals_its_8_graph_slice():
    """BASELINE config 2 geometry (TGT-At 24L, B = 256, N = 32, bf16 autocast, dropouts off): graphs are independent, so the
    outputs of graphs [0:8] inside the 256-graph batch must equal a run on those 8 graphs alone -- the size the oracle pins --
    and the parameter gradients of a loss that only sees those 8 graphs must equal the 8-graph run's.  Not bit for bit: the
    library picks other GEMM tilings for 262144 rows than for 8192; the stated bound is the reference's own bf16 drift
    (tests/golden/bf16_drift.npz, full_at_24L), i.e. the two runs may differ by no more than bf16 arithmetic itself does."""
    from tgt_amd.pcqm import TGT_Multi
    from tgt_amd.training.configs import tgt_at_24l
    from tgt_amd.training.step import StepConfig, preprocess_batch
    from tgt_amd.training.synthetic import make_batch
    cfg = StepConfig(num_dist_bins=512, mixed_precision='bf16', coords_noise=0.0)
    torch.manual_seed(0)
    model = TGT_Multi(**tgt_at_24l(dropouts=False)).cuda().train()
    host = make_batch(256, 32, seed=4242)
    big = preprocess_batch(host, 'cuda', cfg, add_noise=False)
    small = {k: v[:8].contiguous() for k, v in big.items()}
    drift = gu.bf16_drift('full_at_24L')
    keys = gu.FULL_GRAD_KEYS
    runs = []
    for b in (big, small):
        model.zero_grad(set_to_none=True)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            gap, logits = model(b)
        ((gap[:8].float() ** 2).sum() + (logits[:8].float() ** 2).mean()).backward()
        torch.cuda.synchronize()
        runs.append((gap[:8].detach().clone(), logits[:8].detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()
                                                                             if k in keys}))
    (g256, l256, p256), (g8, l8, p8) = runs
    assert rel(l256, l8) < 2.0 * drift['logits'], rel(l256, l8)
    assert float((g256 - g8).abs().max()) < 2e-2, float((g256 - g8).abs().max())
    assert float((l256.argmax(-1) == l8.argmax(-1)).float().mean()) > 0.97
    tol = 2.0 * max(v for k, v in drift.items() if k.startswith('pgrad.'))      # (another loss than the drift file's: one bound for all)
    for k in keys:
        assert rel(p256[k], p8[k]) < tol, (k, rel(p256[k], p8[k]), tol)
