#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REAL reference on CPU.

Runs only in the build container (needs /root/reference, which never travels
to the GPU box).  Usage:  python tools/make_golden.py [/root/reference]

Nothing from the reference is copied: it is imported, driven with inputs and
parameters regenerated from numpy seeds (tests/golden_util.py), and only its
numerical OUTPUTS are stored.  numba is absent here; the reference's scheme
modules are not needed for these vectors (lib.tgt, lib.models.pcqm and
lib.training_schemes.pcqm.commons import without it).
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, REF)

import golden_util as gu                                            # noqa: E402
from lib.tgt.layers import layers as ref_layers                    # noqa: E402
from lib.tgt.layers import triplet as ref_triplet                  # noqa: E402
from lib.models.pcqm.multitask import TGT_Multi                    # noqa: E402
from lib.models.pcqm.distance_predictor import TGT_Distance        # noqa: E402
from lib.models.pcqm.gap_predictor import TGT_Gap                  # noqa: E402
from lib.training_schemes.pcqm import commons as ref_commons       # noqa: E402

torch.set_num_threads(8)
REF_CLASSES = {}
for modl in (ref_layers, ref_triplet):
    for k in dir(modl):
        v = getattr(modl, k)
        if isinstance(v, type) and issubclass(v, torch.nn.Module):
            REF_CLASSES[k] = v
REF_CLASSES.update(TGT_Multi=TGT_Multi, TGT_Distance=TGT_Distance, TGT_Gap=TGT_Gap)


def save(name, arrays):
    os.makedirs(gu.GOLDEN_DIR, exist_ok=True)
    path = os.path.join(gu.GOLDEN_DIR, name + '.npz')
    np.savez_compressed(path, **arrays)
    print(f'  wrote {path}  ({os.path.getsize(path)/1024:.1f} KiB)')


def flatten(res, full):
    out = {}
    for k, t in res.items():
        for kk, a in gu.summarize(t, full).items():
            out[f'{k}::{kk}'] = a
    return out


def op_cases():
    for i, (name, (cls_name, kwargs, geom)) in enumerate(gu.OP_CASES.items()):
        t0 = time.time()
        res = gu.run_op_case(REF_CLASSES[cls_name], kwargs, geom, seed=100 + i)
        full = geom is gu.SMALL
        print(f'{name}: {cls_name} {time.time()-t0:.1f}s')
        save('op_' + name, flatten(res, full))


def pretrain_loss(outputs, batch, num_bins, range_bins=8, weight=0.1):
    """reference lib/training_schemes/pcqm/pretrain/scheme.py:78-88"""
    gap, logits = outputs
    prim = torch.nn.functional.l1_loss(gap, batch['target'])
    dist_targ = ref_commons.coords2dist(batch['dft_coords'])
    dl = ref_commons.DiscreteDistLoss(num_bins, range_bins)(logits, dist_targ, batch['edge_mask'])
    return prim + weight * dl


def model_cases():
    for i, (name, (cls_name, kwargs, geom)) in enumerate(gu.MODEL_CASES.items()):
        model = gu.fill_params(REF_CLASSES[cls_name](**kwargs).double(), seed=500 + i)
        model.train()            # all dropout rates are 0 -> deterministic (training_step runs in train mode)
        batch = gu.model_batch(geom, seed=600 + i)
        batch['dist_input'] = batch['dist_input'].double()
        out = model(batch)
        res = {}
        if cls_name == 'TGT_Multi':
            res['gap'], res['logits'] = out
            loss = pretrain_loss(out, batch, kwargs['num_dist_bins'])
        elif cls_name == 'TGT_Distance':
            res['logits'] = out
            dist_targ = ref_commons.coords2dist(batch['dft_coords'])
            loss = ref_commons.DiscreteDistLoss(kwargs['num_dist_bins'], 8)(out, dist_targ, batch['edge_mask'])
        else:
            res['gap'] = out
            loss = torch.nn.functional.l1_loss(out, batch['target'])
        res['loss'] = loss
        loss.backward()
        named = dict(model.named_parameters())
        for k in gu.GRAD_PROBE_KEYS:
            if k in named and named[k].grad is not None:
                res['pgrad.' + k] = named[k].grad
        print(f'{name}: {cls_name} loss={float(loss):.6f} dtype={loss.dtype}')
        save('model_' + name, flatten(res, True))


def full_width_case():
    """TGT-At 24L at BASELINE widths, fp32 (as the CPU reference path runs it),
    B=2 N=12 ragged; eval-mode forward only; sampled."""
    geom = dict(B=2, N=12, num_nodes=[12, 9])
    torch.manual_seed(0)
    model = gu.fill_params(TGT_Multi(**gu.FULL_AT_CFG), seed=900)
    model.eval()
    batch = gu.model_batch(geom, seed=901)
    t0 = time.time()
    with torch.no_grad():
        gap, logits = model(batch)
    print(f'full_at_24L fwd {time.time()-t0:.1f}s gap={gap.tolist()}')
    res = dict(gap=gap, logits=logits)
    out = {}
    for kk, a in gu.summarize(gap, True).items():
        out[f'gap::{kk}'] = a
    for kk, a in gu.summarize(logits, False).items():
        out[f'logits::{kk}'] = a
    out['logits_argmax::full'] = logits.argmax(-1).numpy().astype(np.int16)
    save('model_full_at_24L_fp32', out)


def sampled(prefix, t, full=False):
    return {f'{prefix}::{kk}': a for kk, a in gu.summarize(t, full).items()}


def full_agx2_case():
    """BASELINE config 1: TGT-Agx2 12L x 2 distance predictor at full width (lib/models/pcqm/distance_predictor.py:9-55), fp32
    eval forward on the 8-graph ragged mini-batch; sampled logits + every argmax bin."""
    torch.manual_seed(0)
    model = gu.fill_params(TGT_Distance(**gu.FULL_AGX2_CFG), seed=920).eval()
    batch = gu.model_batch(gu.FULL_AGX2_GEOM, seed=921)
    t0 = time.time()
    with torch.no_grad():
        logits = model(batch)
    print(f'full_agx2 fwd {time.time()-t0:.1f}s logits {tuple(logits.shape)}')
    out = sampled('logits', logits)
    out['logits_argmax::full'] = logits.argmax(-1).numpy().astype(np.int16)
    save('model_full_agx2_12x2_fp32', out)


def full_at_n32_case(geom=None, name='model_full_at_24L_n32_fp32', seeds=(930, 931)):
    """TGT-At 24L at BASELINE widths AND the benchmark's node count (B = 2, N = 32, one ragged graph), fp32: eval forward,
    then train mode with every dropout off: pretrain loss and the gradients of gu.FULL_GRAD_KEYS.  (geom = gu.FULL_AT_N48_GEOM:
    the same at BASELINE config 4's node count.)"""
    torch.manual_seed(0)
    model = gu.fill_params(TGT_Multi(**gu.FULL_AT_CFG), seed=seeds[0])
    batch = gu.model_batch(geom or gu.FULL_AT_N32_GEOM, seed=seeds[1])
    model.eval()
    t0 = time.time()
    with torch.no_grad():
        gap, logits = model(batch)
    out = sampled('gap', gap, True)
    out.update(sampled('logits', logits))
    out['logits_argmax::full'] = logits.argmax(-1).numpy().astype(np.int16)
    model.train()
    outs = model(batch)
    loss = pretrain_loss(outs, batch, gu.FULL_AT_CFG['num_dist_bins'])
    loss.backward()
    print(f'full_at_24L_n32 fwd+fwd/bwd {time.time()-t0:.1f}s loss={float(loss):.6f} dtype={loss.dtype}')
    out.update(sampled('loss', loss.detach(), True))
    named = dict(model.named_parameters())
    for k in gu.FULL_GRAD_KEYS:
        g = named[k].grad
        out.update(sampled('pgrad.' + k, g, g.numel() <= 4096))
    save(name, out)


def full_gap_agx2_case():
    """BASELINE config 5, second stage: TGT-Agx2 12 x 2 gap predictor at full width (lib/models/pcqm/gap_predictor.py:10-63) on 8
    ragged graphs whose distance input went through the bins format; fp32 eval forward = the golden; and the reference's OWN
    fp16-autocast result on the same batch (CPU autocast: Linear / matmul in half, LayerNorm / softmax fp32) as the anchor of the
    fp16 tolerance."""
    torch.manual_seed(0)
    model = gu.fill_params(TGT_Gap(**gu.FULL_GAP_AGX2_CFG), seed=950).eval()
    batch = gu.model_batch(gu.FULL_AGX2_GEOM, seed=951)
    batch['dist_input'] = gu.binned_dist_input(batch)
    # (cross-check of the helper against the reference's own BinsProcessor arithmetic)
    bp = ref_commons.BinsProcessor.__new__(ref_commons.BinsProcessor)
    bp.shift_half, bp.zero_diag, bp.bin_size = True, True, 8 / 255
    bins = torch.triu((gu.model_batch(gu.FULL_AGX2_GEOM, seed=951)['dist_input'].float() * (255 / 8)).long().clamp(0, 255), 1)
    assert torch.equal(bp.bins2dist(bins.float()), batch['dist_input']), 'binned_dist_input != BinsProcessor.bins2dist'
    t0 = time.time()
    with torch.no_grad():
        gap = model(batch)
        try:
            with torch.autocast('cpu', dtype=torch.float16):
                gap16 = model(batch).float()
        except Exception as e:                      # (a CPU op without a half kernel: the anchor falls back to bf16's)
            print('  fp16 CPU autocast failed:', repr(e)[:200])
            gap16 = None
        with torch.autocast('cpu', dtype=torch.bfloat16):
            gapbf = model(batch).float()
    print(f'full_gap_agx2 fwd {time.time()-t0:.1f}s gap={gap.tolist()}')
    out = sampled('gap', gap, True)
    out['dist_input::full'] = batch['dist_input'].numpy()
    if gap16 is not None:
        out['fp16_drift::full'] = np.array(float((gap16 - gap).abs().max()))
    out['bf16_drift::full'] = np.array(float((gapbf - gap).abs().max()))
    print('  drift (max abs): fp16', None if gap16 is None else float((gap16 - gap).abs().max()), 'bf16', float((gapbf - gap).abs().max()))
    save('model_full_gap_agx2_12x2_fp32', out)


def misc_cases():
    rng = np.random.default_rng(4242)
    coords = torch.from_numpy(rng.standard_normal((2, 5, 3)).astype(np.float32))
    d = ref_commons.coords2dist(coords)
    logits = torch.from_numpy(rng.standard_normal((2, 5, 5, 16)))
    em = torch.ones(2, 5, 5, dtype=torch.uint8)
    em[1, 3:, :] = 0
    em[1, :, 3:] = 0
    lossfn = ref_commons.DiscreteDistLoss(16, 8)
    out = {
        'coords2dist::full': d.numpy(),
        'xent_reduced::full': lossfn(logits, d.double(), em).numpy(),
        'xent_per_graph::full': lossfn(logits, d.double(), em, reduce=False).numpy(),
        'discrete_dist::full': ref_commons.discrete_dist(d, 16, 8).numpy(),
    }
    bp = ref_commons.BinsProcessor.__new__(ref_commons.BinsProcessor)
    bp.shift_half, bp.zero_diag, bp.bin_size = True, True, 8 / 15
    bins = torch.from_numpy(np.triu(rng.integers(0, 16, size=(2, 5, 5)), 1))
    out['bins2dist::full'] = bp.bins2dist(bins).numpy()
    out['bins_in::full'] = bins.numpy()
    # state_dict schema manifest: key -> shape, TGT-At Multi
    sd = TGT_Multi(**gu.FULL_AT_CFG).state_dict()
    out['manifest_keys::full'] = np.array(list(sd.keys()))
    out['manifest_shapes::full'] = np.array([','.join(map(str, v.shape)) for v in sd.values()])
    save('misc', out)


def predict_cases():
    """The reference's own prediction-scheme methods (dist_pred/scheme.py:139-229, gap_pred/scheme.py:78-135,
    bin_ops.py, commons.BinsProcessor) run on stand-in models that replay seeded outputs: the methods are called
    unbound on a small namespace carrying exactly the attributes they read.  numba is absent here: `njit` is
    stubbed to the identity (the functions are plain numpy loops)."""
    import types
    from types import SimpleNamespace
    nb = types.ModuleType('numba')
    nb.njit = lambda *a, **k: (lambda f: f)
    typed = types.ModuleType('numba.typed')
    typed.List = list
    nb.typed = typed
    sys.modules.setdefault('numba', nb)
    sys.modules.setdefault('numba.typed', typed)
    from lib.training_schemes.pcqm.dist_pred.scheme import SCHEME as DistScheme
    from lib.training_schemes.pcqm.gap_pred.scheme import SCHEME as GapScheme
    from lib.data.pcqm import bin_ops

    c = gu.PREDICT_CASE
    out = {}

    def replay():
        it = iter(gu.predict_logit_sequence())
        return lambda batch: next(it)

    batch = gu.model_batch(dict(B=c['B'], N=c['N'], num_nodes=c['num_nodes']), seed=c['seed'] + 1)
    batch['num_nodes'] = torch.tensor(c['num_nodes'])
    batch['idx'] = torch.arange(100, 100 + c['B'])
    lossfn = ref_commons.DiscreteDistLoss(c['num_bins'], c['range_bins'])
    fake = SimpleNamespace(nb_draw_samples=c['nb_samples'], model=replay(),
                           config=SimpleNamespace(num_dist_bins=c['num_bins']), xent_loss_fn=lossfn,
                           get_dist_target=lambda b: ref_commons.coords2dist(b['dft_coords']))
    fake.predict_bins = lambda b: DistScheme.predict_bins(fake, b)
    fake.predict_probs = lambda b: DistScheme.predict_probs(fake, b)
    bins = DistScheme.predict_bins(fake, batch)
    out['bins::full'] = bins.numpy()
    fake.model = replay()
    out['probs::full'] = DistScheme.predict_probs(fake, batch).numpy()
    fake.model = replay()
    out['eval_xent::full'] = DistScheme.prediction_step4eval(fake, batch)['loss'].numpy()
    fake.model = replay()
    saved = DistScheme.prediction_step4savebins(fake, batch)
    assert saved['bins'][0].dtype == np.uint8
    out['saved_idx::full'] = saved['idx']
    out['saved_bins_flat::full'] = np.concatenate(saved['bins'])
    out['saved_bins_lengths::full'] = np.array([len(b) for b in saved['bins']])
    # the gap stage's side: unpack (data.py:232-238) -> float32 -> BinsProcessor.bins2dist
    bp = ref_commons.BinsProcessor.__new__(ref_commons.BinsProcessor)
    bp.shift_half, bp.zero_diag, bp.bin_size = True, True, c['range_bins'] / (c['num_bins'] - 1)
    N = c['N']
    dist_bins = np.zeros((c['B'], c['nb_samples'], N, N), dtype=np.float32)      # padded_collate: zero padding
    for i, n in enumerate(c['num_nodes']):
        packed = saved['bins'][i].reshape(c['nb_samples'], -1)
        dist_bins[i, :, :n, :n] = bin_ops.unpack_bins_multi(packed, n).astype(np.float32)
    out['dist_bins::full'] = dist_bins
    dist_input = bp.bins2dist(torch.from_numpy(dist_bins))
    out['dist_input::full'] = dist_input.numpy()
    gbatch = dict(batch)
    gbatch['dist_input'] = dist_input
    gfake = SimpleNamespace(nb_draw_samples=4, model=gu.predict_gap_sequence())
    res = GapScheme.prediction_step(gfake, gbatch)
    out['gap_pred::full'] = res['gap_pred'].numpy()
    gp = dict(gap_pred=res['gap_pred'].numpy(), gap_target=res['gap_target'].numpy())
    out['gap_mae::full'] = np.array(GapScheme.evaluate_predictions(gfake, gp)['loss'])
    print('predict:', {k: v.shape for k, v in out.items()})
    save('predict', out)


def _rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).norm() / (b.norm() + 1e-300))


def bf16_drift_cases():
    """The reference's OWN bf16-autocast drift, the anchor of the bf16 tolerances of tests/test_hip_model.py
    ('grads <= 2x ...'): the reference model in fp32 under torch.autocast(bfloat16) against the same
    model in fp64 / fp32 without autocast, same parameters and batch as the golden cases; stored: rel-L2 of the
    outputs, the loss and every probed parameter gradient.  (CPU autocast: the container has no GPU; Linear / matmul /
    bmm run in bf16 exactly as under CUDA autocast, softmax / layer_norm / losses stay fp32.)"""
    out = {}

    def run(model, batch, cls_name, num_bins, autocast):
        model.zero_grad(set_to_none=True)
        with torch.autocast('cpu', dtype=torch.bfloat16, enabled=autocast):
            o = model(batch)
            if cls_name == 'TGT_Multi':
                loss = pretrain_loss(o, batch, num_bins)
                res = dict(gap=o[0], logits=o[1])
            elif cls_name == 'TGT_Distance':
                loss = ref_commons.DiscreteDistLoss(num_bins, 8)(o, ref_commons.coords2dist(batch['dft_coords']),
                                                                 batch['edge_mask'])
                res = dict(logits=o)
            else:
                loss = torch.nn.functional.l1_loss(o, batch['target'])
                res = dict(gap=o)
        res['loss'] = loss
        loss.backward()
        for k, p in model.named_parameters():
            if p.grad is not None:
                res['pgrad.' + k] = p.grad.clone()
        return {k: v.detach().clone() for k, v in res.items()}

    for i, (name, (cls_name, kwargs, geom)) in enumerate(gu.MODEL_CASES.items()):
        model = gu.fill_params(REF_CLASSES[cls_name](**kwargs), seed=500 + i).train()
        batch = gu.model_batch(geom, seed=600 + i)
        exact = run(model, batch, cls_name, kwargs.get('num_dist_bins', 0), False)
        low = run(model, batch, cls_name, kwargs.get('num_dist_bins', 0), True)
        for k in exact:
            if k.startswith('pgrad.') and k[6:] not in gu.GRAD_PROBE_KEYS:
                continue
            out[f'{name}::{k}'] = np.array(_rel(low[k], exact[k]))
        print(name, {k.split('::')[1][-28:]: float(v) for k, v in out.items() if k.startswith(name)})
    # TGT-At 24L at BASELINE widths, the case of test_full_width_24L_training_gradients_vs_oracle
    geom = dict(B=2, N=12, num_nodes=[12, 9])
    model = gu.fill_params(TGT_Multi(**gu.FULL_AT_CFG), seed=910).train()
    batch = gu.model_batch(geom, seed=911)
    t0 = time.time()
    exact = run(model, batch, 'TGT_Multi', 512, False)
    low = run(model, batch, 'TGT_Multi', 512, True)
    for k in exact:
        if k.startswith('pgrad.') and k[6:] not in gu.FULL_GRAD_KEYS:
            continue
        out[f'full_at_24L::{k}'] = np.array(_rel(low[k], exact[k]))
    print(f'full_at_24L {time.time()-t0:.0f}s', {k[-40:]: float(v) for k, v in out.items() if k.startswith('full_at_24L')})
    save('bf16_drift', out)


if __name__ == '__main__':
    which = sys.argv[2:] or ['op', 'model', 'misc', 'full', 'full_agx2', 'full_n32', 'full_n48', 'full_n48_b8', 'full_gap_agx2', 'drift', 'predict']
    if 'op' in which:
        op_cases()
    if 'model' in which:
        model_cases()
    if 'misc' in which:
        misc_cases()
    if 'full' in which:
        full_width_case()
    if 'full_agx2' in which:
        full_agx2_case()
    if 'full_n32' in which:
        full_at_n32_case()
    if 'full_n48' in which:
        full_at_n32_case(gu.FULL_AT_N48_GEOM, 'model_full_at_24L_n48_fp32', (940, 941))
    if 'full_n48_b8' in which:
        full_at_n32_case(gu.FULL_AT_N48_B8_GEOM, 'model_full_at_24L_n48_b8_fp32', (960, 961))
    if 'full_gap_agx2' in which:
        full_gap_agx2_case()
    if 'drift' in which:
        bf16_drift_cases()
    if 'predict' in which:
        predict_cases()
