"""Pin the oracle: oracle/ (torch restatement + explicit numpy form) against the
golden vectors captured from the real reference (tools/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import core, explicit_np, modules as om


def load(name):
    z = np.load(os.path.join(gu.GOLDEN_DIR, name + '.npz'), allow_pickle=False)
    out = {}
    for k in z.files:
        base, kind = k.split('::')
        out.setdefault(base, {})[kind] = z[k]
    return out


def check(t, ref, rtol, what):
    a = t.detach().double().cpu().numpy()
    if 'full' in ref:
        r = ref['full']
        assert a.shape == r.shape, what
        err = np.abs(a - r).max()
        scale = max(np.abs(r).max(), 1e-30)
        assert err <= rtol * scale + 1e-13, f'{what}: max-abs {err:.3e} vs scale {scale:.3e}'
    else:
        flat = a.reshape(-1)
        assert tuple(ref['shape']) == a.shape, what
        s = flat[gu.sample_index(flat.size)]
        scale = max(np.abs(ref['samples']).max(), 1e-30)
        assert np.abs(s - ref['samples']).max() <= rtol * scale + 1e-13, what
        assert abs(np.linalg.norm(flat) - ref['norm']) <= rtol * ref['norm'], what


@pytest.mark.parametrize('name', list(gu.OP_CASES))
def test_op_case_matches_reference(name):
    cls_name, kwargs, geom = gu.OP_CASES[name]
    i = list(gu.OP_CASES).index(name)
    res = gu.run_op_case(getattr(om, cls_name), kwargs, geom, seed=100 + i)
    gold = load('op_' + name)
    assert set(res) == set(gold), (sorted(res), sorted(gold))
    for k, t in res.items():
        check(t, gold[k], 1e-11, f'{name}:{k}')


def pretrain_loss(out, batch, num_bins):
    gap, logits = out
    prim = torch.nn.functional.l1_loss(gap, batch['target'])
    dl = core.binned_distance_xent(logits, core.pairwise_dist(batch['dft_coords']),
                                   batch['edge_mask'], num_bins, 8)
    return prim + 0.1 * dl


@pytest.mark.parametrize('name', list(gu.MODEL_CASES))
def test_model_case_matches_reference(name):
    cls_name, kwargs, geom = gu.MODEL_CASES[name]
    i = list(gu.MODEL_CASES).index(name)
    model = gu.fill_params(getattr(om, cls_name)(**kwargs).double(), seed=500 + i)
    model.train()
    batch = gu.model_batch(geom, seed=600 + i)
    batch['dist_input'] = batch['dist_input'].double()
    out = model(batch)
    res = {}
    if cls_name == 'TGT_Multi':
        res['gap'], res['logits'] = out
        loss = pretrain_loss(out, batch, kwargs['num_dist_bins'])
    elif cls_name == 'TGT_Distance':
        res['logits'] = out
        loss = core.binned_distance_xent(out, core.pairwise_dist(batch['dft_coords']),
                                         batch['edge_mask'], kwargs['num_dist_bins'], 8)
    else:
        res['gap'] = out
        loss = torch.nn.functional.l1_loss(out, batch['target'])
    assert loss.dtype == torch.float64          # quirk Q9
    res['loss'] = loss
    loss.backward()
    named = dict(model.named_parameters())
    for k in gu.GRAD_PROBE_KEYS:
        if k in named and named[k].grad is not None:
            res['pgrad.' + k] = named[k].grad
    gold = load('model_' + name)
    assert set(res) == set(gold), (sorted(res), sorted(gold))
    for k, t in res.items():
        check(t, gold[k], 1e-10, f'{name}:{k}')


def test_full_width_24L_fp32_forward():
    geom = dict(B=2, N=12, num_nodes=[12, 9])
    model = gu.fill_params(om.TGT_Multi(**gu.FULL_AT_CFG), seed=900)
    model.eval()
    batch = gu.model_batch(geom, seed=901)
    with torch.no_grad():
        gap, logits = model(batch)
    gold = load('model_full_at_24L_fp32')
    # fp32 vs fp32 run of the reference on the same host library: tight
    check(gap, gold['gap'], 1e-4, 'gap')
    check(logits, gold['logits'], 1e-3, 'logits')
    agree = (logits.argmax(-1).numpy() == gold['logits_argmax']['full']).mean()
    assert agree > 0.995


def test_full_width_agx2_12x2_fp32_forward():
    """BASELINE config 1 exactly: TGT-Agx2 12 shared layers x 2 distance predictor, full width, the 8-graph ragged mini-batch
    (reference lib/models/pcqm/distance_predictor.py:9-55)."""
    model = gu.fill_params(om.TGT_Distance(**gu.FULL_AGX2_CFG), seed=920).eval()
    batch = gu.model_batch(gu.FULL_AGX2_GEOM, seed=921)
    with torch.no_grad():
        logits = model(batch)
    gold = load('model_full_agx2_12x2_fp32')
    check(logits, gold['logits'], 1e-3, 'logits')
    assert (logits.argmax(-1).numpy() == gold['logits_argmax']['full']).mean() > 0.995


FULL_AT_GOLDENS = {'n32': ('model_full_at_24L_n32_fp32', gu.FULL_AT_N32_GEOM, (930, 931)),
                   'n48': ('model_full_at_24L_n48_fp32', gu.FULL_AT_N48_GEOM, (940, 941)),
                   # BASELINE config 4 as a mini-batch: 8 ragged graphs, N <= 48, Gaussian 3-D embedding
                   'n48_b8': ('model_full_at_24L_n48_b8_fp32', gu.FULL_AT_N48_B8_GEOM, (960, 961))}


@pytest.mark.parametrize('which', list(FULL_AT_GOLDENS))
def test_full_width_24L_n32_fp32_forward_and_gradients(which):
    """TGT-At 24L at BASELINE widths and the benchmark's node counts (B = 2; N = 32: config 2, N = 48: config 4): eval forward,
    then loss and parameter gradients in train mode with the dropouts off -- the oracle in fp32 against the reference's fp32 run."""
    name, geom, seeds = FULL_AT_GOLDENS[which]
    gold = load(name)
    model = gu.fill_params(om.TGT_Multi(**gu.FULL_AT_CFG), seed=seeds[0])
    batch = gu.model_batch(geom, seed=seeds[1])
    model.eval()
    with torch.no_grad():
        gap, logits = model(batch)
    check(gap, gold['gap'], 1e-4, 'gap')
    check(logits, gold['logits'], 1e-3, 'logits')
    assert (logits.argmax(-1).numpy() == gold['logits_argmax']['full']).mean() > 0.995
    model.train()
    g, l = model(batch)
    loss = torch.nn.functional.l1_loss(g, batch['target']) + 0.1 * core.binned_distance_xent(
        l, core.pairwise_dist(batch['dft_coords']), batch['edge_mask'], 512, 8)
    loss.backward()
    check(loss, gold['loss'], 1e-5, 'loss')
    named = dict(model.named_parameters())
    for k in gu.FULL_GRAD_KEYS:
        check(named[k].grad, gold['pgrad.' + k], 5e-3, k)


def test_full_width_gap_agx2_12x2_forward():
    """BASELINE config 5, second stage: the TGT-Agx2 12 x 2 gap predictor at full width on 8 ragged graphs whose distance input
    went through the bins format -- the oracle against the reference's fp32 forward (lib/models/pcqm/gap_predictor.py:48-63), and
    the bins -> distance step against the reference's BinsProcessor output stored with the golden"""
    gold = load('model_full_gap_agx2_12x2_fp32')
    batch = gu.model_batch(gu.FULL_AGX2_GEOM, seed=951)
    bins = torch.triu((batch['dist_input'].float() * (255 / 8)).long().clamp(0, 255), 1)
    assert torch.equal(core.bins_to_dist(bins, 8 / 255), torch.from_numpy(gold['dist_input']['full']))      # bit for bit
    batch['dist_input'] = gu.binned_dist_input(batch)
    assert torch.equal(batch['dist_input'], torch.from_numpy(gold['dist_input']['full']))
    model = gu.fill_params(om.TGT_Gap(**gu.FULL_GAP_AGX2_CFG), seed=950).eval()
    with torch.no_grad():
        gap = model(batch)
    check(gap, gold['gap'], 1e-4, 'gap')


def test_state_dict_manifest_matches_reference():
    gold = load('misc')
    sd = om.TGT_Multi(**gu.FULL_AT_CFG).state_dict()
    assert list(sd.keys()) == list(gold['manifest_keys']['full'])
    assert [','.join(map(str, v.shape)) for v in sd.values()] == list(gold['manifest_shapes']['full'])


def test_misc_functions():
    gold = load('misc')
    rng = np.random.default_rng(4242)
    coords = torch.from_numpy(rng.standard_normal((2, 5, 3)).astype(np.float32))
    d = core.pairwise_dist(coords)
    np.testing.assert_allclose(d.numpy(), gold['coords2dist']['full'], rtol=1e-6, atol=1e-6)
    logits = torch.from_numpy(rng.standard_normal((2, 5, 5, 16)))
    em = torch.ones(2, 5, 5, dtype=torch.uint8)
    em[1, 3:, :] = 0
    em[1, :, 3:] = 0
    np.testing.assert_allclose(core.binned_distance_xent(logits, d.double(), em, 16, 8).numpy(),
                               gold['xent_reduced']['full'], rtol=1e-12)
    np.testing.assert_allclose(core.binned_distance_xent(logits, d.double(), em, 16, 8, reduce=False).numpy(),
                               gold['xent_per_graph']['full'], rtol=1e-12)
    bins = torch.from_numpy(gold['bins_in']['full'])
    np.testing.assert_allclose(core.bins_to_dist(bins, 8 / 15).numpy(), gold['bins2dist']['full'], rtol=1e-6)


# ---- the two oracle forms against each other (and hence both vs golden) ----
def _rand_core_inputs(B, N, W, C, Hn, Ht, seed):
    rng = np.random.default_rng(seed)
    r = lambda *s: torch.from_numpy(rng.standard_normal(s))
    num_nodes = [N, max(1, N - 2)][:B] + [N] * max(0, B - 2)
    mask = gu.additive_mask(num_nodes, N, torch.float64)
    return dict(qkv=r(B, N, 3 * W), eg=r(B, N, N, 2 * Hn), tq_in=r(B, N, N, 3 * C), te_in=r(B, N, N, 2 * Ht),
                tq_out=r(B, N, N, 3 * C), te_out=r(B, N, N, 2 * Ht), v2=r(B, N, N, 2 * C),
                eg4=r(B, N, N, 4 * Ht), mask=mask)


def test_explicit_numpy_form_agrees_with_torch_form():
    B, N, W, C, Hn, Ht = 2, 5, 24, 16, 4, 2
    x = _rand_core_inputs(B, N, W, C, Hn, Ht, 3)
    m3 = x['mask'][..., 0].numpy()
    v_att, h_hat = core.egt_attention_core(x['qkv'], x['eg'], x['mask'], Hn)
    v2, h2 = explicit_np.egt_attention(x['qkv'].numpy(), x['eg'].numpy(), m3, Hn)
    np.testing.assert_allclose(v_att.numpy(), v2, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(h_hat.numpy(), h2, rtol=1e-12, atol=1e-13)
    va = core.triplet_attention_core(x['tq_in'], x['te_in'], x['tq_out'], x['te_out'], x['mask'], Ht)
    va2 = explicit_np.triplet_attention(x['tq_in'].numpy(), x['te_in'].numpy(), x['tq_out'].numpy(),
                                        x['te_out'].numpy(), m3, Ht)
    np.testing.assert_allclose(va.numpy(), va2, rtol=1e-12, atol=1e-13)
    ag = core.triplet_aggregate_core(x['v2'], x['eg4'], x['mask'], Ht)
    ag2 = explicit_np.triplet_aggregate(x['v2'].numpy(), x['eg4'].numpy(), m3, Ht)
    np.testing.assert_allclose(ag.numpy(), ag2, rtol=1e-12, atol=1e-13)


def test_fully_padded_rows_stay_finite():
    """Q6: padded query rows give a finite uniform softmax and a zero gate."""
    x = _rand_core_inputs(2, 6, 24, 16, 4, 2, 5)
    for k in ('qkv', 'eg', 'tq_in', 'te_in', 'tq_out', 'te_out', 'mask'):
        x[k] = x[k].float()
    x['mask'] = gu.additive_mask([6, 3], 6, torch.float32)
    v_att, h_hat = core.egt_attention_core(x['qkv'], x['eg'], x['mask'], 4)
    va = core.triplet_attention_core(x['tq_in'], x['te_in'], x['tq_out'], x['te_out'], x['mask'], 2)
    assert torch.isfinite(v_att).all() and torch.isfinite(h_hat).all() and torch.isfinite(va).all()
    assert v_att[1, 3:].abs().max() == 0


def test_bf16_drift_fixture_anchors_every_compared_tensor():
    """tests/golden/bf16_drift.npz (the reference's own bf16-autocast drift, tools/make_golden.py drift) holds a
    positive, small rel-L2 for every tensor the bf16 model tests compare"""
    import golden_util as gu
    for name, (cls_name, kwargs, geom) in gu.MODEL_CASES.items():
        d = gu.bf16_drift(name)
        gold = np.load(os.path.join(gu.GOLDEN_DIR, f'model_{name}.npz'))
        for k in {f.rsplit('::', 1)[0] for f in gold.files}:
            assert k in d and 0 < d[k] < 5e-2, (name, k, d.get(k))
    d = gu.bf16_drift('full_at_24L')
    for k in gu.FULL_GRAD_KEYS:
        assert 0 < d['pgrad.' + k] < 5e-2
    assert 0 < d['logits'] < 3e-2


# ---- MC-sampled prediction steps and the bin formats between the two inference stages ----
def _predict_setup():
    from oracle import predict as op
    c = gu.PREDICT_CASE
    z = np.load(os.path.join(gu.GOLDEN_DIR, 'predict.npz'))
    batch = gu.model_batch(dict(B=c['B'], N=c['N'], num_nodes=c['num_nodes']), seed=c['seed'] + 1)

    def replay():
        it = iter(gu.predict_logit_sequence())
        return lambda b: next(it)
    return op, c, z, batch, replay


def test_prediction_bins_and_probs_match_the_reference_scheme():
    op, c, z, batch, replay = _predict_setup()
    bins = op.predict_bins(replay(), batch, c['nb_samples'])
    assert np.array_equal(bins.numpy(), z['bins::full'])                       # integer work: exact
    probs, valid = op.predict_probs(replay(), batch, c['nb_samples'])
    assert valid == c['nb_samples']
    assert np.abs(probs.numpy() - z['probs::full']).max() < 1e-14
    xent = op.eval_xent_from_probs(probs, core.pairwise_dist(batch['dft_coords']), batch['edge_mask'],
                                   c['num_bins'], c['range_bins'])
    assert np.abs(xent.numpy() - z['eval_xent::full']).max() < 1e-12
    with pytest.raises(ValueError):                                             # every sample NaN
        nan = torch.full((1, 2, 2, 4), float('nan'))
        op.predict_bins(lambda b: nan, {}, 2)


def test_prediction_bin_packing_and_distances_match_the_reference():
    op, c, z, batch, replay = _predict_setup()
    bins = torch.from_numpy(z['bins::full'])
    saved = op.save_bins_step(bins, c['num_nodes'], c['num_bins'])
    assert all(s.dtype == np.uint8 for s in saved)
    assert np.array_equal(np.concatenate(saved), z['saved_bins_flat::full'])
    assert [len(s) for s in saved] == z['saved_bins_lengths::full'].tolist()
    N = c['N']
    dist_bins = np.zeros((c['B'], c['nb_samples'], N, N), dtype=np.float32)
    for i, n in enumerate(c['num_nodes']):
        dist_bins[i, :, :n, :n] = op.unpack_bins_multi(saved[i].reshape(c['nb_samples'], -1), n).astype(np.float32)
    assert np.array_equal(dist_bins, z['dist_bins::full'])
    d = core.bins_to_dist(torch.from_numpy(dist_bins), c['range_bins'] / (c['num_bins'] - 1))
    assert np.array_equal(d.numpy(), z['dist_input::full'])                    # same float32 operations: exact
    assert op.bins_storage_dtype(256) == np.uint8 and op.bins_storage_dtype(512) == np.uint16


def test_gap_prediction_step_matches_the_reference_scheme():
    op, c, z, batch, replay = _predict_setup()
    b = dict(batch)
    b['dist_input'] = torch.from_numpy(z['dist_input::full'])
    pred = op.gap_prediction_step(gu.predict_gap_sequence(), b, 4)
    assert np.array_equal(pred.numpy(), z['gap_pred::full'])
    assert abs(op.evaluate_gap(pred.numpy(), batch['target'].numpy()) - float(z['gap_mae::full'])) < 1e-12
