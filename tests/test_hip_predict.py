"""MC-sampled prediction steps on the device (tgt_amd/pcqm/predict.py, csrc/predict.hip) against the reference's
golden outputs (tests/golden/predict.npz: its own scheme methods) and the oracle restatement (oracle/predict.py)."""
import os

import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import core, predict as op, modules as om

pytestmark = pytest.mark.gpu


def _case():
    c = gu.PREDICT_CASE
    z = np.load(os.path.join(gu.GOLDEN_DIR, 'predict.npz'))
    batch = {k: v.cuda() for k, v in gu.model_batch(dict(B=c['B'], N=c['N'], num_nodes=c['num_nodes']), seed=c['seed'] + 1).items()}
    batch['num_nodes'] = torch.tensor(c['num_nodes']).cuda()
    batch['idx'] = torch.arange(100, 100 + c['B']).cuda()
    return c, z, batch


def _replay(dtype=torch.float32):
    it = iter(gu.predict_logit_sequence(dtype=dtype))
    return lambda b: next(it).cuda()


def test_predict_bins_matches_the_reference_scheme_and_skips_the_nan_sample():
    from tgt_amd.pcqm import predict as pp
    c, z, batch = _case()
    bins = pp.predict_bins(_replay(), batch, c['nb_samples'])
    assert bins.dtype == torch.uint8 and bins.shape == (c['B'], c['nb_samples'], c['N'], c['N'])
    assert np.array_equal(bins.cpu().numpy(), z['bins::full'])                 # index work: exact
    # 16-bit logits (what an autocast model emits): against the oracle on the SAME rounded logits
    for dt in (torch.float16, torch.bfloat16):
        seq = gu.predict_logit_sequence(dtype=dt)
        it = iter(seq)
        want = op.predict_bins(lambda b: next(it).float(), {}, c['nb_samples'])
        got = pp.predict_bins(_replay(dt), batch, c['nb_samples'])
        assert np.array_equal(got.cpu().numpy(), want.numpy())
    with pytest.raises(ValueError):                                             # every sample NaN: 2S tries, then the error
        pp.predict_bins(lambda b: torch.full((2, 3, 3, 8), float('nan'), device='cuda'), batch, 2)


def test_predict_probs_and_eval_loss_match_the_reference_scheme():
    from tgt_amd.pcqm import predict as pp
    c, z, batch = _case()
    probs = pp.predict_probs(_replay(), batch, c['nb_samples'])
    assert probs.dtype == torch.float32
    assert np.abs(probs.cpu().numpy() - z['probs::full']).max() < 2e-7          # float32 softmax vs the float64 golden
    res = pp.prediction_step4eval(_replay(), batch, c['nb_samples'], c['num_bins'], c['range_bins'])
    assert np.abs(res['loss'].cpu().numpy() - z['eval_xent::full']).max() < 2e-5 * np.abs(z['eval_xent::full']).max()


def test_pack_bins_and_bins2dist_are_bit_exact():
    from tgt_amd.pcqm import predict as pp
    c, z, batch = _case()
    bins = torch.from_numpy(z['bins::full']).to(torch.uint8).cuda()
    flat, off = pp.pack_bins(bins, batch['num_nodes'])
    assert np.array_equal(flat.cpu().numpy(), z['saved_bins_flat::full'])
    assert np.diff(off.cpu().numpy()).tolist() == z['saved_bins_lengths::full'].tolist()
    saved = pp.prediction_step4savebins(_replay(), batch, c['nb_samples'])
    assert np.array_equal(saved['idx'], z['saved_idx::full'])
    assert np.array_equal(np.concatenate(saved['bins']), z['saved_bins_flat::full'])
    # the gap stage's input: from the unpacked float32 bins (what the dataloader delivers) ...
    bs = c['range_bins'] / (c['num_bins'] - 1)
    d = pp.bins2dist(torch.from_numpy(z['dist_bins::full']).cuda(), bs)
    assert np.array_equal(d.cpu().numpy(), z['dist_input::full'])
    # ... and straight from the distance stage's dense bins on the device (pack + unpack as a mask)
    d2 = pp.bins2dist(bins, bs, num_nodes=batch['num_nodes'])
    assert np.array_equal(d2.cpu().numpy(), z['dist_input::full'])
    for dt in (torch.int32, torch.int64, torch.uint16):
        assert np.array_equal(pp.bins2dist(bins.to(dt), bs, num_nodes=batch['num_nodes']).cpu().numpy(), z['dist_input::full'])
    # no shift / diagonal kept: the oracle's restatement of the other switches
    raw = torch.from_numpy(z['dist_bins::full'])
    want = core.bins_to_dist(raw, bs, shift_half=False, zero_diag=False)
    assert np.array_equal(pp.bins2dist(raw.cuda(), bs, shift_half=False, zero_diag=False).cpu().numpy(), want.numpy())


@pytest.mark.parametrize('N,B,S,nb', [(32, 5, 3, 256), (48, 3, 2, 512), (9, 4, 4, 24)])
def test_pack_bins_ragged_against_the_oracle(N, B, S, nb):
    from tgt_amd.pcqm import predict as pp
    rng = np.random.default_rng(N + B)
    nn_ = rng.integers(2, N + 1, B)
    nn_[0], nn_[-1] = N, 2
    dt = pp.bins_dtype(nb)
    bins = torch.from_numpy(rng.integers(0, nb, (B, S, N, N)).astype(np.uint16 if nb > 256 else np.uint8))
    want = op.save_bins_step(bins, nn_, nb)
    flat, off = pp.pack_bins(bins.cuda(), torch.from_numpy(nn_))
    assert flat.dtype == dt
    flat, off = flat.cpu().numpy(), off.cpu().numpy()
    for i in range(B):
        assert np.array_equal(flat[off[i]:off[i + 1]], want[i])


def test_gap_prediction_step_matches_the_reference_scheme():
    from tgt_amd.pcqm import predict as pp
    c, z, batch = _case()
    b = dict(batch)
    b['dist_input'] = torch.from_numpy(z['dist_input::full']).cuda()
    res = pp.gap_prediction_step(gu.predict_gap_sequence(), b, 4)
    assert res['gap_pred'].shape == (c['B'], 4)
    assert np.abs(res['gap_pred'].cpu().numpy() - z['gap_pred::full']).max() < 1e-5 * np.abs(z['gap_pred::full']).max()
    mae = pp.evaluate_gap_predictions(res['gap_pred'], batch['target'])
    assert abs(mae - float(z['gap_mae::full'])) < 1e-5 * abs(float(z['gap_mae::full']))


def test_bins_round_trip_through_the_parquet_files(tmp_path):
    """save (per-rank parquet + meta.json, dist_pred/scheme.py:256-305) -> load -> unpack -> the batch's dist_bins"""
    from tgt_amd.pcqm import predict as pp
    c, z, batch = _case()
    saved = pp.prediction_step4savebins(_replay(), batch, c['nb_samples'])
    pp.save_bins(str(tmp_path), 'valid', 0, [saved], c['num_bins'], c['range_bins'], c['nb_samples'])
    meta, rows = pp.load_bins(str(tmp_path))
    assert meta == dict(num_bins=c['num_bins'], range_bins=c['range_bins'], num_samples=c['nb_samples'])
    for i, n in enumerate(c['num_nodes']):
        packed = rows[100 + i].reshape(c['nb_samples'], -1)
        assert np.array_equal(op.unpack_bins_multi(packed, n).astype(np.float32), z['dist_bins::full'][i, :, :n, :n])


def test_two_stage_inference_on_models_fp16():
    """BASELINE config 5 shape class end to end: TGT-Agx2 distance predictor -> sampled bins -> bins2dist -> gap
    predictor, dropout ON (predict_in_train), fp16 autocast; the device pipeline against the oracle pipeline driven
    with the SAME logits (stage hand-off exact) and the same gap stand-in; then on the real HIP models: finite, and
    sample v of the gap stage sees bins sample v."""
    from tgt_amd.pcqm import TGT_Distance, TGT_Gap, predict as pp
    dk = dict(gu.MODEL_CASES['dist_agx2_tiny'][1])
    gk = dict(gu.MODEL_CASES['gap_at_tiny'][1])
    gk.update(embed_3d_type='gaussian', triplet_type='aggregate', layer_multiplier=2)
    dk.update(source_dropout=0.3, drop_path=0.1, node_act_dropout=0.1, edge_act_dropout=0.1)
    geom = dict(B=4, N=9, num_nodes=[9, 6, 9, 4])
    cpu = gu.model_batch(geom, seed=31)
    cpu['num_nodes'] = torch.tensor(geom['num_nodes'])
    batch = {k: v.cuda() for k, v in cpu.items()}
    d_hip = gu.fill_params(TGT_Distance(**dk), seed=32).cuda().train()
    g_hip = gu.fill_params(TGT_Gap(**gk), seed=33).cuda().train()
    S, nbins = 3, dk['num_dist_bins']
    bins, gap = pp.two_stage_predict(d_hip, g_hip, batch, S, nbins, 8, autocast_dtype=torch.float16)
    assert bins.shape == (4, S, 9, 9) and gap.shape == (4, S) and torch.isfinite(gap).all()
    assert (bins == bins.transpose(-1, -2)).all()                               # symmetrised probabilities
    assert len({bins[:, s].cpu().numpy().tobytes() for s in range(S)}) > 1      # dropout is on: samples differ
    # stage hand-off vs the oracle: same bins -> same distances -> (eval-mode, deterministic) gap model agrees
    bn = bins.cpu().numpy()
    dense = np.zeros((4, S, 9, 9), dtype=np.float32)
    for b, n in enumerate(geom['num_nodes']):                                  # save -> load: pack, unpack, zero padding
        dense[b, :, :n, :n] = op.unpack_bins_multi(op.pack_bins_multi(bn[b][:, :n, :n]), n)
    d_want = core.bins_to_dist(torch.from_numpy(dense), 8 / (nbins - 1))
    d_got = pp.bins2dist(bins, 8 / (nbins - 1), num_nodes=batch['num_nodes'])
    assert np.array_equal(d_got.cpu().numpy(), d_want.numpy())
    g_ref = gu.fill_params(om.TGT_Gap(**gk), seed=33).eval()
    g_hip.eval()
    with torch.no_grad():
        for s in range(S):
            b2 = dict(cpu)
            b2['dist_input'] = d_want[:, s]
            want = g_ref(b2)
            hb = dict(batch)
            hb['dist_input'] = d_got[:, s]
            with torch.autocast('cuda', dtype=torch.float16):
                got = g_hip(hb)
            assert (got.float().cpu() - want).abs().max() < 3e-2 * want.abs().max()


def test_prediction_loop_restores_mode_and_runs_without_autograd():
    from tgt_amd.pcqm import predict as pp
    m = torch.nn.Dropout(0.5).eval()
    seen = []

    def step(b):
        seen.append((m.training, torch.is_grad_enabled()))
        return b
    assert pp.prediction_loop(m, [1, 2], step, predict_in_train=True) == [1, 2]
    assert seen == [(True, False), (True, False)] and not m.training


def test_cpu_tensors_raise():
    from tgt_amd.pcqm import predict as pp
    with pytest.raises(RuntimeError):
        pp.bins2dist(torch.zeros(2, 3, 3), 0.1)
    with pytest.raises(RuntimeError):
        pp.predict_bins(lambda b: torch.zeros(1, 2, 2, 8), {}, 1)


def test_hipgraph_replay_of_the_forward_is_bit_identical():
    """tgt_amd/pcqm/graphed.py: one captured forward replayed on new inputs == the eager forward, bit for bit"""
    from tgt_amd.pcqm import TGT_Distance
    from tgt_amd.pcqm.graphed import GraphedForward
    dk = dict(gu.MODEL_CASES['dist_agx2_tiny'][1])
    geom = dict(B=3, N=7, num_nodes=[7, 5, 3])
    model = gu.fill_params(TGT_Distance(**dk), seed=32).cuda().eval()
    b0 = {k: v.cuda() for k, v in gu.model_batch(geom, seed=41).items()}
    b1 = {k: v.cuda() for k, v in gu.model_batch(geom, seed=42).items()}
    for dt in (None, torch.bfloat16):
        gf = GraphedForward(model, b0, autocast_dtype=dt)
        for b in (b1, b0):
            with torch.no_grad(), torch.autocast('cuda', dtype=dt or torch.bfloat16, enabled=dt is not None):
                want = model(b)
            assert torch.equal(gf(b), want)
    with pytest.raises(RuntimeError):
        gf({k: (v[:2] if k == 'node_mask' else v) for k, v in b0.items()})
    # a missing / extra key must not silently replay the example batch's data
    with pytest.raises(RuntimeError, match='keys'):
        gf({k: v for k, v in b0.items() if k != 'node_mask'})
    with pytest.raises(RuntimeError, match='keys'):
        gf(dict(b0, extra=b0['node_mask']))
    # a train-mode model with dropout would replay ONE captured drop pattern (S identical "Monte-Carlo samples"): refused
    dk2 = dict(dk, source_dropout=0.3, drop_path=0.1, edge_act_dropout=0.1)
    noisy = gu.fill_params(TGT_Distance(**dk2), seed=32).cuda().train()
    with pytest.raises(RuntimeError, match='train mode'):
        GraphedForward(noisy, b0)
    GraphedForward(noisy.eval(), b0)
