"""MC-sampled prediction steps of the PCQM schemes and the bin format between the two inference stages, on the
device (kernels: csrc/predict.hip).

Mirrors, with the reference's names and error behaviour (paths relative to /root/reference):
  predict_bins / predict_probs / prediction_step4eval / prediction_step4savebins
                         lib/training_schemes/pcqm/dist_pred/scheme.py:139-229
  gap_prediction_step / evaluate_gap_predictions
                         lib/training_schemes/pcqm/gap_pred/scheme.py:78-135
  pack_bins / bins2dist  lib/data/pcqm/bin_ops.py:32-46, lib/training_schemes/pcqm/commons.py:72-82
  prediction_loop        lib/training/training.py:700-722 (model.train() when predict_in_train: dropout stays ON)
  save_bins / load_bins  dist_pred/scheme.py:256-305 (per-rank parquet `idx`, flat `bins` + meta.json), data.py:215-239

Difference in mechanics, not in results: the reference asks the host after EVERY stochastic forward whether the sample
was finite; here that accept/skip decision is device state (tgt_sample_commit), and the host reads the sample count once
after the S forwards it certainly needs (further tries only while samples are missing, at most 2S in total).
No CPU path: CPU tensors raise.
"""
import json
import os

import numpy as np
import torch

from .. import _lib, ops
from ..ops import _DT, _dev, _ptr, _stream


def _state(device):
    return torch.zeros(4, dtype=torch.int32, device=device)       # {valid, nonfinite, tries, -}


def bins_dtype(num_dist_bins):
    """storage type of saved bins (dist_pred/scheme.py:215-218)"""
    if num_dist_bins <= 256:
        return torch.uint8
    if num_dist_bins <= 65536:
        return torch.uint16
    raise ValueError('more than 65536 distance bins')


def _sample_loop(nb_samples, one_try, state, what, need_all):
    """S tries, one read of the valid count, then one try at a time while samples are missing (<= 2S tries in all)."""
    tries = 0
    for _ in range(nb_samples):
        one_try()
        tries += 1
    valid = int(state[0])
    while valid < nb_samples and tries < 2 * nb_samples:
        one_try()
        tries += 1
        valid = int(state[0])
    if need_all and valid < nb_samples:
        raise ValueError(f'{nb_samples - valid}/{nb_samples} predictions were NaN')
    if not valid:
        raise ValueError('All predictions were NaN')
    if valid < nb_samples:
        print(f'Warning: {nb_samples - valid}/{nb_samples} predictions were NaN')
    return valid


def predict_bins(model, batch, nb_samples):
    """(B, S, N, N) bins of S valid stochastic forwards of a distance predictor: argmax of the softmax symmetrised over
    the pair axes (dist_pred/scheme.py:181-205).  uint8 for <= 256 bins, else uint16 (the reference's saved types)."""
    L = _lib.lib()
    st, out = None, None

    def one_try():
        nonlocal st, out
        logits = model(batch)
        _dev(logits)
        logits = logits.contiguous()
        B, N, _, NB = logits.shape
        if out is None:
            st = _state(logits.device)
            out = torch.empty(B, nb_samples, N, N, dtype=bins_dtype(NB), device=logits.device)
        _lib.check(L.tgt_dist_bins_argmax(_ptr(logits), _DT[logits.dtype], B, N, NB, _ptr(out), out.element_size(),
                                          nb_samples * N * N, nb_samples, _ptr(st), _stream()), 'tgt_dist_bins_argmax')
        _lib.check(L.tgt_sample_commit(_ptr(st), nb_samples, _stream()), 'tgt_sample_commit')

    _sample_loop(nb_samples, one_try, _LazyState(lambda: st), 'bins', need_all=True)
    return out


class _LazyState:
    """the state tensor exists only after the first forward (its device is the logits')"""

    def __init__(self, get):
        self._get = get

    def __getitem__(self, i):
        return self._get()[i]


def predict_probs(model, batch, nb_samples, as_log_eps=None):
    """mean symmetrised bin probabilities over the valid samples, (B,N,N,bins) float32 (dist_pred/scheme.py:139-167);
    as_log_eps: return log(probs + eps) instead (what prediction_step4eval feeds the cross entropy, :173)"""
    L = _lib.lib()
    st, acc = None, None

    def one_try():
        nonlocal st, acc
        logits = model(batch)
        _dev(logits)
        logits = logits.contiguous()
        if acc is None:
            st = _state(logits.device)
            acc = torch.zeros(logits.shape, dtype=torch.float32, device=logits.device)
        NB = logits.shape[-1]
        _lib.check(L.tgt_softmax_accumulate(_ptr(logits), _DT[logits.dtype], logits.numel() // NB, NB, _ptr(acc), _ptr(st),
                                            nb_samples, _stream()), 'tgt_softmax_accumulate')
        _lib.check(L.tgt_sample_commit(_ptr(st), nb_samples, _stream()), 'tgt_sample_commit')

    _sample_loop(nb_samples, one_try, _LazyState(lambda: st), 'probs', need_all=False)
    B, N, _, NB = acc.shape
    out = torch.empty_like(acc)
    _lib.check(L.tgt_probs_finish(_ptr(acc), B, N, NB, _ptr(st), 0 if as_log_eps is None else 1,
                                  float(as_log_eps or 0.0), _ptr(out), _stream()), 'tgt_probs_finish')
    return out


def prediction_step4eval(model, batch, nb_samples, num_dist_bins, range_dist_bins=8):
    """per-graph cross entropy of log(mean probs + 1e-9) against the binned DFT distances (dist_pred/scheme.py:170-179)"""
    from ..training.step import coords2dist
    logp = predict_probs(model, batch, nb_samples, as_log_eps=1e-9)
    B = logp.shape[0]
    target = (coords2dist(batch['dft_coords']) * ((num_dist_bins - 1) / range_dist_bins)).long().clamp_(0, num_dist_bins - 1)
    xent = ops.cross_entropy_rows(logp.view(-1, num_dist_bins), target.view(-1)).view(B, -1)
    m = batch['edge_mask'].to(xent.dtype).view(B, -1)
    return dict(loss=(xent * m).sum(1) / (m.sum(1) + 1e-9))


def pack_bins(bins, num_nodes):
    """triu-pack the bins of every graph's real nodes on the device (bin_ops.py:32-37 per graph, dist_pred/scheme.py:221-226).
    bins (B,S,N,N) uint8/uint16/int32, num_nodes (B).  Returns (flat, offsets): graph b owns flat[offsets[b]:offsets[b+1]]
    = its (S, n_b(n_b-1)/2) block, row-major -- exactly the reference's `packed_bins_i.reshape(-1)`."""
    _dev(bins)
    B, S, N, _ = bins.shape
    nn_ = num_nodes.to(bins.device, torch.int64).contiguous()
    per = S * (nn_ * (nn_ - 1) // 2)
    offsets = torch.zeros(B + 1, dtype=torch.int64, device=bins.device)
    torch.cumsum(per, 0, out=offsets[1:])
    total = int(offsets[-1])                                            # the one sync: the output's size
    flat = torch.empty(total, dtype=bins.dtype, device=bins.device)
    bins = bins.contiguous()
    _lib.check(_lib.lib().tgt_pack_triu(_ptr(bins), bins.element_size(), B, S, N, _ptr(nn_), _ptr(offsets), _ptr(flat), total,
                                        _stream()), 'tgt_pack_triu')
    return flat, offsets


def prediction_step4savebins(model, batch, nb_samples):
    """dict(idx, bins): per graph the flat packed bins (numpy), as the reference hands them to pyarrow (:208-229)"""
    bins = predict_bins(model, batch, nb_samples)
    flat, offsets = pack_bins(bins, batch['num_nodes'])
    flat, off = flat.cpu().numpy(), offsets.cpu().numpy()
    return dict(idx=batch['idx'].cpu().numpy(), bins=[flat[off[i]:off[i + 1]] for i in range(len(off) - 1)])


_KIND = {torch.uint8: _lib.BINS_U8, torch.uint16: _lib.BINS_U16, torch.int32: _lib.BINS_I32, torch.int64: _lib.BINS_I64,
         torch.float32: _lib.BINS_F32}


def bins2dist(bins, bin_size, shift_half=True, zero_diag=True, num_nodes=None):
    """`BinsProcessor.bins2dist` (commons.py:72-82) on the device, bit for bit: bins (..., N, N) -> float32 distances.
    num_nodes (B) with bins (B,S,N,N): first reduce the bins to the strict upper triangle of each graph's real nodes --
    what saving them (pack) and loading them into the zero-padded batch (unpack, bin_ops.py:39-46) does -- so that the
    distance stage's output can feed the gap stage without leaving the device."""
    _dev(bins)
    if bins.dtype not in _KIND:
        raise RuntimeError(f'bins2dist: bins of {bins.dtype}; uint8, uint16, int32, int64 or float32 expected')
    bins = bins.contiguous()
    N = bins.shape[-1]
    R = bins.numel() // (N * N)
    out = torch.empty(bins.shape, dtype=torch.float32, device=bins.device)
    nn_, S = None, 0
    if num_nodes is not None:
        assert bins.ndim == 4 and num_nodes.numel() == bins.shape[0]
        nn_, S = num_nodes.to(bins.device, torch.int64).contiguous(), bins.shape[1]
    _lib.check(_lib.lib().tgt_bins_to_dist(_ptr(bins), _KIND[bins.dtype], R, N, _ptr(nn_), S, float(bin_size), int(shift_half),
                                           int(zero_diag), _ptr(out), _stream()), 'tgt_bins_to_dist')
    return out


def gap_prediction_step(model, batch, nb_samples):
    """dict(idx, gap_pred (B, valid samples) float32, gap_target): sample v of the gap predictor runs on
    dist_input[:, v % num_dist_inputs]; NaN/Inf samples are skipped (gap_pred/scheme.py:78-110)."""
    all_d = batch['dist_input']
    assert all_d.ndim == 4
    _dev(all_d)
    L = _lib.lib()
    st = _state(all_d.device)
    B, n_in = all_d.shape[0], all_d.shape[1]
    out = torch.zeros(B, nb_samples, dtype=torch.float32, device=all_d.device)
    b = dict(batch)

    def one_try():
        slot = (st[0:1] % n_in).long()                                  # device-side `valid_samples % num_dist_inputs`
        b['dist_input'] = all_d.index_select(1, slot).squeeze(1)
        g = model(b)
        g = (g if g.dtype in _DT else g.float()).contiguous()      # (a float64 head: the kernel reads f32/bf16/f16)
        _lib.check(L.tgt_gap_commit(_ptr(g), _DT[g.dtype], B, _ptr(out), nb_samples, _ptr(st), _stream()), 'tgt_gap_commit')

    valid = _sample_loop(nb_samples, one_try, st, 'gap', need_all=False)
    res = dict(gap_pred=out[:, :valid], gap_target=batch['target'])
    if 'idx' in batch:
        res['idx'] = batch['idx']
    return res


def evaluate_gap_predictions(gap_pred, gap_target):
    """MAE of the sample mean (gap_pred/scheme.py:116-135)"""
    return float((gap_pred.double().mean(-1) - gap_target.double()).abs().mean())


def prediction_loop(model, batches, prediction_step, predict_in_train=True):
    """reference training.py:700-722: dropout stays ON when predict_in_train (the schemes' default, tgt_training.py:42);
    no autograd; one prediction_step(batch) per (already preprocessed, device-resident) batch."""
    was_training = model.training
    model.train(predict_in_train)
    try:
        with torch.no_grad():
            return [prediction_step(b) for b in batches]
    finally:
        model.train(was_training)


def two_stage_predict(dist_model, gap_model, batch, nb_samples, num_dist_bins, range_dist_bins=8, autocast_dtype=None):
    """BASELINE config 5 end to end on the device: S sampled bin matrices from the distance predictor
    (predict_bins), turned into the gap predictor's `dist_input` exactly as a save / load round trip through the
    packed-bins files would (triu of the real nodes, bins2dist with the +0.5 shift, commons.py:72-82), then S sampled
    gap predictions (sample v on bins sample v).  Returns (bins (B,S,N,N), gap_pred (B,S) float32)."""
    ctx = torch.autocast('cuda', dtype=autocast_dtype) if autocast_dtype is not None else torch.autocast('cuda', enabled=False)
    with torch.no_grad(), ctx:
        bins = predict_bins(dist_model, batch, nb_samples)
        b2 = dict(batch)
        b2['dist_input'] = bins2dist(bins, range_dist_bins / (num_dist_bins - 1), num_nodes=batch['num_nodes'])
        res = gap_prediction_step(gap_model, b2, nb_samples)
    return bins, res['gap_pred']


# ---- the on-disk format between the stages (dist_pred/scheme.py:256-305, data.py:215-239) ---------------------------
def save_bins(save_dir, dataset_name, rank, outputs, num_dist_bins, range_dist_bins, nb_samples):
    """outputs: list of prediction_step4savebins results.  Writes `<save_dir>/data/<dataset>_<rank:03d>.parquet`
    (columns idx, bins) and, on rank 0, `<save_dir>/meta.json` -- the files lib/data/pcqm/data.py::Bins reads."""
    import pyarrow as pa
    import pyarrow.parquet as pq
    data_dir = os.path.join(save_dir, 'data')
    os.makedirs(data_dir, exist_ok=True)
    table = pa.Table.from_pydict(dict(idx=np.concatenate([o['idx'] for o in outputs]),
                                      bins=[b for o in outputs for b in o['bins']]))
    path = os.path.join(data_dir, f'{dataset_name}_{rank:03d}.parquet')
    pq.write_table(table, path)
    if rank == 0:
        with open(os.path.join(save_dir, 'meta.json'), 'w') as f:
            json.dump(dict(num_bins=num_dist_bins, range_bins=range_dist_bins, num_samples=nb_samples), f)
    return path


def load_bins(save_dir):
    """(meta, {idx: flat packed bins}) of a directory written by save_bins / by the reference"""
    import pyarrow.dataset as pds
    with open(os.path.join(save_dir, 'meta.json')) as f:
        meta = json.load(f)
    t = pds.dataset(os.path.join(save_dir, 'data')).to_table().sort_by('idx')
    return meta, {int(i): np.asarray(b) for i, b in zip(t['idx'].to_pylist(), t['bins'].to_pylist())}
