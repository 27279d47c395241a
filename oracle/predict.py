"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU restatement of the reference's MC-sampled prediction steps and of the bin
formats between its two inference stages (paths relative to
/root/reference):

  predict_bins            lib/training_schemes/pcqm/dist_pred/scheme.py:181-205
  predict_probs           lib/training_schemes/pcqm/dist_pred/scheme.py:139-167
  eval_xent_from_probs    lib/training_schemes/pcqm/dist_pred/scheme.py:170-179
  save_bins_step          lib/training_schemes/pcqm/dist_pred/scheme.py:208-229
  gap_prediction_step     lib/training_schemes/pcqm/gap_pred/scheme.py:78-110
  evaluate_gap            lib/training_schemes/pcqm/gap_pred/scheme.py:116-135
  flat_triu_indices / pack_bins_multi / unpack_bins_multi
                          lib/data/pcqm/bin_ops.py:5-46
  (bins -> distances is oracle.core.bins_to_dist, commons.py:72-82)

`model` is any callable batch -> tensor (the reference calls self.model with
dropout ON: predict_in_train=True, tgt_training.py:42).

Pinned: tests/test_oracle_golden.py::test_prediction_* against
tests/golden/predict.npz, produced by running the reference's own scheme
methods (tools/make_golden.py predict).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import core


def _finite(t):
    return not (torch.isnan(t).any() or torch.isinf(t).any())


def predict_bins(model, batch, nb_samples):
    """(B, S, N, N) argmax bins of the symmetrised bin probabilities of S valid stochastic forwards; a forward with a
    NaN/Inf logit is skipped (at most 2S tries), fewer than S valid ones is an error."""
    bins = []
    for _ in range(nb_samples * 2):
        logits = model(batch)
        if not _finite(logits):
            continue
        p = torch.softmax(logits, dim=-1)
        p = p + p.transpose(-2, -3)
        bins.append(p.argmax(dim=-1))
        if len(bins) >= nb_samples:
            break
    if len(bins) < nb_samples:
        raise ValueError(f'{nb_samples - len(bins)}/{nb_samples} predictions were NaN')
    return torch.stack(bins, dim=1)


def predict_probs(model, batch, nb_samples):
    """mean over the valid samples of the softmax, symmetrised over the pair axes; (probs, number of valid samples)"""
    probs, valid = None, 0
    for _ in range(nb_samples * 2):
        logits = model(batch)
        if not _finite(logits):
            continue
        p = F.softmax(logits, dim=-1)
        probs = p if probs is None else probs + p
        valid += 1
        if valid >= nb_samples:
            break
    if not valid:
        raise ValueError('All predictions were NaN')
    probs = probs + probs.transpose(-2, -3)
    return probs / (valid * 2), valid


def eval_xent_from_probs(probs, dist_target, edge_mask, num_bins, range_bins):
    """per-graph cross entropy of log(probs + 1e-9) (prediction_step4eval)"""
    return core.binned_distance_xent(torch.log(probs + 1e-9), dist_target, edge_mask, num_bins, range_bins, reduce=False)


def flat_triu_indices(n):
    """flat indices i*n + j of the strict upper triangle, row by row"""
    i, j = np.triu_indices(n, 1)
    return (i * n + j).astype(np.int64)


def pack_bins_multi(bins):
    """(S, n, n) -> (S, n(n-1)/2)"""
    s, n, _ = bins.shape
    return bins.reshape(s, n * n)[:, flat_triu_indices(n)]


def unpack_bins_multi(packed, n):
    """(S, n(n-1)/2) -> (S, n, n), zeros on and below the diagonal"""
    s = packed.shape[0]
    m = np.zeros((s, n * n), dtype=packed.dtype)
    m[:, flat_triu_indices(n)] = packed
    return m.reshape(s, n, n)


def bins_storage_dtype(num_dist_bins):
    return np.uint8 if num_dist_bins <= 256 else (np.uint16 if num_dist_bins <= 65536 else np.int64)


def save_bins_step(bins, num_nodes, num_dist_bins):
    """per graph: the (S, n, n) block of its real nodes, narrowed to uint8/uint16, triu-packed and flattened"""
    b = bins.cpu().numpy().astype(bins_storage_dtype(num_dist_bins))
    return [pack_bins_multi(b[i, :, :n, :n]).reshape(-1) for i, n in enumerate(np.asarray(num_nodes).tolist())]


def gap_prediction_step(model, batch, nb_samples):
    """(B, S_valid) gap predictions; sample v is run on dist_input[:, v % num_dist_inputs]; NaN/Inf samples skipped"""
    all_d = batch['dist_input']
    assert all_d.ndim == 4
    preds = []
    for _ in range(nb_samples * 2):
        b = dict(batch)
        b['dist_input'] = all_d[:, len(preds) % all_d.size(1)]
        g = model(b)
        if not _finite(g):
            continue
        preds.append(g)
        if len(preds) >= nb_samples:
            break
    if not preds:
        raise ValueError('All predictions were NaN')
    return torch.stack(preds, dim=-1)


def evaluate_gap(gap_pred, gap_target):
    """mean absolute error of the sample mean"""
    return np.abs(np.mean(np.asarray(gap_pred), axis=-1) - np.asarray(gap_target)).mean()
