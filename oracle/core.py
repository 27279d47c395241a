"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU restatement, in plain torch, of the arithmetic of the TGT hot path:
node attention with edge bias/gate, the triplet edge updates, and the small
pieces either side of them.  Every function cites the reference lines it
follows (paths relative to /root/reference).  Works in float32 or float64 and
is differentiable, so autograd of these functions is the gradient oracle.

Pinned: `tests/test_oracle_golden.py` checks these functions against
`tests/golden/*.npz`, which were produced by importing the real reference in
the build container (`tools/make_golden.py`).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may
import this package.
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# channel <-> (dot_dim, head) split.  The reference views the channel axis as
# (D, H) with the HEAD index minor: c = d*H + h
# (lib/tgt/layers/layers.py:62-64, lib/tgt/layers/triplet.py:213-215).
# --------------------------------------------------------------------------
def heads_minor(x, num_heads):
    return x.reshape(*x.shape[:-1], x.shape[-1] // num_heads, num_heads)


def degree_scaler(gates):
    """log(1 + sum_m gates)   lib/tgt/layers/layers.py:8-12"""
    return torch.log(1 + gates.sum(dim=2, keepdim=True))


def egt_attention_core(qkv, eg, mask, num_heads, scale_degree=True,
                       want_nodes=True):
    """Node attention with edge bias and gate.  lib/tgt/layers/layers.py:52-82.

    qkv : (B,N,3W) output of lin_QKV     eg : (B,N,N,2H) output of lin_EG
    mask: (B,N,N,1) additive mask (0 / finfo.min, possibly + source-drop mask)
    returns V_att (B,N,W) [channel = d*H+h] and H_hat (B,N,N,H) (pre-mask,
    pre-softmax logits that feed lin_O_e, layers.py:82).
    """
    B, N, W3 = qkv.shape
    W = W3 // 3
    D = W // num_heads
    q, k, v = (heads_minor(t, num_heads) for t in qkv.split(W, dim=-1))
    e_bias, g_logit = eg.split(num_heads, dim=-1)
    # H_hat[b,l,m,h] = s * sum_d Q[b,l,d,h] K[b,m,d,h] + E[b,l,m,h]   (:66,:69)
    h_hat = torch.einsum('bldh,bmdh->blmh', q, k) * (D ** -0.5) + e_bias
    if not want_nodes:
        return None, h_hat
    gates = torch.sigmoid(g_logit + mask)                              # :68
    att = torch.softmax(h_hat + mask, dim=2) * gates                   # :70
    # V_att[b,l,d,h] = sum_m att[b,l,m,h] V[b,m,d,h]                    # :71
    v_att = torch.einsum('blmh,bmdh->bldh', att, v)
    if scale_degree:                                                   # :73-75
        v_att = v_att * degree_scaler(gates)
    return v_att.reshape(B, N, W), h_hat


def edge_update_core(qk, e_bias, num_heads):
    """lib/tgt/layers/layers.py:116-127 (EdgeUpdate: logits only)."""
    B, N, W2 = qk.shape
    W = W2 // 2
    D = W // num_heads
    q, k = (heads_minor(t, num_heads) for t in qk.split(W, dim=-1))
    return torch.einsum('bldh,bmdh->blmh', q, k) * (D ** -0.5) + e_bias


def _triplet_dir(q, k, v, bias, gate_logit, mask, inward, drop=None):
    """One direction of the triplet attention.  q,k,v: (B,N,N,D,H).
    bias/gate_logit: (B,N,N,H) or None.  mask: (B,N,N,1).

    inward  (triplet.py:216-227):  S[b,i,j,k,h] = Q[b,i,j]·K[b,j,k] + E[b,i,k] + M[b,i,k]
                                   out[b,i,j]   = sum_k softmax_k(S) σ(G[b,i,k]+M[b,i,k]) V[b,j,k]
    outward (triplet.py:235-246):  S[b,i,j,k,h] = Q[b,i,j]·K[b,k,j] + E[b,k,i] + M[b,k,i]
                                   out[b,i,j]   = sum_k softmax_k(S) σ(G[b,k,i]+M[b,k,i]) V[b,k,j]
    (q already carries the D^-1/2 scale.)
    """
    if not inward:
        # rename so that the contraction partner is always indexed [b,j,k]
        k = k.transpose(1, 2)
        v = v.transpose(1, 2)
        mask = mask.transpose(1, 2)
        if bias is not None:
            bias = bias.transpose(1, 2)
        if gate_logit is not None:
            gate_logit = gate_logit.transpose(1, 2)
    # scores over the third node k: (B,i,j,k,H)
    s = torch.einsum('bijdh,bjkdh->bijkh', q, k)
    if bias is not None:
        s = s + bias.unsqueeze(2)
    p = torch.softmax(s + mask.unsqueeze(2), dim=3)
    if gate_logit is not None:
        p = p * torch.sigmoid(gate_logit + mask).unsqueeze(2)
    if drop is not None:            # F.dropout(A, p) with a GIVEN keep pattern (triplet.py:223-225, :242-244)
        keep, scale = drop          # keep: (B,i,j,k,H) bool
        p = p * keep.to(p.dtype) * scale
    return torch.einsum('bijkh,bjkdh->bijdh', p, v)


def triplet_attention_core(qkv_in, eg_in, qkv_out, eg_out, mask, num_heads,
                           gated=True, biased=True, dropout=None):
    """lib/tgt/layers/triplet.py:209-248 (and :276-320 ungated, :343-385 axial).

    qkv_* : (B,N,N,3C)   eg_*: (B,N,N,2H) gated | (B,N,N,H) ungated | None axial
    returns Va (B,N,N,2C) with channel = d*2H + dir*H + h  (triplet.py:248).
    dropout: None, or (keep_in, keep_out, scale) with keep_* (B,i,j,k,H) bool: the attention dropout
    with a given keep pattern (the tests pass the kernels' counter-based pattern).
    """
    B, N, _, C3 = qkv_in.shape
    C = C3 // 3
    D = C // num_heads
    outs = []
    for qkv, eg, inward in ((qkv_in, eg_in, True), (qkv_out, eg_out, False)):
        drop = None if dropout is None else (dropout[0 if inward else 1], dropout[2])
        q, k, v = (heads_minor(t, num_heads) for t in qkv.split(C, dim=-1))
        q = q * (D ** -0.5)
        bias = gate = None
        if biased and gated:
            bias, gate = eg.split(num_heads, dim=-1)
        elif biased:
            bias = eg
        outs.append(_triplet_dir(q, k, v, bias, gate, mask, inward, drop))
    return torch.cat(outs, dim=-1).reshape(B, N, N, 2 * C)


def triplet_aggregate_core(v_both, eg, mask, num_heads, gated=True, dropout=None):
    """lib/tgt/layers/triplet.py:50-70 (gated; outward unmasked, quirk Q2) and
    :100-123 (ungated; both directions masked).

    v_both: (B,N,N,2C) = [V_in | V_out]; eg: (B,N,N,4H) gated = [E_in|G_in|E_out|G_out]
    or (B,N,N,2H) ungated = [E_in|E_out].
    """
    B, N, _, C2 = v_both.shape
    C = C2 // 2
    v_in, v_out = (heads_minor(t, num_heads) for t in v_both.split(C, dim=-1))
    if gated:
        e_in, g_in, e_out, g_out = eg.split(num_heads, dim=-1)
        a_in = torch.softmax(e_in + mask, dim=2) * torch.sigmoid(g_in + mask)
        a_out = torch.softmax(e_out, dim=1) * torch.sigmoid(g_out)
    else:
        e_in, e_out = eg.split(num_heads, dim=-1)
        a_in = torch.softmax(e_in + mask, dim=2)
        a_out = torch.softmax(e_out + mask, dim=1)
    if dropout is not None:         # (keep_in (B,i,k,H), keep_out (B,k,i,H), scale): given keep pattern (:59-60, :66-67)
        a_in = a_in * dropout[0].to(a_in.dtype) * dropout[2]
        a_out = a_out * dropout[1].to(a_out.dtype) * dropout[2]
    # o_in[b,i,j,d,h]  = sum_k a_in[b,i,k,h]  V_in[b,j,k,d,h]     (:61)
    # o_out[b,i,j,d,h] = sum_k a_out[b,k,i,h] V_out[b,k,j,d,h]    (:68)
    o_in = torch.einsum('bikh,bjkdh->bijdh', a_in, v_in)
    o_out = torch.einsum('bkih,bkjdh->bijdh', a_out, v_out)
    return torch.cat([o_in, o_out], dim=-1).reshape(B, N, N, 2 * C)


def triangular_update_core(v4, e4, mask, num_heads):
    """lib/tgt/layers/triplet.py:156-172.  v4/e4: (B,N,N,4H) chunks
    [in_gate, in_lin, out_gate, out_lin]; returns (B,N,N,2H)."""
    vig, vil, vog, vol = v4.split(num_heads, dim=-1)
    eig, eil, eog, eol = e4.split(num_heads, dim=-1)
    sl = lambda g, l: torch.sigmoid(g + mask) * l
    v_in, v_out, e_in, e_out = sl(vig, vil), sl(vog, vol), sl(eig, eil), sl(eog, eol)
    o_in = torch.einsum('bikh,bjkh->bijh', e_in, v_in)
    o_out = torch.einsum('bkih,bkjh->bijh', e_out, v_out)
    return torch.cat([o_in, o_out], dim=-1)


# --------------------------------------------------------------------------
# small pieces either side of the path
# --------------------------------------------------------------------------
def pairwise_dist(coords):
    """lib/training_schemes/pcqm/commons.py:6-8"""
    return torch.norm(coords.unsqueeze(-2) - coords.unsqueeze(-3), dim=-1)


def smoothed_coord_noise(coords, edge_mask, level, smoothing, generator=None):
    """lib/training_schemes/pcqm/commons.py:10-16"""
    noise = torch.randn(coords.shape, dtype=coords.dtype, device=coords.device,
                        generator=generator) * level
    dm = pairwise_dist(coords) + (1 - edge_mask.to(coords.dtype)) * 1e9
    w = torch.softmax(-dm / smoothing, dim=-1)
    return coords + w @ noise


def binned_distance_xent(logits, dist_target, edge_mask, num_bins, range_bins,
                         reduce=True):
    """lib/training_schemes/pcqm/commons.py:19-48"""
    bsz = logits.size(0)
    bins = (dist_target * ((num_bins - 1) / range_bins)).long().clamp(0, num_bins - 1)
    xent = F.cross_entropy(logits.reshape(-1, num_bins), bins.reshape(-1),
                           reduction='none').view(bsz, -1)
    m = edge_mask.to(xent.dtype).view(bsz, -1)
    if reduce:
        return (xent * m).sum() / (m.sum() + 1e-9)
    return (xent * m).sum(dim=1) / (m.sum(dim=1) + 1e-9)


def gaussian_kernel(x, mean, std):
    """lib/models/pcqm/layers.py:129-134 (note pi = 3.14159 there)."""
    a = (2 * 3.14159) ** 0.5
    return torch.exp(-0.5 * (((x - mean) / std) ** 2)) / (a * std)


def bins_to_dist(bins, bin_size, shift_half=True, zero_diag=True):
    """lib/training_schemes/pcqm/commons.py:72-82"""
    b = bins.float()
    if shift_half:
        b = b + 0.5
    d = b * bin_size
    d = d + d.transpose(-2, -1)
    if zero_diag:
        d = d * (1 - torch.eye(d.size(-1), dtype=d.dtype, device=d.device))
    return d
