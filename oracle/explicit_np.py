"""ORACLE, second form (test infrastructure only -- see oracle/core.py header).

Index-by-index numpy (float64) restatement of the three hot-path contractions
exactly as SURVEY.md Appendix A writes them, with python loops over the graph
nodes.  Deliberately shares no code with `oracle.core` (which is einsum-based)
so the two can check each other and the golden vectors; small shapes only.
"""
import numpy as np


def _sigmoid(x):
    with np.errstate(over='ignore'):
        return 1.0 / (1.0 + np.exp(-x))


def _softmax(x, axis):
    m = x.max(axis=axis, keepdims=True)
    ex = np.exp(x - m)
    return ex / ex.sum(axis=axis, keepdims=True)


def egt_attention(qkv, eg, mask, H, scale_degree=True):
    """lib/tgt/layers/layers.py:52-82.  qkv (B,N,3W), eg (B,N,N,2H), mask (B,N,N)."""
    B, N, W3 = qkv.shape
    W = W3 // 3
    D = W // H
    s = D ** -0.5
    v_att = np.zeros((B, N, W))
    h_hat = np.zeros((B, N, N, H))
    for b in range(B):
        for h in range(H):
            ch = np.arange(D) * H + h                         # c = d*H + h
            Q, K, V = qkv[b][:, ch], qkv[b][:, W + ch], qkv[b][:, 2 * W + ch]
            for l in range(N):
                logits = np.array([s * Q[l] @ K[m] + eg[b, l, m, h] for m in range(N)])
                h_hat[b, l, :, h] = logits
                gate = _sigmoid(eg[b, l, :, H + h] + mask[b, l])
                att = _softmax(logits + mask[b, l], 0) * gate
                out = sum(att[m] * V[m] for m in range(N))
                if scale_degree:
                    out = out * np.log(1 + gate.sum())
                v_att[b, l, ch] = out
    return v_att, h_hat


def triplet_attention(qkv_in, eg_in, qkv_out, eg_out, mask, H):
    """lib/tgt/layers/triplet.py:209-248 (gated).  qkv_* (B,N,N,3C), eg_* (B,N,N,2H),
    mask (B,N,N).  Returns Va (B,N,N,2C), channel = d*2H + dir*H + h."""
    B, N, _, C3 = qkv_in.shape
    C = C3 // 3
    D = C // H
    s = D ** -0.5
    va = np.zeros((B, N, N, D, 2 * H))
    for b in range(B):
        for h in range(H):
            ch = np.arange(D) * H + h
            Qi, Ki, Vi = qkv_in[b][..., ch], qkv_in[b][..., C + ch], qkv_in[b][..., 2 * C + ch]
            Qo, Ko, Vo = qkv_out[b][..., ch], qkv_out[b][..., C + ch], qkv_out[b][..., 2 * C + ch]
            for i in range(N):
                for j in range(N):
                    # inward: pairs (i,j),(j,k); third arm (i,k)
                    sc = np.array([s * Qi[i, j] @ Ki[j, k] + eg_in[b, i, k, h] + mask[b, i, k]
                                   for k in range(N)])
                    a = _softmax(sc, 0) * _sigmoid(eg_in[b, i, :, H + h] + mask[b, i, :])
                    va[b, i, j, :, h] = sum(a[k] * Vi[j, k] for k in range(N))
                    # outward: pairs (i,j),(k,j); third arm (k,i)
                    sc = np.array([s * Qo[i, j] @ Ko[k, j] + eg_out[b, k, i, h] + mask[b, k, i]
                                   for k in range(N)])
                    a = _softmax(sc, 0) * _sigmoid(eg_out[b, :, i, H + h] + mask[b, :, i])
                    va[b, i, j, :, H + h] = sum(a[k] * Vo[k, j] for k in range(N))
    return va.reshape(B, N, N, 2 * C)


def triplet_aggregate(v_both, eg, mask, H):
    """lib/tgt/layers/triplet.py:50-70 (gated; outward direction unmasked)."""
    B, N, _, C2 = v_both.shape
    C = C2 // 2
    D = C // H
    va = np.zeros((B, N, N, D, 2 * H))
    for b in range(B):
        for h in range(H):
            ch = np.arange(D) * H + h
            Vi, Vo = v_both[b][..., ch], v_both[b][..., C + ch]
            e_in, g_in = eg[b, :, :, h], eg[b, :, :, H + h]
            e_out, g_out = eg[b, :, :, 2 * H + h], eg[b, :, :, 3 * H + h]
            a_in = _softmax(e_in + mask[b], 1) * _sigmoid(g_in + mask[b])      # over k of [i,k]
            a_out = _softmax(e_out, 0) * _sigmoid(g_out)                       # over k of [k,i]
            for i in range(N):
                for j in range(N):
                    va[b, i, j, :, h] = sum(a_in[i, k] * Vi[j, k] for k in range(N))
                    va[b, i, j, :, H + h] = sum(a_out[k, i] * Vo[k, j] for k in range(N))
    return va.reshape(B, N, N, 2 * C)
