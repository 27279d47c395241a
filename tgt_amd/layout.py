"""Channel-order bookkeeping between the reference's HEAD-MINOR split
(c = d*H + h; reference lib/tgt/layers/triplet.py:213-215, :248) and the
HEAD-MAJOR order (c = h*D + d) the triplet kernels read.

The permutation is applied to the (small) projection WEIGHTS each step, never
to activations: W_hm = W[idx] makes the GEMM emit head-major channels directly,
while the canonical parameters (and hence state_dict / checkpoints) stay in
the reference's layout.
"""
import torch


def head_major_index(C, H):
    """idx (C,) with  x_headmajor[..., j] = x_reference[..., idx[j]],  j = h*D + d."""
    D = C // H
    j = torch.arange(C)
    return (j % D) * H + (j // D)


def qkv_rows_head_major(C, H, parts=3):
    """Row index for a lin_QKV-style weight (parts*C, in): each C-block head-major."""
    base = head_major_index(C, H)
    return torch.cat([base + p * C for p in range(parts)])


def va_cols_head_major(C, H):
    """Column index for lin_O (C, 2C): kernel channel dir*C + h*D + d  <->
    reference channel d*2H + dir*H + h (triplet.py:248)."""
    D = C // H
    j = torch.arange(2 * C)
    dr, rem = j // C, j % C
    h, d = rem // D, rem % D
    return d * (2 * H) + dr * H + h
