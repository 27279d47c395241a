"""Triplet edge-update modules on the HIP kernels.

Module API, parameter names and shapes = reference lib/tgt/layers/triplet.py
(so `model_state.pt` files load unchanged); the einsum/softmax/gate chains of
the reference's forward() are one call into libtgt_hip.so each.

How the projections feed the kernels: the reference splits channels
HEAD-MINOR (c = d*H + h); the matrix-core kernels want each head's D values
contiguous.  Instead of permuting (B,N,N,C) activations, the rows of the
projection WEIGHTS are gathered into head-major order every step (a few
hundred KB), all projections of the module are fused into ONE GEMM whose
output row is [Q_in|K_in|V_in|Q_out|K_out|V_out|E_in|G_in|E_out|G_out], and the
columns of lin_O are gathered to match the kernel's output order.  Autograd
sends the gradients back through the gathers to the canonical parameters.
"""
import torch
from torch import nn
import torch.nn.functional as F

from ... import layout, ops


class Linear(nn.Linear):
    """nn.Linear parameters; GEMMs through ops.linear (see layers.Linear)."""

    def forward(self, x):
        return ops.linear(x, self.weight, self.bias)


class LayerNorm(nn.LayerNorm):
    """nn.LayerNorm parameters, HIP kernels (see layers.LayerNorm)."""

    def forward(self, x):
        return ops.layer_norm(x, self.weight, self.bias, self.eps)


class _TripletBase(nn.Module):
    def __init__(self, edge_width, num_heads, attention_dropout=0):
        super().__init__()
        self.edge_width = edge_width
        self.num_heads = num_heads
        self.attention_dropout = attention_dropout
        self._idx = {}

    def _index(self, name, make, device):
        """(permutation, inverse permutation) on `device`, cached"""
        key = (name, device)
        if key not in self._idx:
            perm = make()
            inv = torch.empty_like(perm)
            inv[perm] = torch.arange(perm.numel())
            self._idx[key] = (perm.to(device), inv.to(device))
        return self._idx[key]

    def _va_perm(self, device):
        key = ('va32', device)
        if key not in self._idx:
            perm, inv = self._index('va', lambda: layout.va_cols_head_major(self.edge_width, self.num_heads), device)
            self._idx[key] = (perm.int(), inv.int())
        return self._idx[key]

    def _out_proj(self, va):
        """lin_O on the kernel's [dir][h][d] channel order (reference order is
        d*2H + dir*H + h, triplet.py:248): the weight's input columns are re-ordered, one launch"""
        return ops.linear_permuted_cols(va, self.lin_O.weight, self.lin_O.bias, *self._va_perm(va.device))

    def out_proj_residual_ln(self, va, res, scale, ln):
        """(s, LayerNorm(s)), s = res + scale * lin_O(va): the module's output projection, the DropPath + residual add that follows
        it in TGT_Layer (reference layers.py:284-287) and the LayerNorm that opens the edge FFN, ONE launch where the shape
        qualifies (ops.linear_residual_layer_norm, K = 512)"""
        return ops.linear_residual_layer_norm(va, self.lin_O.weight, self.lin_O.bias, res, scale, ln.weight, ln.bias, ln.eps,
                                              col_perm=self._va_perm(va.device))

    def _param_table(self, blocks, width):
        """ops.ParamTable of a fused projection: blocks = [(source id, row index tensor)], in
        fused-row order; rows past the blocks (alignment padding) are zero rows"""
        src = torch.cat([torch.full((len(idx),), sid, dtype=torch.int32) for sid, idx in blocks])
        row = torch.cat([idx.to(torch.int32) for _, idx in blocks])
        pad = width - src.numel()
        if pad:
            src = torch.cat([src, torch.full((pad,), -1, dtype=torch.int32)])
            row = torch.cat([row, torch.zeros(pad, dtype=torch.int32)])
        return ops.ParamTable(src, row, self.edge_width, 1 + max(sid for sid, _ in blocks))


class TripletAttention(_TripletBase):
    """Reference lib/tgt/layers/triplet.py:179-250."""
    gated, biased = True, True

    def __init__(self, edge_width, num_heads, attention_dropout=0):
        super().__init__(edge_width, num_heads, attention_dropout)
        assert not (edge_width % num_heads), 'edge_width must be divisible by num_heads'
        self._dot_dim = edge_width // num_heads
        self._scale_factor = self._dot_dim ** -0.5
        nb = num_heads * (2 if self.gated else 1)
        bias_name = 'lin_EG' if self.gated else 'lin_E'
        self.tri_ln_e = LayerNorm(edge_width)
        self.lin_QKV_in = Linear(edge_width, edge_width * 3)
        if self.biased:
            setattr(self, bias_name + '_in', Linear(edge_width, nb))
        self.lin_QKV_out = Linear(edge_width, edge_width * 3)
        if self.biased:
            setattr(self, bias_name + '_out', Linear(edge_width, nb))
        self.lin_O = Linear(edge_width * 2, edge_width)
        self._bias_name = bias_name
        self._layout = ops.TripletLayout(edge_width, num_heads, gated=self.gated, biased=self.biased)
        # fused projection rows: [QKV_in (head-major) | QKV_out (head-major) | E(G)_in | E(G)_out | pad]
        rows = layout.qkv_rows_head_major(edge_width, num_heads)
        blocks = [(0, rows), (1, rows)]
        if self.biased:
            blocks += [(2, torch.arange(nb)), (3, torch.arange(nb))]
        self._table = self._param_table(blocks, self._layout.width)

    def _projection_params(self):
        lins = [self.lin_QKV_in, self.lin_QKV_out]
        if self.biased:
            lins += [getattr(self, self._bias_name + '_in'), getattr(self, self._bias_name + '_out')]
        return tuple(t for lin in lins for t in (lin.weight, lin.bias))

    def forward(self, e, mask):
        return self.forward_normed(self.tri_ln_e(e), mask)

    takes_graph_scale = True          # forward_normed(..., graph_scale=): DropPath-dropped graphs are not computed

    def forward_normed(self, x, mask, graph_scale=None):
        """the block after tri_ln_e (TGT_Layer fuses that LayerNorm with the residual add before it).
        graph_scale (B,) float32: the DropPath factor the CALLER multiplies this block's result with at the residual add
        (reference layers.py:286-287); graphs whose factor is 0 are skipped by the attention kernels"""
        return self._out_proj(self.attend(x, mask, graph_scale))

    def attend(self, x, mask, graph_scale=None):
        """forward_normed without lin_O: Va in the kernels' channel order (TGT_Layer fuses lin_O with what follows it:
        out_proj_residual_ln)"""
        B, N = x.shape[0], x.shape[1]
        return ops.projected_triplet_attention(x, self._projection_params(), None, ops.as_mask3(mask, B, N),
                                               self._layout, table=self._table,
                                               dropout=ops.draw_dropout(self.attention_dropout, self.training),
                                               graph_scale=graph_scale)


class TripletAttentionUngated(TripletAttention):
    """Reference lib/tgt/layers/triplet.py:253-322 (params lin_E_in / lin_E_out)."""
    gated, biased = False, True


class AxialAttention(TripletAttention):
    """Reference lib/tgt/layers/triplet.py:325-387 (no third-arm bias or gate)."""
    gated, biased = False, False


class TripletAggregate(_TripletBase):
    """Reference lib/tgt/layers/triplet.py:22-73."""
    gated = True

    def __init__(self, edge_width, num_heads, attention_dropout=0):
        super().__init__(edge_width, num_heads, attention_dropout)
        assert not (edge_width % num_heads), 'edge_width must be divisible by num_heads'
        self._dot_dim = edge_width // num_heads
        self._scale_factor = self._dot_dim ** -0.5
        self.tri_ln_e = LayerNorm(edge_width)
        self.lin_V = Linear(edge_width, edge_width * 2)
        if self.gated:
            self.lin_EG = Linear(edge_width, num_heads * 4)
        else:
            self.lin_E = Linear(edge_width, num_heads * 2)
        self.lin_O = Linear(edge_width * 2, edge_width)
        self._layout = ops.AggregateLayout(edge_width, num_heads, gated=self.gated)
        # fused projection rows: [V_in | V_out (head-major) | E(G) | pad]
        self._table = self._param_table([(0, layout.qkv_rows_head_major(edge_width, num_heads, parts=2)),
                                         (1, torch.arange(num_heads * (4 if self.gated else 2)))], self._layout.width)

    def forward(self, e, mask):
        return self.forward_normed(self.tri_ln_e(e), mask)

    def forward_normed(self, x, mask):
        return self._out_proj(self.attend(x, mask))

    def attend(self, x, mask, graph_scale=None):
        """forward_normed without lin_O (see TripletAttention.attend; the aggregate kernels compute every graph)"""
        B, N = x.shape[0], x.shape[1]
        lin_b = self.lin_EG if self.gated else self.lin_E
        fused = ops.fused_linear(x, self._table, (self.lin_V.weight, self.lin_V.bias, lin_b.weight, lin_b.bias))
        return ops.triplet_aggregate(fused, ops.as_mask3(mask, B, N), self._layout,
                                     ops.draw_dropout(self.attention_dropout, self.training))


class TripletAggregateUngated(TripletAggregate):
    """Reference lib/tgt/layers/triplet.py:77-127."""
    gated = False


class TriangularUpdate(_TripletBase):
    """Reference lib/tgt/layers/triplet.py:134-176 (scalar values per head, sigmoid-gated
    linear units instead of softmax)."""

    def __init__(self, edge_width, num_heads, attention_dropout=0):
        super().__init__(edge_width, num_heads, attention_dropout)
        self.tri_ln_e = LayerNorm(edge_width)
        self.lin_V = Linear(edge_width, num_heads * 4)
        self.lin_E = Linear(edge_width, num_heads * 4)
        self.lin_O = Linear(num_heads * 2, edge_width * 2)

    def forward(self, e, mask):
        return self.forward_normed(self.tri_ln_e(e), mask)

    def forward_normed(self, x, mask):
        B, N = x.shape[0], x.shape[1]
        va = ops.triangular_update(self.lin_E(x), self.lin_V(x), ops.as_mask3(mask, B, N), self.num_heads)
        gate, lin = self.lin_O(va).chunk(2, dim=-1)
        return torch.sigmoid(gate) * lin


_LAYERS = {
    'aggregate': TripletAggregate,
    'aggregate_ungated': TripletAggregateUngated,
    'attention': TripletAttention,
    'attention_ungated': TripletAttentionUngated,
    'tiangular_update': TriangularUpdate,        # spelling of the reference factory (triplet.py:15)
    'axial_attention': AxialAttention,
}


def get_triplet_layer(layer_type):
    """Reference lib/tgt/layers/triplet.py:6-20."""
    if layer_type not in _LAYERS:
        raise ValueError(f'Invalid layer_type: {layer_type}')
    return _LAYERS[layer_type]
