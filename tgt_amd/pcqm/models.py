"""Input embedding and prediction heads of the PCQM task models.

state_dict schema = SURVEY App. B (reference lib/models/pcqm/*.py).  These run
as ordinary device ops once per step (<0.5 % of the forward); the encoder
between them is the HIP path.
"""
import math

import torch
from torch import nn
import torch.nn.functional as F

from .. import ops
from ..tgt import TGT_Encoder, Graph
from ..tgt.layers.blocks import LayerNorm, Linear

from ..knobs import K as _K
_EMBED_GEMM = _K.embed_gemm      # A/B knob (round 5): per-node mul / bias lookups without nn.Embedding's backward
NODE_FEATURES_OFFSET = 128      # reference lib/models/pcqm/consts.py:1-7
NUM_NODE_FEATURES = 9
EDGE_FEATURES_OFFSET = 8
NUM_EDGE_FEATURES = 3
HL_MEAN = 5.6894608
HL_STD = 1.1621397


class GaussianLayer(nn.Module):
    """Gaussian basis of (type-pair scaled) distances; reference layers.py:129-157."""

    def __init__(self, K=128, edge_types=512 * 3):
        super().__init__()
        self.K = K
        self.means = nn.Embedding(1, K)
        self.stds = nn.Embedding(1, K)
        self.mul = nn.Embedding(edge_types, 1, padding_idx=0)
        self.bias = nn.Embedding(edge_types, 1, padding_idx=0)
        nn.init.uniform_(self.means.weight, 0, 3)
        nn.init.uniform_(self.stds.weight, 0, 3)
        nn.init.constant_(self.bias.weight, 0)
        nn.init.constant_(self.mul.weight, 1)

    def pair_affine(self, type_i, type_j):
        """mul/bias summed over the (i-type, j-type) pair, built from PER-NODE gathers and a
        broadcast add: same value as embedding the (B,N,N,2) pair tensor, N times fewer
        indices through the embedding backward."""
        if type_i.is_cuda and _EMBED_GEMM:
            # Both tables as ONE gather per node role whose backward is a count-matrix GEMM (ops.gather_embed): nn.Embedding's
            # backward (embedding_dense_backward) sorts the indices and reads the number of distinct ones back on the HOST -- four
            # hipStreamSynchronize per step at the very end of the backward (rocprim radix sort + ~200 us of idle GPU each,
            # profiles/r05y_trace_edges.txt), which also threw away the lead the host had built up over the step.
            w2 = torch.cat([self.mul.weight, self.bias.weight], dim=1)             # (types, 2): [mul | bias]
            both = ops.gather_embed(type_i, w2, padding_idx=0).unsqueeze(2) + \
                ops.gather_embed(type_j, w2, padding_idx=0).unsqueeze(1)          # (B,N,N,2)
            return both[..., :1], both[..., 1:]
        mul = self.mul(type_i).unsqueeze(2) + self.mul(type_j).unsqueeze(1)        # (B,N,N,1)
        bias = self.bias(type_i).unsqueeze(2) + self.bias(type_j).unsqueeze(1)
        return mul, bias

    def forward(self, x, edge_types=None, affine=None):
        if affine is None:
            mul = self.mul(edge_types).sum(dim=-2)
            bias = self.bias(edge_types).sum(dim=-2)
        else:
            mul, bias = affine
        K = self.means.weight.numel()
        if x.is_cuda and mul.shape[:-1] == x.shape and K % 2 == 0 and K <= 512 and self.means.weight.dtype == torch.float32:
            # one HIP pass each way (csrc/gaussian.hip) instead of ~8 elementwise passes over the (B,N,N,K) tensor
            return ops.gaussian_basis(x, mul, bias, self.means.weight, self.stds.weight)
        x = (mul * x.unsqueeze(-1) + bias).float()
        mean = self.means.weight.float().view(-1)
        std = self.stds.weight.float().view(-1).abs() + 1e-2
        norm = (2 * 3.14159) ** 0.5                        # sic (layers.py:132)
        y = torch.exp(-0.5 * ((x - mean) / std) ** 2) / (norm * std)
        return y.type_as(self.means.weight)


class NonLinear(nn.Module):
    def __init__(self, input, output_size, hidden=None):
        super().__init__()
        hidden = input if hidden is None else hidden
        # (blocks.Linear: the weight gradients contract over B*N*N rows -- as plain library GEMMs they were 0.29 + 0.27 ms of a
        #  TGT-At step on a handful of workgroups, split over row chunks 0.03 ms each)
        self.layer1 = Linear(input, hidden)
        self.layer2 = Linear(hidden, output_size)

    def forward(self, x):
        return self.layer2(F.gelu(self.layer1(x)))


class Gaussian3DEmbed(nn.Module):
    def __init__(self, num_heads, num_edges, num_kernel):
        super().__init__()
        self.num_heads, self.num_edges, self.num_kernel = num_heads, num_edges, num_kernel
        self.gbf = GaussianLayer(num_kernel, num_edges)
        self.gbf_proj = NonLinear(num_kernel, num_heads)

    def forward(self, dist, node_type_edge=None, node_types=None):
        if node_types is not None:          # (type_i, type_j) per node: the fast path
            return self.gbf_proj(self.gbf(dist, affine=self.gbf.pair_affine(*node_types)))
        return self.gbf_proj(self.gbf(dist, node_type_edge.long()))


class Fourier3DEmbed(nn.Module):
    def __init__(self, num_heads, num_kernel, min_dist=0.01, max_dist=20):
        assert num_kernel % 2 == 0
        super().__init__()
        self.num_heads, self.num_kernel = num_heads, num_kernel
        self.min_dist, self.max_dist = min_dist, max_dist
        wave_lengths = torch.exp(torch.linspace(math.log(2 * min_dist), math.log(2 * max_dist), num_kernel // 2))
        self.register_buffer('angular_freqs', 2 * math.pi / wave_lengths)
        self.proj = nn.Linear(num_kernel, num_heads)

    def forward(self, dist):
        phase = dist.unsqueeze(-1) * self.angular_freqs
        return self.proj(torch.cat([torch.sin(phase), torch.cos(phase)], dim=-1))


class EmbedInput(nn.Module):
    """int features / hop distances / 3-D distances -> (h, e, mask).
    Reference lib/models/pcqm/layers.py:11-83."""

    def __init__(self, node_width, edge_width, upto_hop=32, embed_3d_type='gaussian', num_3d_kernels=128):
        super().__init__()
        self.node_width, self.edge_width = node_width, edge_width
        self.upto_hop, self.num_3d_kernels, self.embed_3d_type = upto_hop, num_3d_kernels, embed_3d_type
        self.nodef_embed = nn.Embedding(NUM_NODE_FEATURES * NODE_FEATURES_OFFSET + 1, node_width, padding_idx=0)
        self.dist_embed = nn.Embedding(upto_hop + 2, edge_width)
        self.featm_embed = nn.Embedding(NUM_EDGE_FEATURES * EDGE_FEATURES_OFFSET + 1, edge_width, padding_idx=0)
        if embed_3d_type == 'gaussian':
            self.m3d_embed = Gaussian3DEmbed(edge_width, 2 * NODE_FEATURES_OFFSET + 1, num_3d_kernels)
        elif embed_3d_type == 'fourier':
            self.m3d_embed = Fourier3DEmbed(edge_width, num_3d_kernels)
        elif embed_3d_type != 'none':
            raise ValueError('Invalid 3D embedding type')
        self._uses_3d = embed_3d_type != 'none'

    def forward(self, inputs):
        g = Graph(inputs)
        nodef = g.node_features.long()
        hops = g.distance_matrix.long().clamp(max=self.upto_hop + 1)
        # sums of embedding rows as count-matrix GEMMs (forward AND backward)
        h = ops.multi_hot_embed(nodef, self.nodef_embed.weight, padding_idx=0)
        n_hop = self.dist_embed.weight.shape[0]
        pair_idx = torch.cat([hops.unsqueeze(-1), g.feature_matrix.long() + n_hop], dim=-1)
        pair_w = torch.cat([self.dist_embed.weight, self.featm_embed.weight], dim=0)
        e = ops.multi_hot_embed(pair_idx, pair_w, padding_idx=n_hop)
        if self.embed_3d_type == 'gaussian':
            atom = nodef[:, :, 0]
            e = e + self.m3d_embed(g.dist_input, node_types=(atom, atom + NODE_FEATURES_OFFSET))
        elif self.embed_3d_type == 'fourier':
            e = e + self.m3d_embed(g.dist_input)
        edge_mask = g.edge_mask.unsqueeze(-1).to(e.dtype)
        g.h, g.e = h, e
        g.mask = (1 - edge_mask) * torch.finfo(e.dtype).min
        return g


class _Task(nn.Module):
    _node_ended, _edge_ended = True, True

    def _build(self, model_height, layer_multiplier, upto_hop, embed_3d_type, num_3d_kernels,
               num_dist_bins, layer_configs):
        self.model_height, self.layer_multiplier = model_height, layer_multiplier
        self.upto_hop, self.embed_3d_type, self.num_3d_kernels = upto_hop, embed_3d_type, num_3d_kernels
        self.node_width, self.edge_width = layer_configs['node_width'], layer_configs['edge_width']
        self.layer_configs = layer_configs
        self.encoder = TGT_Encoder(model_height=model_height, layer_multiplier=layer_multiplier,
                                   node_ended=self._node_ended, edge_ended=self._edge_ended,
                                   egt_simple=False, **layer_configs)
        self.input_embed = EmbedInput(node_width=self.node_width, edge_width=self.edge_width,
                                      upto_hop=upto_hop, embed_3d_type=embed_3d_type,
                                      num_3d_kernels=num_3d_kernels)
        if self._node_ended:
            self.final_ln_node = LayerNorm(self.node_width)
            self.pred = nn.Linear(self.node_width, 1)
            nn.init.constant_(self.pred.bias, HL_MEAN)
        if self._edge_ended:
            self.num_dist_bins = num_dist_bins
            self.final_ln_edge = LayerNorm(self.edge_width)
            self.dist_pred = Linear(self.edge_width, num_dist_bins)

    def _gap_head(self, g):
        h = self.final_ln_node(g.h)
        nodem = g.node_mask.float().unsqueeze(-1)
        h = (h * nodem).sum(dim=1) / (nodem.sum(dim=1) + 1e-9)
        return self.pred(h).squeeze(-1)

    def _dist_head(self, g):
        return self.dist_pred(self.final_ln_edge(g.e))


class TGT_Multi(_Task):
    """gap + binned-distance heads.  Reference lib/models/pcqm/multitask.py:10-68."""

    def __init__(self, model_height, layer_multiplier=1, upto_hop=32, embed_3d_type='gaussian',
                 num_3d_kernels=128, num_dist_bins=128, **layer_configs):
        super().__init__()
        self._build(model_height, layer_multiplier, upto_hop, embed_3d_type, num_3d_kernels,
                    num_dist_bins, layer_configs)

    def forward(self, inputs):
        g = self.encoder(self.input_embed(inputs))
        return self._gap_head(g), self._dist_head(g)


class TGT_Distance(_Task):
    """Reference lib/models/pcqm/distance_predictor.py:9-55 (last layer: EdgeUpdate only)."""
    _node_ended = False

    def __init__(self, model_height, layer_multiplier=1, upto_hop=32, embed_3d_type='gaussian',
                 num_3d_kernels=128, num_dist_bins=128, **layer_configs):
        super().__init__()
        self._build(model_height, layer_multiplier, upto_hop, embed_3d_type, num_3d_kernels,
                    num_dist_bins, layer_configs)

    def forward(self, inputs):
        return self._dist_head(self.encoder(self.input_embed(inputs)))


class TGT_Gap(_Task):
    """Reference lib/models/pcqm/gap_predictor.py:10-59 (last layer: no edge update)."""
    _edge_ended = False

    def __init__(self, model_height, layer_multiplier=1, upto_hop=32, embed_3d_type='gaussian',
                 num_3d_kernels=128, **layer_configs):
        super().__init__()
        self._build(model_height, layer_multiplier, upto_hop, embed_3d_type, num_3d_kernels,
                    None, layer_configs)

    def forward(self, inputs):
        return self._gap_head(self.encoder(self.input_embed(inputs)))
