"""ORACLE modules (test infrastructure only -- see oracle/core.py header).

nn.Module restatement of the reference's `lib/tgt` layers and the PCQM task
models with the SAME class names, constructor keywords and `state_dict` keys
(SURVEY App. B), so a reference-initialised state_dict loads with
strict=True.  All arithmetic lives in `oracle.core`; these classes only own
parameters and wiring.  Used as (a) the checker for the HIP path and (b) the
CPU baseline that `bench.py` times next to the GPU number.
"""
import math

import torch
from torch import nn
import torch.nn.functional as F

from . import core


# ---- activations ---------------------------------------------------------
def resolve_activation(name):
    """lib/tgt/layers/activations.py:4-25: GLU family doubles lin_W1 width."""
    def _glu(gate_fn):
        def f(x):
            g, lin = x.chunk(2, dim=-1)
            return lin * gate_fn(g)
        return f
    table = {
        'geglu': _glu(F.gelu),
        'glu': _glu(torch.sigmoid),
        'swiglu': _glu(lambda g: torch.sigmoid(g) * g),
    }
    if name in table:
        return table[name], 2
    return getattr(F, name), 1


class Graph(dict):
    """attr-dict; lib/tgt/encoder.py:7-21"""
    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError('No such attribute: ' + key)

    def __setattr__(self, key, value):
        self[key] = value

    def __dir__(self):
        return list(super().__dir__()) + list(self.keys())

    def copy(self):
        return type(self)(self)


def _head_dims(width, heads, what):
    if width % heads:
        raise AssertionError(f'{what} must be divisible by num_heads')
    return width // heads


# ---- node attention ------------------------------------------------------
class EGT_Attention(nn.Module):
    """lib/tgt/layers/layers.py:15-84"""
    def __init__(self, node_width, edge_width, num_heads, source_dropout=0,
                 scale_degree=True, edge_update=True):
        super().__init__()
        self.node_width, self.edge_width, self.num_heads = node_width, edge_width, num_heads
        self.source_dropout, self.scale_degree, self.edge_update = source_dropout, scale_degree, edge_update
        self._dot_dim = _head_dims(node_width, num_heads, 'node_width')
        self.mha_ln_h = nn.LayerNorm(node_width)
        self.mha_ln_e = nn.LayerNorm(edge_width)
        self.lin_QKV = nn.Linear(node_width, node_width * 3)
        self.lin_EG = nn.Linear(edge_width, num_heads * 2)
        self.lin_O_h = nn.Linear(node_width, node_width)
        if edge_update:
            self.lin_O_e = nn.Linear(num_heads, edge_width)

    def forward(self, h, e, mask):
        qkv = self.lin_QKV(self.mha_ln_h(h))
        eg = self.lin_EG(self.mha_ln_e(e))
        if self.source_dropout > 0 and self.training:          # layers.py:55-59
            drop = torch.empty(h.size(0), 1, h.size(1), 1, dtype=h.dtype, device=h.device)
            drop = drop.bernoulli_(self.source_dropout) * torch.finfo(mask.dtype).min
            mask = mask + drop
        v_att, h_hat = core.egt_attention_core(qkv, eg, mask, self.num_heads, self.scale_degree)
        h = self.lin_O_h(v_att)
        if self.edge_update:
            e = self.lin_O_e(h_hat)
        return h, e


class EdgeUpdate(nn.Module):
    """lib/tgt/layers/layers.py:87-130"""
    def __init__(self, node_width, edge_width, num_heads):
        super().__init__()
        self.node_width, self.edge_width, self.num_heads = node_width, edge_width, num_heads
        self._dot_dim = _head_dims(node_width, num_heads, 'node_width')
        self.mha_ln_h = nn.LayerNorm(node_width)
        self.mha_ln_e = nn.LayerNorm(edge_width)
        self.lin_QK = nn.Linear(node_width, node_width * 2)
        self.lin_E = nn.Linear(edge_width, num_heads)
        self.lin_O_e = nn.Linear(num_heads, edge_width)

    def forward(self, h, e, mask):
        qk = self.lin_QK(self.mha_ln_h(h))
        bias = self.lin_E(self.mha_ln_e(e))
        return h, self.lin_O_e(core.edge_update_core(qk, bias, self.num_heads))


# ---- FFN / DropPath ------------------------------------------------------
class FFN(nn.Module):
    """lib/tgt/layers/layers.py:134-160"""
    def __init__(self, width, multiplier=1., act_dropout=0., activation='gelu'):
        super().__init__()
        self.width, self.multiplier, self.act_dropout, self.activation = width, multiplier, act_dropout, activation
        self.ffn_fn, self.act_mul = resolve_activation(activation)
        inner = round(width * multiplier)
        self.ffn_ln = nn.LayerNorm(width)
        self.lin_W1 = nn.Linear(width, inner * self.act_mul)
        self.lin_W2 = nn.Linear(inner, width)
        self.dropout = nn.Dropout(act_dropout)

    def forward(self, x):
        return self.lin_W2(self.dropout(self.ffn_fn(self.lin_W1(self.ffn_ln(x)))))


class DropPath(nn.Module):
    """per-sample stochastic depth; lib/tgt/layers/layers.py:163-177"""
    def __init__(self, drop_path=0.):
        super().__init__()
        self.drop_path = drop_path
        self._keep_prob = 1 - drop_path

    def forward(self, x):
        if self.drop_path > 0 and self.training:
            keep = torch.empty([x.size(0)] + [1] * (x.ndim - 1), dtype=x.dtype,
                               device=x.device).bernoulli_(self._keep_prob)
            x = x.div(self._keep_prob) * keep
        return x


# ---- triplet modules -----------------------------------------------------
class _TripletBase(nn.Module):
    def __init__(self, edge_width, num_heads, attention_dropout=0):
        super().__init__()
        self.edge_width, self.num_heads, self.attention_dropout = edge_width, num_heads, attention_dropout
        if attention_dropout:
            # p=0 in every shipped config (tgt_training.py:35); the oracle
            # restates the deterministic path only.
            raise NotImplementedError('oracle: triplet attention_dropout > 0 not restated')
        self.tri_ln_e = nn.LayerNorm(edge_width)


class TripletAttention(_TripletBase):
    """lib/tgt/layers/triplet.py:179-250"""
    gated, biased = True, True

    def __init__(self, edge_width, num_heads, attention_dropout=0):
        super().__init__(edge_width, num_heads, attention_dropout)
        self._dot_dim = _head_dims(edge_width, num_heads, 'edge_width')
        n_bias = num_heads * (2 if self.gated else 1)
        self.lin_QKV_in = nn.Linear(edge_width, edge_width * 3)
        if self.biased:
            setattr(self, 'lin_EG_in' if self.gated else 'lin_E_in', nn.Linear(edge_width, n_bias))
        self.lin_QKV_out = nn.Linear(edge_width, edge_width * 3)
        if self.biased:
            setattr(self, 'lin_EG_out' if self.gated else 'lin_E_out', nn.Linear(edge_width, n_bias))
        self.lin_O = nn.Linear(edge_width * 2, edge_width)

    def _bias(self, x, which):
        if not self.biased:
            return None
        return getattr(self, ('lin_EG_' if self.gated else 'lin_E_') + which)(x)

    def forward(self, e, mask):
        x = self.tri_ln_e(e)
        va = core.triplet_attention_core(self.lin_QKV_in(x), self._bias(x, 'in'),
                                         self.lin_QKV_out(x), self._bias(x, 'out'),
                                         mask, self.num_heads, self.gated, self.biased)
        return self.lin_O(va)


class TripletAttentionUngated(TripletAttention):
    """lib/tgt/layers/triplet.py:253-322"""
    gated, biased = False, True


class AxialAttention(TripletAttention):
    """lib/tgt/layers/triplet.py:325-387"""
    gated, biased = False, False


class TripletAggregate(_TripletBase):
    """lib/tgt/layers/triplet.py:22-73"""
    gated = True

    def __init__(self, edge_width, num_heads, attention_dropout=0):
        super().__init__(edge_width, num_heads, attention_dropout)
        self._dot_dim = _head_dims(edge_width, num_heads, 'edge_width')
        self.lin_V = nn.Linear(edge_width, edge_width * 2)
        if self.gated:
            self.lin_EG = nn.Linear(edge_width, num_heads * 4)
        else:
            self.lin_E = nn.Linear(edge_width, num_heads * 2)
        self.lin_O = nn.Linear(edge_width * 2, edge_width)

    def forward(self, e, mask):
        x = self.tri_ln_e(e)
        eg = self.lin_EG(x) if self.gated else self.lin_E(x)
        return self.lin_O(core.triplet_aggregate_core(self.lin_V(x), eg, mask, self.num_heads, self.gated))


class TripletAggregateUngated(TripletAggregate):
    """lib/tgt/layers/triplet.py:77-127"""
    gated = False


class TriangularUpdate(_TripletBase):
    """lib/tgt/layers/triplet.py:134-176"""
    def __init__(self, edge_width, num_heads, attention_dropout=0):
        super().__init__(edge_width, num_heads, attention_dropout)
        self.lin_V = nn.Linear(edge_width, num_heads * 4)
        self.lin_E = nn.Linear(edge_width, num_heads * 4)
        self.lin_O = nn.Linear(num_heads * 2, edge_width * 2)

    def forward(self, e, mask):
        x = self.tri_ln_e(e)
        va = core.triangular_update_core(self.lin_V(x), self.lin_E(x), mask, self.num_heads)
        g, lin = self.lin_O(va).chunk(2, dim=-1)
        return torch.sigmoid(g) * lin


_TRIPLET_TYPES = {
    'aggregate': TripletAggregate,
    'aggregate_ungated': TripletAggregateUngated,
    'attention': TripletAttention,
    'attention_ungated': TripletAttentionUngated,
    'tiangular_update': TriangularUpdate,      # sic -- lib/tgt/layers/triplet.py:15
    'axial_attention': AxialAttention,
}


def get_triplet_layer(layer_type):
    """lib/tgt/layers/triplet.py:6-20"""
    try:
        return _TRIPLET_TYPES[layer_type]
    except KeyError:
        raise ValueError(f'Invalid layer_type: {layer_type}')


# ---- layer / encoder -----------------------------------------------------
class TGT_Layer(nn.Module):
    """lib/tgt/layers/layers.py:180-302"""
    def __init__(self, node_width, edge_width, num_heads, activation='gelu',
                 scale_degree=True, node_update=True, edge_update=True,
                 triplet_heads=0, triplet_type='aggregate', triplet_dropout=0,
                 node_ffn_multiplier=1., edge_ffn_multiplier=1., source_dropout=0,
                 drop_path=0, node_act_dropout=0, edge_act_dropout=0):
        super().__init__()
        self.node_width, self.edge_width, self.num_heads = node_width, edge_width, num_heads
        self.node_update, self.edge_update = node_update, edge_update
        self.triplet_heads, self.triplet_type = triplet_heads, triplet_type
        self._triplet_update = triplet_heads > 0
        if node_update:
            self.update = EGT_Attention(node_width, edge_width, num_heads,
                                        source_dropout=source_dropout,
                                        scale_degree=scale_degree, edge_update=edge_update)
            self.node_ffn = FFN(node_width, node_ffn_multiplier, node_act_dropout, activation)
        elif edge_update:
            self.update = EdgeUpdate(node_width, edge_width, num_heads)
        else:
            raise ValueError('At least one of node_update and edge_update must be True')
        if edge_update:
            if self._triplet_update:
                self.tria = get_triplet_layer(triplet_type)(
                    edge_width=edge_width, num_heads=triplet_heads,
                    attention_dropout=triplet_dropout)
            self.edge_ffn = FFN(edge_width, edge_ffn_multiplier, edge_act_dropout, activation)
        self.drop_path = DropPath(drop_path)

    def forward(self, g):
        h, e, mask = g.h, g.e, g.mask
        dh, de = self.update(h, e, mask)
        if self.node_update:
            h = h + self.drop_path(dh)
            h = h + self.drop_path(self.node_ffn(h))
        if self.edge_update:
            e = e + self.drop_path(de)
            if self._triplet_update:
                e = e + self.drop_path(self.tria(e, mask))   # un-dropped mask (Q5)
            e = e + self.drop_path(self.edge_ffn(e))
        g = g.copy()
        g.h, g.e = h, e
        return g


class TGT_Encoder(nn.Module):
    """lib/tgt/encoder.py:24-90"""
    class IndivConfig(list):
        pass

    def __init__(self, model_height=4, layer_multiplier=1, node_ended=True,
                 edge_ended=True, egt_simple=False, **layer_configs):
        super().__init__()
        self.model_height, self.layer_multiplier = model_height, layer_multiplier
        self.node_ended, self.edge_ended, self.egt_simple = node_ended, edge_ended, egt_simple
        self.layer_configs = layer_configs
        assert node_ended or edge_ended, 'At least one of node_ended and edge_ended must be True'
        self.TGT_layers = nn.ModuleList(
            TGT_Layer(**self.get_layer_kwargs(i)) for i in range(model_height))

    def get_layer_kwargs(self, i):
        kw = {}
        for key, val in self.layer_configs.items():
            if isinstance(val, self.IndivConfig):
                kw[key] = val[i]
            elif key == 'drop_path':
                kw[key] = val * i / (self.model_height - 1)      # Q8: 1-layer model divides by 0
            else:
                kw[key] = val
        last = i == self.model_height - 1
        kw['node_update'] = not (last and not self.node_ended)
        kw['edge_update'] = (not self.egt_simple) and not (last and not self.edge_ended)
        return kw

    def forward(self, inputs):
        g = Graph(inputs)
        for layer in self.TGT_layers:
            for _ in range(self.layer_multiplier):       # weight-shared repeats, encoder.py:80-84
                g = layer(g)
        return g


# ---- PCQM input embedding and task heads ---------------------------------
NODE_FEATURES_OFFSET, NUM_NODE_FEATURES = 128, 9      # lib/models/pcqm/consts.py
EDGE_FEATURES_OFFSET, NUM_EDGE_FEATURES = 8, 3
HL_MEAN, HL_STD = 5.6894608, 1.1621397


class GaussianLayer(nn.Module):
    """lib/models/pcqm/layers.py:137-157"""
    def __init__(self, K=128, edge_types=512 * 3):
        super().__init__()
        self.K = K
        self.means, self.stds = nn.Embedding(1, K), nn.Embedding(1, K)
        self.mul = nn.Embedding(edge_types, 1, padding_idx=0)
        self.bias = nn.Embedding(edge_types, 1, padding_idx=0)
        nn.init.uniform_(self.means.weight, 0, 3)
        nn.init.uniform_(self.stds.weight, 0, 3)
        nn.init.constant_(self.bias.weight, 0)
        nn.init.constant_(self.mul.weight, 1)

    def forward(self, x, edge_types):
        x = self.mul(edge_types).sum(dim=-2) * x.unsqueeze(-1) + self.bias(edge_types).sum(dim=-2)
        x = x.expand(-1, -1, -1, self.K)
        mean = self.means.weight.float().view(-1)
        std = self.stds.weight.float().view(-1).abs() + 1e-2
        return core.gaussian_kernel(x.float(), mean, std).type_as(self.means.weight)


class NonLinear(nn.Module):
    """lib/models/pcqm/layers.py:160-173"""
    def __init__(self, input, output_size, hidden=None):
        super().__init__()
        hidden = input if hidden is None else hidden
        self.layer1, self.layer2 = nn.Linear(input, hidden), nn.Linear(hidden, output_size)

    def forward(self, x):
        return self.layer2(F.gelu(self.layer1(x)))


class Gaussian3DEmbed(nn.Module):
    """lib/models/pcqm/layers.py:112-126"""
    def __init__(self, num_heads, num_edges, num_kernel):
        super().__init__()
        self.gbf = GaussianLayer(num_kernel, num_edges)
        self.gbf_proj = NonLinear(num_kernel, num_heads)

    def forward(self, dist, node_type_edge):
        return self.gbf_proj(self.gbf(dist, node_type_edge.long()))


class Fourier3DEmbed(nn.Module):
    """lib/models/pcqm/layers.py:86-109"""
    def __init__(self, num_heads, num_kernel, min_dist=0.01, max_dist=20):
        assert num_kernel % 2 == 0
        super().__init__()
        wl = torch.exp(torch.linspace(math.log(2 * min_dist), math.log(2 * max_dist), num_kernel // 2))
        self.register_buffer('angular_freqs', 2 * math.pi / wl)
        self.proj = nn.Linear(num_kernel, num_heads)

    def forward(self, dist):
        ph = dist.unsqueeze(-1) * self.angular_freqs
        return self.proj(torch.cat([torch.sin(ph), torch.cos(ph)], dim=-1))


class EmbedInput(nn.Module):
    """lib/models/pcqm/layers.py:11-83"""
    def __init__(self, node_width, edge_width, upto_hop=32, embed_3d_type='gaussian', num_3d_kernels=128):
        super().__init__()
        self.upto_hop, self.embed_3d_type = upto_hop, embed_3d_type
        self.nodef_embed = nn.Embedding(NUM_NODE_FEATURES * NODE_FEATURES_OFFSET + 1, node_width, padding_idx=0)
        self.dist_embed = nn.Embedding(upto_hop + 2, edge_width)
        self.featm_embed = nn.Embedding(NUM_EDGE_FEATURES * EDGE_FEATURES_OFFSET + 1, edge_width, padding_idx=0)
        if embed_3d_type == 'gaussian':
            self.m3d_embed = Gaussian3DEmbed(edge_width, 2 * NODE_FEATURES_OFFSET + 1, num_3d_kernels)
        elif embed_3d_type == 'fourier':
            self.m3d_embed = Fourier3DEmbed(edge_width, num_3d_kernels)
        elif embed_3d_type != 'none':
            raise ValueError('Invalid 3D embedding type')

    def forward(self, inputs):
        g = Graph(inputs)
        nodef = g.node_features.long()
        h = self.nodef_embed(nodef).sum(dim=2)
        dm = g.distance_matrix.long().clamp(max=self.upto_hop + 1)
        e = self.dist_embed(dm) + self.featm_embed(g.feature_matrix.long()).sum(dim=-2)
        if self.embed_3d_type == 'gaussian':
            n = nodef.size(1)
            ti = nodef[:, :, 0]
            pair = torch.stack([ti.unsqueeze(2).expand(-1, -1, n),
                                (ti + NODE_FEATURES_OFFSET).unsqueeze(1).expand(-1, n, -1)], dim=-1)
            e = e + self.m3d_embed(g.dist_input, pair)
        elif self.embed_3d_type == 'fourier':
            e = e + self.m3d_embed(g.dist_input)
        em = g.edge_mask.unsqueeze(-1).to(e.dtype)
        g.h, g.e, g.mask = h, e, (1 - em) * torch.finfo(e.dtype).min      # layers.py:78-80
        return g


class _TaskModel(nn.Module):
    node_ended = edge_ended = True

    def __init__(self, model_height, layer_multiplier=1, upto_hop=32, embed_3d_type='gaussian',
                 num_3d_kernels=128, num_dist_bins=None, **layer_configs):
        super().__init__()
        self.node_width, self.edge_width = layer_configs['node_width'], layer_configs['edge_width']
        self.encoder = TGT_Encoder(model_height=model_height, layer_multiplier=layer_multiplier,
                                   node_ended=self.node_ended, edge_ended=self.edge_ended,
                                   egt_simple=False, **layer_configs)
        self.input_embed = EmbedInput(self.node_width, self.edge_width, upto_hop, embed_3d_type, num_3d_kernels)
        if self.node_ended:
            self.final_ln_node = nn.LayerNorm(self.node_width)
            self.pred = nn.Linear(self.node_width, 1)
            nn.init.constant_(self.pred.bias, HL_MEAN)
        if self.edge_ended:
            self.final_ln_edge = nn.LayerNorm(self.edge_width)
            self.dist_pred = nn.Linear(self.edge_width, num_dist_bins)

    def _gap(self, g):
        h = self.final_ln_node(g.h)
        nm = g.node_mask.float().unsqueeze(-1)
        return self.pred((h * nm).sum(dim=1) / (nm.sum(dim=1) + 1e-9)).squeeze(-1)

    def _bins(self, g):
        return self.dist_pred(self.final_ln_edge(g.e))


class TGT_Multi(_TaskModel):
    """lib/models/pcqm/multitask.py:10-68"""
    def __init__(self, model_height, layer_multiplier=1, upto_hop=32, embed_3d_type='gaussian',
                 num_3d_kernels=128, num_dist_bins=128, **layer_configs):
        super().__init__(model_height, layer_multiplier, upto_hop, embed_3d_type,
                         num_3d_kernels, num_dist_bins, **layer_configs)

    def forward(self, inputs):
        g = self.encoder(self.input_embed(inputs))
        return self._gap(g), self._bins(g)


class TGT_Distance(_TaskModel):
    """lib/models/pcqm/distance_predictor.py:9-55"""
    node_ended = False

    def __init__(self, model_height, layer_multiplier=1, upto_hop=32, embed_3d_type='gaussian',
                 num_3d_kernels=128, num_dist_bins=128, **layer_configs):
        super().__init__(model_height, layer_multiplier, upto_hop, embed_3d_type,
                         num_3d_kernels, num_dist_bins, **layer_configs)

    def forward(self, inputs):
        return self._bins(self.encoder(self.input_embed(inputs)))


class TGT_Gap(_TaskModel):
    """lib/models/pcqm/gap_predictor.py:10-59"""
    edge_ended = False

    def __init__(self, model_height, layer_multiplier=1, upto_hop=32, embed_3d_type='gaussian',
                 num_3d_kernels=128, **layer_configs):
        super().__init__(model_height, layer_multiplier, upto_hop, embed_3d_type,
                         num_3d_kernels, None, **layer_configs)

    def forward(self, inputs):
        return self._gap(self.encoder(self.input_embed(inputs)))
