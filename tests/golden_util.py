"""Shared between tools/make_golden.py (runs the real reference, in the build
container only) and the tests (which never see the reference): how golden
cases are named, how their inputs and parameters are regenerated from a seed.

Inputs and parameters come from numpy's default_rng so they are bit-identical
on every host; the .npz fixtures hold only OUTPUTS (full tensors for small
cases, a fixed sample of elements + L2 norms for wide ones).
"""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def fill_params(module, seed):
    """Overwrite every parameter/buffer of `module` from default_rng(seed).
    Keys are visited in sorted order so two modules with the same state_dict
    schema (reference / oracle / product) get identical values."""
    rng = np.random.default_rng(seed)
    sd = module.state_dict()
    new = {}
    for key in sorted(sd.keys()):
        t = sd[key]
        shape = tuple(t.shape)
        x = rng.standard_normal(shape)
        leaf = key.split('.')[-2] if '.' in key else key
        if t.ndim >= 2:
            x = x * (0.7 / np.sqrt(shape[-1]))
            if 'embed' in key and 'gbf' not in key:
                x = rng.standard_normal(shape) * 0.3
            if key.endswith('gbf.means.weight') or key.endswith('gbf.stds.weight'):
                x = rng.uniform(0.2, 3.0, shape)
            if key.endswith('gbf.mul.weight'):
                x = 1.0 + 0.1 * rng.standard_normal(shape)
            if key.endswith('gbf.bias.weight'):
                x = 0.1 * rng.standard_normal(shape)
        elif key.endswith('.weight') and ('ln' in leaf):
            x = 1.0 + 0.1 * x
        elif key == 'angular_freqs' or key.endswith('angular_freqs'):
            new[key] = t.clone()
            continue
        else:
            x = 0.1 * x
        new[key] = torch.from_numpy(np.ascontiguousarray(x)).to(t.dtype)
    module.load_state_dict(new, strict=True)
    return module


def additive_mask(num_nodes, N, dtype):
    """(B,N,N,1) mask = (1-edge_mask)*finfo.min, as EmbedInput builds it
    (reference lib/models/pcqm/layers.py:78-80)."""
    nm = (torch.arange(N)[None, :] < torch.as_tensor(num_nodes)[:, None])
    em = (nm[:, :, None] & nm[:, None, :]).to(dtype)
    return ((1 - em) * torch.finfo(dtype).min).unsqueeze(-1)


def op_inputs(B, N, W, C, num_nodes, seed, dtype=torch.float64):
    rng = np.random.default_rng(seed)
    h = torch.from_numpy(rng.standard_normal((B, N, W))).to(dtype)
    e = torch.from_numpy(rng.standard_normal((B, N, N, C))).to(dtype)
    gh = torch.from_numpy(rng.standard_normal((B, N, W))).to(dtype)      # cotangents
    ge = torch.from_numpy(rng.standard_normal((B, N, N, C))).to(dtype)
    mask = additive_mask(num_nodes, N, dtype)
    return h, e, mask, gh, ge


# name -> (class name, ctor kwargs, geometry)
SMALL = dict(B=2, N=6, W=48, C=32, num_nodes=[6, 4])
WIDE = dict(B=2, N=20, W=768, C=256, num_nodes=[20, 13])

OP_CASES = {
    # tiny, full tensors
    'egt_small': ('EGT_Attention', dict(node_width=48, edge_width=32, num_heads=4), SMALL),
    'egt_small_nodeg': ('EGT_Attention', dict(node_width=48, edge_width=32, num_heads=4, scale_degree=False, edge_update=False), SMALL),
    'edgeupd_small': ('EdgeUpdate', dict(node_width=48, edge_width=32, num_heads=4), SMALL),
    'tri_att_small': ('TripletAttention', dict(edge_width=32, num_heads=4), SMALL),
    'tri_att_ungated_small': ('TripletAttentionUngated', dict(edge_width=32, num_heads=4), SMALL),
    'axial_small': ('AxialAttention', dict(edge_width=32, num_heads=4), SMALL),
    'tri_agg_small': ('TripletAggregate', dict(edge_width=32, num_heads=4), SMALL),
    'tri_agg_ungated_small': ('TripletAggregateUngated', dict(edge_width=32, num_heads=4), SMALL),
    'triupd_small': ('TriangularUpdate', dict(edge_width=32, num_heads=4), SMALL),
    'ffn_small': ('FFN', dict(width=32, multiplier=2., activation='gelu'), SMALL),
    'ffn_geglu_small': ('FFN', dict(width=32, multiplier=1., activation='geglu'), SMALL),
    # BASELINE widths, sampled
    'egt_wide': ('EGT_Attention', dict(node_width=768, edge_width=256, num_heads=64), WIDE),
    'tri_att_wide': ('TripletAttention', dict(edge_width=256, num_heads=16), WIDE),
    'tri_agg_wide': ('TripletAggregate', dict(edge_width=256, num_heads=16), WIDE),
}

NODE_OPS = {'EGT_Attention', 'EdgeUpdate'}
N_SAMPLES = 512


def sample_index(numel, seed=7):
    rng = np.random.default_rng(seed + numel)
    return rng.integers(0, numel, size=min(N_SAMPLES, numel))


def summarize(t, full):
    """What is stored for tensor t: everything (small) or samples + norm."""
    a = t.detach().double().cpu().numpy()
    if full:
        return dict(full=a)
    flat = a.reshape(-1)
    return dict(samples=flat[sample_index(flat.size)], norm=np.array(np.linalg.norm(flat)),
                shape=np.array(a.shape))


def run_op_case(cls, kwargs, geom, seed):
    """Build module (float64), run forward + backward with fixed cotangents.
    Returns dict name -> tensor (outputs, input grads, parameter grads)."""
    mod = fill_params(cls(**kwargs).double(), seed)
    mod.eval()
    h, e, mask, gh, ge = op_inputs(geom['B'], geom['N'], geom['W'], geom['C'],
                                   geom['num_nodes'], seed + 1)
    h.requires_grad_(True)
    e.requires_grad_(True)
    name = cls.__name__
    res = {}
    if name in NODE_OPS:
        ho, eo = mod(h, e, mask)
        loss = 0.
        if ho is not h:
            res['out_h'] = ho
            loss = loss + (ho * gh).sum()
        if eo is not e:
            res['out_e'] = eo
            loss = loss + (eo * ge).sum()
    elif name == 'FFN':
        eo = mod(e)
        res['out_e'] = eo
        loss = (eo * ge).sum()
    else:
        eo = mod(e, mask)
        res['out_e'] = eo
        loss = (eo * ge).sum()
    loss.backward()
    if h.grad is not None:
        res['grad_h'] = h.grad
    if e.grad is not None:
        res['grad_e'] = e.grad
    for k, p in mod.named_parameters():
        if p.grad is not None:
            res['pgrad.' + k] = p.grad
    return res


# ---- model-level cases ----------------------------------------------------
TINY_LAYER_CFG = dict(node_width=48, edge_width=32, num_heads=4, activation='gelu',
                      scale_degree=True, triplet_heads=4, triplet_dropout=0,
                      node_ffn_multiplier=1., edge_ffn_multiplier=1.,
                      source_dropout=0, drop_path=0, node_act_dropout=0, edge_act_dropout=0)

MODEL_CASES = {
    # name: (class, kwargs, batch geometry)
    'multi_at_tiny': ('TGT_Multi', dict(model_height=3, layer_multiplier=1, upto_hop=32,
                                        embed_3d_type='gaussian', num_3d_kernels=16, num_dist_bins=24,
                                        triplet_type='attention', **TINY_LAYER_CFG),
                      dict(B=3, N=7, num_nodes=[7, 5, 3])),
    'dist_agx2_tiny': ('TGT_Distance', dict(model_height=3, layer_multiplier=2, upto_hop=32,
                                            embed_3d_type='gaussian', num_3d_kernels=16, num_dist_bins=24,
                                            triplet_type='aggregate', **TINY_LAYER_CFG),
                       dict(B=3, N=7, num_nodes=[7, 5, 3])),
    'gap_at_tiny': ('TGT_Gap', dict(model_height=3, layer_multiplier=1, upto_hop=32,
                                    embed_3d_type='fourier', num_3d_kernels=16,
                                    triplet_type='attention', **TINY_LAYER_CFG),
                    dict(B=3, N=7, num_nodes=[7, 5, 3])),
}

# full TGT-At 24L width (BASELINE cfg 2 architecture) on a 2-graph batch; sampled
FULL_AT_CFG = dict(model_height=24, layer_multiplier=1, upto_hop=32, embed_3d_type='gaussian',
                   num_3d_kernels=128, num_dist_bins=512, node_width=768, edge_width=256,
                   num_heads=64, activation='gelu', scale_degree=True, triplet_heads=16,
                   triplet_type='attention', triplet_dropout=0, node_ffn_multiplier=1.,
                   edge_ffn_multiplier=1., source_dropout=0, drop_path=0,
                   node_act_dropout=0, edge_act_dropout=0)


# BASELINE config 1: TGT-Agx2 12 shared layers x 2 distance predictor at full width (reference configs/pcqm/tgt_agx2_100m/dist_pred/
# tgt_agx2_dp_nordkit.yaml: coords_input none, 256 bins, aggregate triplets) on an 8-graph ragged mini-batch, N <= 32
FULL_AGX2_CFG = dict(model_height=12, layer_multiplier=2, upto_hop=32, embed_3d_type='none',
                     num_3d_kernels=128, num_dist_bins=256, node_width=768, edge_width=256,
                     num_heads=64, activation='gelu', scale_degree=True, triplet_heads=16,
                     triplet_type='aggregate', triplet_dropout=0, node_ffn_multiplier=1.,
                     edge_ffn_multiplier=1., source_dropout=0, drop_path=0,
                     node_act_dropout=0, edge_act_dropout=0)
FULL_AGX2_GEOM = dict(B=8, N=32, num_nodes=[32, 29, 27, 24, 21, 18, 14, 9])
# the BASELINE config-2 architecture at the benchmark's node count: TGT-At 24L, B = 2, N = 32 (one ragged graph)
FULL_AT_N32_GEOM = dict(B=2, N=32, num_nodes=[32, 27])
# ... and of BASELINE config 4: N up to 48 (the 16-wide triplet kernels, the lane-per-head node attention)
FULL_AT_N48_GEOM = dict(B=2, N=48, num_nodes=[48, 39])
# BASELINE config 4 as a mini-batch: 8 ragged graphs, N <= 48, Gaussian 3-D embedding of (RDKit-like) coordinates
FULL_AT_N48_B8_GEOM = dict(B=8, N=48, num_nodes=[48, 45, 41, 37, 33, 30, 24, 17])
# BASELINE config 5, second stage: the TGT-Agx2 12 x 2 GAP predictor at full width (reference configs/pcqm/tgt_agx2_100m/gap_pred/
# tgt_agx2_tp_nordkit.yaml; lib/models/pcqm/gap_predictor.py:10-63), fed with distances that went through the bins format
# (bins2dist of 256-bin indices, lib/training_schemes/pcqm/gap_pred/scheme.py:70-73), 8 ragged graphs
FULL_GAP_AGX2_CFG = dict(model_height=12, layer_multiplier=2, upto_hop=32, embed_3d_type='gaussian',
                         num_3d_kernels=128, node_width=768, edge_width=256,
                         num_heads=64, activation='gelu', scale_degree=True, triplet_heads=16,
                         triplet_type='aggregate', triplet_dropout=0, node_ffn_multiplier=1.,
                         edge_ffn_multiplier=1., source_dropout=0, drop_path=0,
                         node_act_dropout=0, edge_act_dropout=0)


def binned_dist_input(batch, num_bins=256, range_bins=8.0):
    """the gap stage's distance input as the reference builds it: distances -> bin indices (commons.discrete_dist, float32
    arithmetic) -> upper triangle kept (bin_ops packs triu only) -> bins2dist with shift_half and zero_diag
    (commons.BinsProcessor.bins2dist: (bins + 0.5) * bin_size, symmetrised, zero diagonal)"""
    d = batch['dist_input'].float()
    bins = (d * ((num_bins - 1) / range_bins)).long().clamp(0, num_bins - 1)
    bins = torch.triu(bins, 1).float()
    bin_size = range_bins / (num_bins - 1)
    dist = (bins + 0.5) * bin_size                   # (every element, the empty lower triangle included: commons.py:72-82)
    dist = dist + dist.transpose(-2, -1)
    return dist * (1 - torch.eye(dist.size(-1), dtype=dist.dtype))


def model_batch(geom, seed):
    """Synthetic batch + the two keys the scheme adds on device
    (reference lib/training_schemes/pcqm/pretrain/scheme.py:60-76), without
    coordinate noise so that the case is deterministic."""
    from tgt_amd.training.synthetic import make_batch
    b = make_batch(geom['B'], geom['N'], seed, num_nodes=geom['num_nodes'])
    nm = b['node_mask']
    b['edge_mask'] = nm.unsqueeze(-1) * nm.unsqueeze(-2)
    c = b['dft_coords']
    b['dist_input'] = torch.norm(c.unsqueeze(-2) - c.unsqueeze(-3), dim=-1)
    return b


GRAD_PROBE_KEYS = [
    'encoder.TGT_layers.0.update.lin_QKV.weight',
    'encoder.TGT_layers.0.update.lin_EG.weight',
    'encoder.TGT_layers.0.update.lin_O_e.weight',
    'encoder.TGT_layers.1.tria.tri_ln_e.weight',
    'encoder.TGT_layers.1.tria.lin_O.weight',
    'encoder.TGT_layers.1.edge_ffn.lin_W1.bias',
    'encoder.TGT_layers.2.node_ffn.lin_W2.weight',
    'input_embed.dist_embed.weight',
    'input_embed.nodef_embed.weight',
]

# parameters whose gradients test_full_width_24L_training_gradients_vs_oracle compares (TGT-At 24L)
FULL_GRAD_KEYS = [
    'encoder.TGT_layers.0.update.lin_QKV.weight', 'encoder.TGT_layers.0.tria.lin_QKV_in.weight',
    'encoder.TGT_layers.5.tria.lin_EG_out.weight', 'encoder.TGT_layers.11.tria.tri_ln_e.weight',
    'encoder.TGT_layers.17.edge_ffn.lin_W1.weight', 'encoder.TGT_layers.23.tria.lin_O.weight',
    'encoder.TGT_layers.23.update.lin_O_e.bias', 'input_embed.dist_embed.weight', 'dist_pred.weight',
]


# ---- MC-sampled prediction cases ------------------------------------------------------------
PREDICT_CASE = dict(B=3, N=7, num_nodes=[7, 5, 3], num_bins=24, range_bins=8, nb_samples=3, nan_at=(1,), seed=4711)


def predict_logit_sequence(case=None, dtype=torch.float32):
    """the stochastic forwards of a stand-in distance predictor: 2*S logit tensors (B, N, N, bins) from a seed; the ones
    listed in `nan_at` carry a NaN (sample skipped by the prediction loops).  Sharp enough that argmax ties are absent."""
    c = case or PREDICT_CASE
    rng = np.random.default_rng(c['seed'])
    seq = []
    for t in range(2 * c['nb_samples']):
        x = rng.standard_normal((c['B'], c['N'], c['N'], c['num_bins'])) * 3.0
        if t in c['nan_at']:
            x[c['B'] - 1, 0, 1, 2] = np.nan
        seq.append(torch.from_numpy(x).to(dtype))
    return seq


def predict_gap_sequence(case=None):
    """stand-in gap predictor: value depends on the dist_input it is given (so that the sample -> bins-sample cycling of
    the gap prediction loop is observable); call 2 returns an Inf"""
    c = case or PREDICT_CASE
    calls = []

    def model(batch):
        t = len(calls)
        calls.append(t)
        g = batch['dist_input'].double().sum((-1, -2)) * 1e-2 + t
        if t == 2:
            g = g.clone()
            g[0] = float('inf')
        return g
    return model


def bf16_drift(case):
    """{tensor name: rel-L2 drift of the REFERENCE under bf16 autocast vs itself without} for a golden case
    (tests/golden/bf16_drift.npz, written by tools/make_golden.py drift)"""
    z = np.load(os.path.join(GOLDEN_DIR, 'bf16_drift.npz'))
    return {k.split('::', 1)[1]: float(z[k]) for k in z.files if k.startswith(case + '::')}


# ---------------------------------------------------------------------------
# numpy restatement of the kernels' attention-dropout generator
# (tgt_amd/csrc/triplet_common.hpp: tri_drop_bits) -- test infrastructure
# ---------------------------------------------------------------------------
def _mix32(h):
    import numpy as np
    h = h.astype(np.uint64) & 0xFFFFFFFF
    h ^= h >> 16
    h = (h * 0x7feb352d) & 0xFFFFFFFF
    h ^= h >> 15
    h = (h * 0x846ca68b) & 0xFFFFFFFF
    h ^= h >> 16
    return h


def triplet_dropout_keep(seed, p, units, N):
    """keep[u, i, k] (bool) and the scale 1/(1-p) for the given `unit` numbers (1-D int array):
    attention: unit = ((b*2 + dir)*H + h)*N + j;  aggregate: unit = (b*2 + dir)*H + h"""
    import numpy as np
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    lo, hi = seed & 0xFFFFFFFF, seed >> 32
    thresh = int(min(65535, max(1, np.rint(np.float32(p) * np.float32(65536.0)))))
    units = np.asarray(units, dtype=np.uint64)
    base = (_mix32(np.uint64(lo) ^ _mix32(units)) + np.uint64(hi)) & 0xFFFFFFFF            # (U,)
    i = np.arange(N, dtype=np.uint64)[:, None]
    k = np.arange(N, dtype=np.uint64)[None, :]
    word = (i * 64 + k) >> 1                                                                # (N, N)
    r = _mix32((base[:, None, None] + word[None] * 0x9e3779b9) & 0xFFFFFFFF)                # (U, N, N)
    bits = np.where((k & 1)[None] == 1, r >> 16, r & 0xFFFF)
    return bits >= thresh, 1.0 / (1.0 - float(np.float32(p)))
