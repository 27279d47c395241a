"""The A/B switches of the tgt_amd host side, in ONE place.

Every switch selects between two COMPLETE paths and exists for same-box measurements (`tools/ab_knobs.sh KNOB=v ...`); the
defaults are the measured winners.  The environment is read once, here, when the package is imported: `K` is the resulting
record, the modules bind their private names to its fields (`ops._TRI_PROJ = K.tri_proj`, ...; tests patch those names).
`non_default()` is what `bench.py` puts into its JSON line, so that a number can never be mistaken for the default path's.

Kernel-side switches (`TGT_TRI_BWD2`, `TGT_TRI_BWD2_DMA`: read by the library when it launches; `TGT_HIP_LIB`: which library
`_lib` loads; the `-DTGT_PROBES` ablation variables) are not host knobs and are listed in `ENV_OF_LIBRARY` only for the report.
"""
import os
from dataclasses import dataclass, fields

# field -> (environment variable, default, kind, one line)
_SPEC = {
    'tri_split': ('TGT_TRI_SPLIT', True, 'flag', 'Q/K/V and E/G projected by two GEMMs (only without tri_proj)'),
    'tri_proj': ('TGT_TRI_PROJ', True, 'flag', 'Q/K/V projection inside the triplet forward kernel'),
    'tri_colsum': ('TGT_TRI_COLSUM', True, 'flag', 'bias gradient of the fused projection from the backward kernel'),
    'tri_skip': ('TGT_TRI_SKIP', 1, 'int', 'triplet kernels skip DropPath-dropped graphs: 0 off, 1 forward, 2 backward too'),
    'defer_sums': ('TGT_DEFER_SUMS', False, 'flag', 'closing sums of the backward collected into one launch (measured slower)'),
    'defer_max': ('TGT_DEFER_MAX', 56, 'int', 'sums per queue before it flushes itself'),
    'epi_ln_bwd': ('TGT_EPI_LN_BWD', True, 'flag', 'LayerNorm backward as the epilogue of the data-gradient GEMM'),
    'wgrad_stream': ('TGT_WGRAD_STREAM', False, 'flag', 'parameter gradients on a third stream inside the Trainer backward'),
    'wgrad_keep': ('TGT_WGRAD_KEEP', True, 'flag', 'forked operands kept referenced in a window instead of record_stream (with wgrad_stream)'),
    'wgrad_depth': ('TGT_WGRAD_DEPTH', 8, 'int', 'forks whose operands stay referenced before their origin stream waits for them'),
    'terminal_sums': ('TGT_TERMINAL_SUMS', True, 'flag', 'closing sums on the forked stream too (with wgrad_stream)'),
    'side_prio': ('TGT_SIDE_PRIO', -1, 'int', 'HIP priority of the node side stream (-1 = high)'),
    'wt_cache': ('TGT_WT_CACHE', True, 'flag', 'cached W^T for the data-gradient kernels'),
    'edge_gemm': ('TGT_EDGE_GEMM', True, 'flag', 'edge Linears on tgt_edge_linear'),
    'edge_n512': ('TGT_EDGE_N512', True, 'flag', "lin_O's and the narrow data gradients on own kernels"),
    'ffn_gelu_epi': ('TGT_FFN_GELU_EPI', True, 'flag', 'lin_W1 + GELU + dropout as one launch'),
    'ffn_gelu_bwd_epi': ('TGT_FFN_GELU_BWD_EPI', True, 'flag', 'GELU backward as the epilogue of lin_W2 data gradient'),
    'gelu_bwd_epi_colsum': ('TGT_GELU_BWD_EPI_COLSUM', True, 'flag', "lin_W1's bias gradient from the GELU_BWD epilogue"),
    'edge_k512': ('TGT_EDGE_K512', True, 'flag', 'lin_O (K = 512) + residual + LayerNorm as one launch'),
    'prescale': ('TGT_PRESCALE', True, 'flag', 'DropPath factor folded into the producer of the branch'),
    'stream_keepalive': ('TGT_STREAM_KEEPALIVE', False, 'flag', 'keep cross-stream tensors referenced instead of record_stream'),
    'node_stream': ('TGT_NODE_STREAM', True, 'flag', 'node channel on a second HIP stream'),
    'node_chain': ('TGT_NODE_CHAIN', True, 'flag', "the next layer's node projections chained on the side stream"),
    'flat_grad_dst': ('TGT_FLAT_GRAD_DST', True, 'flag', 'weight gradients written into the flat gradient buffer inside a Trainer backward'),
    'defer_edge': ('TGT_DEFER_EDGE', True, 'flag', 'closing edge residual performed by the next layer entry'),
    'own_gemm': ('TGT_OWN_GEMM', True, 'flag', 'library GEMMs of the step dispatched from cached plans (tgt_amd/gemm.py) instead of through torch: same library call, less host time'),
    'edge_wgrad': ('TGT_EDGE_WGRAD', False, 'flag', 'weight gradient of the 256 x 256 edge Linears inside their data-gradient launch (csrc/edge_wgrad.hip)'),
    'edge_wgrad_spare_cus': ('TGT_EDGE_WGRAD_SPARE_CUS', 0, 'int', 'CUs the fused data + weight gradient launch leaves to the side stream (it takes all 160 KB of LDS where it runs)'),
    'wgrad_maxp': ('TGT_WGRAD_MAXP', 128, 'int', 'cap on the row chunks of a split-M weight gradient (in-step sweep 32..256: 128)'),
    'embed_gemm': ('TGT_EMBED_GEMM', True, 'flag', "per-node mul / bias tables of the Gaussian 3-D embedding without nn.Embedding's sort-based backward (a host read)"),
    'gate_node_bwd': ('TGT_GATE_NODE_BWD', 0, 'int', "the node side stream's backward chain of a layer waits for that layer's triplet backward kernel (1) / for the projection's data-gradient GEMM behind it (2)"),
}
ENV_OF_LIBRARY = ('TGT_TRI_BWD2', 'TGT_TRI_BWD2_DMA', 'TGT_HIP_LIB', 'TGT_NODE_MFMA', 'TGT_NODE_MFMA16', 'TGT_NODE_KB', 'TGT_TUNING_FILE')


def _read(var, default, kind):
    raw = os.environ.get(var)
    if raw is None:
        return default
    if kind == 'int':
        return int(raw)
    # a flag that defaults ON is switched off by "0"; one that defaults OFF is switched on by "1" (the rules the knobs always had)
    return raw != '0' if default else raw == '1'


@dataclass(frozen=True)
class Knobs:
    tri_split: bool
    tri_proj: bool
    tri_colsum: bool
    tri_skip: int
    defer_sums: bool
    defer_max: int
    epi_ln_bwd: bool
    wgrad_stream: bool
    wgrad_keep: bool
    wgrad_depth: int
    terminal_sums: bool
    side_prio: int
    wt_cache: bool
    edge_gemm: bool
    edge_n512: bool
    ffn_gelu_epi: bool
    ffn_gelu_bwd_epi: bool
    gelu_bwd_epi_colsum: bool
    edge_k512: bool
    prescale: bool
    stream_keepalive: bool
    node_stream: bool
    node_chain: bool
    flat_grad_dst: bool
    defer_edge: bool
    own_gemm: bool
    edge_wgrad: bool
    edge_wgrad_spare_cus: int
    wgrad_maxp: int
    embed_gemm: bool
    gate_node_bwd: int

    @classmethod
    def from_env(cls):
        return cls(**{name: _read(*_SPEC[name][:3]) for name in _SPEC})

    def non_default(self):
        """{ENV_VAR: value} of every switch that is not at its default, plus the library-side variables that are set"""
        out = {_SPEC[f.name][0]: getattr(self, f.name) for f in fields(self) if getattr(self, f.name) != _SPEC[f.name][1]}
        out.update({v: os.environ[v] for v in ENV_OF_LIBRARY if v in os.environ})
        return out


assert set(_SPEC) == {f.name for f in fields(Knobs)}
K = Knobs.from_env()
