"""The layer stack (`TGT_Encoder`) and the attribute-style batch container (`Graph`).
Same public behaviour and state_dict prefixes (`TGT_layers.{i}.`) as the reference's
lib/tgt/encoder.py."""
from torch import nn

from ..knobs import K
from .layers import TGT_Layer
from .layers.blocks import PendingResidual


_DEFER_EDGE = K.defer_edge      # A/B knob (tgt_amd/knobs.py)


class Graph(dict):
    """A dict whose keys read and write as attributes (g.h, g.e, g.mask, ...)."""

    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError('No such attribute: ' + name)

    __setattr__ = dict.__setitem__

    def __dir__(self):
        return [*super().__dir__(), *self]

    def copy(self):
        return type(self)(self)


class TGT_Encoder(nn.Module):
    """`model_height` TGT layers, each applied `layer_multiplier` times in a row with shared
    weights.  Per-layer keyword handling follows reference encoder.py:52-78: `IndivConfig`
    lists give one value per layer, `drop_path` ramps linearly with depth, and the last layer
    drops its node (edge) half when the model is not node (edge) ended."""

    class IndivConfig(list):
        pass

    def __init__(self, model_height=4, layer_multiplier=1, node_ended=True,
                 edge_ended=True, egt_simple=False, **layer_configs):
        super().__init__()
        if not (node_ended or edge_ended):
            raise AssertionError('At least one of node_ended and edge_ended must be True')
        vars(self).update(model_height=model_height, layer_multiplier=layer_multiplier,
                          node_ended=node_ended, edge_ended=edge_ended, egt_simple=egt_simple,
                          layer_configs=layer_configs, **layer_configs)
        self.TGT_layers = nn.ModuleList(TGT_Layer(**self.get_layer_kwargs(i)) for i in range(model_height))

    def _value_for_layer(self, key, value, depth):
        if isinstance(value, self.IndivConfig):
            return value[depth]
        if key == 'drop_path':                       # 0 at the first layer ... `value` at the last
            return value * depth / (self.model_height - 1)
        return value

    def get_layer_kwargs(self, i):
        kwargs = {k: self._value_for_layer(k, v, i) for k, v in self.layer_configs.items()}
        last = i == self.model_height - 1
        kwargs.update(node_update=self.node_ended or not last,
                      edge_update=not self.egt_simple and (self.edge_ended or not last))
        return kwargs

    def apply_layer(self, layer_idx, graph, defer_edge=False):
        for _ in range(self.layer_multiplier):
            graph = self.TGT_layers[layer_idx](graph, defer_edge=defer_edge)
        return graph

    def forward(self, inputs):
        graph = Graph(inputs)
        for idx in range(self.model_height):
            # between layers the closing edge residual travels un-added (layers.PendingResidual)
            graph = self.apply_layer(idx, graph, defer_edge=_DEFER_EDGE)
        if isinstance(graph.e, PendingResidual):
            graph.e = graph.e.materialize()
        side = graph.pop('node_side', None)
        if side is not None:
            side.join(graph.h)
        return graph
