"""Layer stack.  API / state_dict prefixes (`TGT_layers.{i}`) = reference
lib/tgt/encoder.py."""
from torch import nn

from .layers import TGT_Layer


class Graph(dict):
    """dict with attribute access (reference lib/tgt/encoder.py:7-21)."""

    def __dir__(self):
        return list(super().__dir__()) + list(self.keys())

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError('No such attribute: ' + key)

    def __setattr__(self, key, value):
        self[key] = value

    def copy(self):
        return self.__class__(self)


class TGT_Encoder(nn.Module):
    class IndivConfig(list):
        """per-layer value list (reference encoder.py:25,55-56)"""

    def __init__(self, model_height=4, layer_multiplier=1, node_ended=True,
                 edge_ended=True, egt_simple=False, **layer_configs):
        super().__init__()
        self.model_height = model_height
        self.layer_multiplier = layer_multiplier
        self.node_ended = node_ended
        self.edge_ended = edge_ended
        self.egt_simple = egt_simple
        self.layer_configs = layer_configs
        for k, v in layer_configs.items():
            setattr(self, k, v)
        assert node_ended or edge_ended, 'At least one of node_ended and edge_ended must be True'
        self.TGT_layers = nn.ModuleList([TGT_Layer(**self.get_layer_kwargs(i))
                                         for i in range(model_height)])

    def get_layer_kwargs(self, i):
        kwargs = {}
        for k, v in self.layer_configs.items():
            if isinstance(v, self.IndivConfig):
                kwargs[k] = v[i]
            elif k == 'drop_path':
                kwargs[k] = v * i / (self.model_height - 1)     # linear ramp (encoder.py:57-58)
            else:
                kwargs[k] = v
        is_last = i == self.model_height - 1
        kwargs['node_update'] = not (is_last and not self.node_ended)
        kwargs['edge_update'] = (not self.egt_simple) and not (is_last and not self.edge_ended)
        return kwargs

    def apply_layer(self, layer_idx, graph):
        layer = self.TGT_layers[layer_idx]
        for _ in range(self.layer_multiplier):      # weight-shared repeats (encoder.py:80-84)
            graph = layer(graph)
        return graph

    def forward(self, inputs):
        g = Graph(inputs)
        for i in range(self.model_height):
            g = self.apply_layer(i, g)
        return g
