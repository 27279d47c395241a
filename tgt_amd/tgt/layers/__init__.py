from .blocks import TGT_Layer, EGT_Attention, EdgeUpdate, FFN, DropPath, LayerNorm, Linear, get_activation
from .triplet import get_triplet_layer
