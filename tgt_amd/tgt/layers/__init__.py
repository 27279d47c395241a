from .layers import TGT_Layer, EGT_Attention, EdgeUpdate, FFN, DropPath
from .triplet import get_triplet_layer
