"""MI355X-native mirror of the reference's `lib.tgt` package: same names
(`from tgt_amd.tgt import TGT_Encoder, Graph`), same constructor keywords,
same `state_dict` keys; the hot arithmetic runs in libtgt_hip.so."""
from .stack import TGT_Encoder, Graph
from . import stack as encoder          # `lib.tgt.encoder` alias
