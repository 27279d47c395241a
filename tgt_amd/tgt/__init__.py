"""MI355X-native mirror of the reference's `lib.tgt` package: same names
(`from tgt_amd.tgt import TGT_Encoder, Graph`), same constructor keywords,
same `state_dict` keys; the hot arithmetic runs in libtgt_hip.so."""
from .encoder import TGT_Encoder, Graph
