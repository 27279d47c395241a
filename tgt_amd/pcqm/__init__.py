"""PCQM4Mv2 task models around the TGT encoder (the producer and the consumers
either side of the hot path).  Same class
names, constructor keywords and state_dict keys as the reference's
lib/models/pcqm package."""
from .models import TGT_Multi, TGT_Distance, TGT_Gap, EmbedInput
