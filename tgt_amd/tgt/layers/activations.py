"""Activation lookup with the reference's names (lib/tgt/layers/activations.py:4-25).
GLU-family activations consume a doubled lin_W1 width (second return value)."""
import torch
import torch.nn.functional as F


def geglu(x):
    gate, lin = x.chunk(2, dim=-1)
    return lin * F.gelu(gate)


def glu(x):
    gate, lin = x.chunk(2, dim=-1)
    return lin * torch.sigmoid(gate)


def swiglu(x):
    gate, lin = x.chunk(2, dim=-1)
    return lin * torch.sigmoid(gate) * gate


glu_dict = {'geglu': geglu, 'glu': glu, 'swiglu': swiglu}


def get_activation(activation):
    if activation in glu_dict:
        return glu_dict[activation], 2
    return getattr(F, activation), 1
