"""CPU-side checks of the boundary: the C-ABI library builds/loads and exports
every symbol include/tgt_hip.h declares; the ctypes structs match the header;
the product ops refuse to run without a GPU (no silent fallback)."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_text():
    return open(os.path.join(ROOT, 'include', 'tgt_hip.h')).read()


def test_library_builds_and_exports_every_declared_symbol():
    from tgt_amd import _lib
    _lib.build_library()
    L = _lib.lib()
    declared = set(re.findall(r'\b(tgt_[a-z_0-9]+)\s*\(', header_text()))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(L, name)
    assert L.tgt_abi_version() == _lib.ABI_VERSION


def _c_struct_sizes():
    """Compile a tiny C program against the header to get sizeof/offsetof."""
    import subprocess, tempfile
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "tgt_hip.h"
int main(void){
 printf("%zu %zu %zu\n", sizeof(tgt_triplet_attention_args), sizeof(tgt_triplet_aggregate_args), sizeof(tgt_node_attention_args));
 printf("%zu %zu %zu\n", offsetof(tgt_triplet_attention_args, mask), offsetof(tgt_triplet_attention_args, d_out), offsetof(tgt_triplet_attention_args, d_eg));
 printf("%zu %zu\n", offsetof(tgt_triplet_aggregate_args, out), offsetof(tgt_triplet_aggregate_args, d_eg));
 printf("%zu %zu %zu\n", offsetof(tgt_node_attention_args, eg), offsetof(tgt_node_attention_args, gsum), offsetof(tgt_node_attention_args, d_eg));
 return 0; }'''
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, 't.c')
        open(c, 'w').write(src)
        exe = os.path.join(td, 't')
        subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), c, '-o', exe])
        out = subprocess.check_output([exe]).decode().split()
    return [int(x) for x in out]


def test_ctypes_structs_match_header():
    from tgt_amd import _lib
    v = _c_struct_sizes()
    TA, AG, NA = _lib.TripletAttentionArgs, _lib.TripletAggregateArgs, _lib.NodeAttentionArgs
    assert v[:3] == [C.sizeof(TA), C.sizeof(AG), C.sizeof(NA)]
    assert v[3:6] == [TA.mask.offset, TA.d_out.offset, TA.d_eg.offset]
    assert v[6:8] == [AG.out.offset, AG.d_eg.offset]
    assert v[8:11] == [NA.eg.offset, NA.gsum.offset, NA.d_eg.offset]


def test_invalid_arguments_return_error_codes():
    from tgt_amd import _lib
    L = _lib.lib()
    assert L.tgt_triplet_attention_fwd(None, None) != 0
    assert b'null' in L.tgt_last_error()
    a = _lib.TripletAttentionArgs()
    a.B, a.N, a.H, a.D = 1, 100, 4, 16
    assert L.tgt_triplet_attention_fwd(C.byref(a), None) != 0
    assert L.tgt_adam_step(None, None, None, None, 10, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, 1.0, 0.0, None, None, 0, None) != 0
    assert L.tgt_grad_scaler_step(None, 10, None, None, 1, 0.0, 0.0, 1, 2.0, 0.5, 2000, None) != 0
    assert L.tgt_loss_accumulate(None, 0, 1.0, None, 1, 3, None) != 0


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU behaviour')
def test_ops_fail_loudly_without_gpu():
    from tgt_amd import ops
    L = ops.TripletLayout(32, 4)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ops.triplet_attention(torch.zeros(1, 4, 4, L.width), torch.zeros(1, 4, 4), L)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ops.node_attention(torch.zeros(1, 4, 24), torch.zeros(1, 4, 4, 8), torch.zeros(1, 4, 4), 4)
