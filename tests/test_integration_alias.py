"""The drop-in seam described in INTEGRATION.md, exercised against the REAL reference when it
is present (build container only; skipped on the GPU box where /root/reference does not
exist): alias `lib.tgt` -> `tgt_amd.tgt`, build the reference's own task model on top of it,
and check that its state_dict schema is unchanged (so reference checkpoints load)."""
import importlib
import os
import sys

import pytest

REF = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'lib', 'tgt')),
                                reason='reference checkout not available')


def _purge(prefix):
    for k in [k for k in sys.modules if k == prefix or k.startswith(prefix + '.')]:
        del sys.modules[k]


def test_reference_task_model_builds_on_the_hip_mirror():
    import golden_util as gu
    _purge('lib')
    sys.path.insert(0, REF)
    try:
        ref_sd = None
        multitask = importlib.import_module('lib.models.pcqm.multitask')
        ref_model = multitask.TGT_Multi(**gu.MODEL_CASES['multi_at_tiny'][1])
        ref_sd = {k: tuple(v.shape) for k, v in ref_model.state_dict().items()}
        ref_layer_cls = type(ref_model.encoder.TGT_layers[0])

        _purge('lib')
        import tgt_amd.tgt
        import tgt_amd.tgt.layers
        sys.modules['lib.tgt'] = tgt_amd.tgt
        sys.modules['lib.tgt.layers'] = tgt_amd.tgt.layers
        multitask = importlib.import_module('lib.models.pcqm.multitask')     # the reference file, unchanged
        model = multitask.TGT_Multi(**gu.MODEL_CASES['multi_at_tiny'][1])
        layer_cls = type(model.encoder.TGT_layers[0])
        assert layer_cls.__module__.startswith('tgt_amd.') and layer_cls is not ref_layer_cls
        sd = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        assert sd == ref_sd
        model.load_state_dict(ref_model.state_dict(), strict=True)             # a reference checkpoint loads
    finally:
        _purge('lib')
        if REF in sys.path:
            sys.path.remove(REF)
