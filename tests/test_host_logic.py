"""Host-side logic that needs no GPU: the row tables that assemble the kernel-order projections from
the reference's separate nn.Linear parameters, and the numpy restatement of the kernels'
attention-dropout generator."""
import numpy as np
import pytest
import torch

import golden_util as gu


def test_triplet_attention_param_table_matches_reference_channel_order():
    from tgt_amd import layout
    from tgt_amd.tgt.layers.triplet import TripletAttention, TripletAttentionUngated, AxialAttention
    C, H = 64, 8
    D = C // H
    for cls, nb in ((TripletAttention, 2 * H), (TripletAttentionUngated, H), (AxialAttention, 0)):
        mod = cls(C, H)
        t, L = mod._table, mod._layout
        src, idx = t.row_src.tolist(), t.row_idx.tolist()
        assert len(src) == L.width and t.n_cols == C
        for dir_ in (0, 1):
            for part, off in enumerate((L.q[dir_], L.k[dir_], L.v[dir_])):
                for h in range(H):
                    for d in range(D):
                        r = off + h * D + d                       # kernel row: head-major
                        assert src[r] == dir_                      # lin_QKV_in / lin_QKV_out
                        assert idx[r] == part * C + d * H + h      # reference channel: head-minor (triplet.py:213-215)
        if nb:
            for dir_ in (0, 1):
                for j in range(nb):
                    r = 6 * C + dir_ * nb + j
                    assert src[r] == 2 + dir_ and idx[r] == j
        assert all(s == -1 for s in src[L.used:])
        # the permutation really is one: every reference row of the QKV weights is used exactly once
        assert sorted(i for s_, i in zip(src, idx) if s_ == 0) == list(range(3 * C))
        assert torch.equal(torch.tensor(idx[:3 * C]), layout.qkv_rows_head_major(C, H).to(torch.int64))


def test_triplet_aggregate_param_table():
    from tgt_amd.tgt.layers.triplet import TripletAggregate, TripletAggregateUngated
    C, H = 32, 4
    D = C // H
    for cls, nb in ((TripletAggregate, 4 * H), (TripletAggregateUngated, 2 * H)):
        mod = cls(C, H)
        src, idx = mod._table.row_src.tolist(), mod._table.row_idx.tolist()
        for dir_ in (0, 1):
            for h in range(H):
                for d in range(D):
                    r = dir_ * C + h * D + d
                    assert src[r] == 0 and idx[r] == dir_ * C + d * H + h
        assert src[2 * C:2 * C + nb] == [1] * nb and idx[2 * C:2 * C + nb] == list(range(nb))


def test_dropout_generator_restatement_is_deterministic_and_calibrated():
    units = np.arange(40)
    keep, scale = gu.triplet_dropout_keep(0xDEADBEEF12345678, 0.2, units, 32)
    keep2, _ = gu.triplet_dropout_keep(0xDEADBEEF12345678, 0.2, units, 32)
    other, _ = gu.triplet_dropout_keep(0xDEADBEEF12345679, 0.2, units, 32)
    assert keep.shape == (40, 32, 32) and keep.dtype == bool
    assert (keep == keep2).all() and (keep != other).any()
    assert abs(keep.mean() - 0.8) < 0.01 and abs(scale - 1.25) < 1e-6
    # neighbouring keys share one hash word but not their decision
    assert (keep[:, :, 0::2] != keep[:, :, 1::2]).mean() > 0.2


def test_knobs_are_read_once_in_one_place(monkeypatch):
    """tgt_amd/knobs.py: every host-side A/B switch in one record; flags that default on go off with "0", flags that default off go
    on with "1", and non_default() names exactly what differs (bench.py prints it)"""
    from tgt_amd import knobs
    for var, *_ in knobs._SPEC.values():
        monkeypatch.delenv(var, raising=False)
    for var in knobs.ENV_OF_LIBRARY:
        monkeypatch.delenv(var, raising=False)
    base = knobs.Knobs.from_env()
    assert base.non_default() == {}
    assert base.tri_proj and base.node_stream and not base.defer_sums and base.side_prio == -1 and base.tri_skip == 1
    monkeypatch.setenv('TGT_TRI_PROJ', '0')
    monkeypatch.setenv('TGT_DEFER_SUMS', '1')
    monkeypatch.setenv('TGT_WGRAD_STREAM', '0')          # a default-off flag set to its default: not reported
    monkeypatch.setenv('TGT_TRI_SKIP', '2')
    monkeypatch.setenv('TGT_TRI_BWD2_DMA', '0')          # read by the library, listed in the report
    k = knobs.Knobs.from_env()
    assert k.non_default() == {'TGT_TRI_PROJ': False, 'TGT_DEFER_SUMS': True, 'TGT_TRI_SKIP': 2, 'TGT_TRI_BWD2_DMA': '0'}
    import dataclasses
    with pytest.raises(dataclasses.FrozenInstanceError):
        k.tri_proj = True


def test_no_environment_switch_outside_knobs():
    """every host-side switch is a field of knobs.K (so that bench.py's knobs_not_default names it); the
    work-skipping timing probes are not in the product at all (tools/probes/skip_probes.py, bench.py --timing-probe)"""
    import os
    import re
    import tgt_amd
    root = os.path.dirname(tgt_amd.__file__)
    allowed = {'knobs.py': None, '_lib.py': ('TGT_HIP_LIB', 'HIPCC', 'TGT_KEEP_ISA'), 'torch_ops.py': ('CXX',),
               os.path.join('training', 'gemm_tuning.py'): ('TGT_TUNING_FILE',)}
    for d, _, files in os.walk(root):
        for f in files:
            if not f.endswith('.py'):
                continue
            rel = os.path.relpath(os.path.join(d, f), root)
            src = open(os.path.join(d, f)).read()
            assert 'PROBE_SKIP' not in src, rel
            names = re.findall(r"os\.environ(?:\.get)?[\[(]\s*'([A-Z_0-9]+)'", src)
            if rel in allowed:
                assert allowed[rel] is None or set(names) <= set(allowed[rel]), (rel, names)
            else:
                assert not [n for n in names if n.startswith('TGT_')], (rel, names)
    from tgt_amd import knobs
    assert knobs._SPEC['wgrad_maxp'][0] == 'TGT_WGRAD_MAXP' and knobs._SPEC['embed_gemm'][0] == 'TGT_EMBED_GEMM'


def test_parameter_reuse_is_found_in_the_graph():
    """the forked parameter-gradient stream / deferred closing sums are only safe when every parameter that one of the
    package's own autograd Functions differentiates enters the graph once; the Trainer looks for reuse of ANY kind in the graph
    of its first step (not only layer_multiplier > 1)"""
    import torch
    from tgt_amd.training.step import _parameter_reused

    class Lin(torch.autograd.Function):                 # stands for the package's Functions (a Python-defined node)
        @staticmethod
        def forward(ctx, x, w):
            ctx.save_for_backward(x, w)
            return x @ w.t()

        @staticmethod
        def backward(ctx, dy):
            x, w = ctx.saved_tensors
            return dy @ w, dy.t() @ x

    lin, other = torch.nn.Linear(4, 4), torch.nn.Linear(4, 4)
    params = list(lin.parameters()) + list(other.parameters())
    x = torch.randn(3, 4)
    assert not _parameter_reused(Lin.apply(Lin.apply(x, lin.weight), other.weight).sum(), params)
    assert _parameter_reused(Lin.apply(Lin.apply(x, lin.weight), lin.weight).sum(), params)     # a module applied twice
    assert _parameter_reused((Lin.apply(x, lin.weight) @ lin.weight).sum(), params)             # a tied weight, one own edge
    assert not _parameter_reused(lin(lin(x)).sum(), params)            # library nodes only: autograd orders those itself
    assert not _parameter_reused(x.sum().requires_grad_(), params)     # no graph at all


def test_numa_affinity_helper_parses_and_declines_gracefully():
    """bench.py binds a rank to the CPUs of its GPU's NUMA node (tgt_amd/training/affinity.py); without the topology it must
    leave the process alone and say why"""
    import os
    from tgt_amd.training import affinity
    assert affinity._cpulist('0-3,8,10-11\n') == {0, 1, 2, 3, 8, 10, 11}
    assert affinity._cpulist('') == set()
    assert affinity.bind_to_gpu_numa(0, enabled=False) == {'bound': False, 'why': 'disabled'}
    before = os.sched_getaffinity(0)
    real = affinity.gpu_numa_node
    try:
        affinity.gpu_numa_node = lambda i: None              # a container that hides the PCI topology
        rep = affinity.bind_to_gpu_numa(0)
        assert rep['bound'] is False and 'NUMA' in rep['why']
        affinity.gpu_numa_node = lambda i: 10 ** 6           # a node without a cpulist
        rep = affinity.bind_to_gpu_numa(0)
        assert rep['bound'] is False
    finally:
        affinity.gpu_numa_node = real
    assert os.sched_getaffinity(0) == before


def test_graph_safe_host_seeds_are_a_function_of_the_call_position():
    """training/graphed.py: a captured step bakes its host-drawn dropout seeds into the graph, so in graph-safe mode they are a fixed
    function of the call's position inside the step (the step enters through the device counter instead): every step draws the
    same sequence, every call of a step another seed; outside the mode seeds come from torch's CPU generator again"""
    import torch
    from tgt_amd import ops
    try:
        ops.graph_safe_rng(True)
        ops.begin_step()
        a = [ops.draw_dropout(0.1, True)[1] for _ in range(50)]
        ops.begin_step()
        b = [ops.draw_dropout(0.1, True)[1] for _ in range(50)]
        assert a == b and len(set(a)) == 50 and all(0 <= s < 2 ** 63 for s in a)
        assert ops.draw_dropout(0.1, False) == (0.0, 0)            # not training: no dropout, no position consumed
        assert ops.draw_dropout(0.0, True) == (0.0, 0)
        ops.begin_step()
        assert ops.draw_dropout(0.1, True)[1] == a[0]
    finally:
        ops.graph_safe_rng(False)
    torch.manual_seed(5)
    c = ops.draw_dropout(0.1, True)[1]
    torch.manual_seed(5)
    assert ops.draw_dropout(0.1, True)[1] == c and c not in a


def test_flat_gradient_destination_refuses_a_second_gradient_for_one_parameter():
    """inside one Trainer backward a parameter's slice of the flat gradient buffer is handed out ONCE; a graph
    that starts re-using a parameter after the first step (which is the only one the Trainer walks) is an error, not a silent
    double write"""
    import pytest
    import torch
    from tgt_amd import ops
    v = torch.zeros(4, 3)
    ops.register_flat_grads([v], [v])
    try:
        assert ops._grad_dst(v.data_ptr(), (4, 3), torch.float32) is None          # outside a Trainer backward: ordinary tensors
        with ops.trainer_backward():
            a = ops._grad_dst(v.data_ptr(), (4, 3), torch.float32)
            assert a is not None and a.data_ptr() == v.data_ptr()
            with pytest.raises(RuntimeError, match='second weight gradient'):
                ops._grad_dst(v.data_ptr(), (4, 3), torch.float32)
        with ops.trainer_backward():                                                # the next backward starts clean
            assert ops._grad_dst(v.data_ptr(), (4, 3), torch.float32) is not None
    finally:
        ops.unregister_flat_grads([v.data_ptr()])


def test_affinity_binds_every_thread_of_the_process():
    """sched_setaffinity(0, ..) binds the calling thread only; torch's thread pools exist before bench.py binds"""
    import os
    import threading
    import pytest
    from tgt_amd.training import affinity
    if not hasattr(os, 'sched_setaffinity'):
        pytest.skip('no sched_setaffinity')
    before = os.sched_getaffinity(0)
    if len(before) < 2:
        pytest.skip('one CPU')
    stop = threading.Event()
    seen = {}

    def worker():
        stop.wait()
        seen['mask'] = os.sched_getaffinity(0)
    t = threading.Thread(target=worker)
    t.start()
    target = set(sorted(before)[:1])
    try:
        done, total = affinity._bind_all_threads(target)
        assert done >= 2 and total >= 2
        stop.set()
        t.join()
        assert seen['mask'] == target and os.sched_getaffinity(0) == target
    finally:
        stop.set()
        affinity.restore_affinity(before)
    assert os.sched_getaffinity(0) == before


def test_slow_kernel_family_is_announced_once():
    """a BASELINE-sized call that falls off the hot kernels says so, once per reason"""
    import warnings
    from tgt_amd import ops
    ops._slow_path_said.discard(('x', 'y'))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        ops._slow_path_notice(('x', 'y'), 'first')
        ops._slow_path_notice(('x', 'y'), 'second')
    assert len(w) == 1 and 'first' in str(w[0].message) and issubclass(w[0].category, RuntimeWarning)
