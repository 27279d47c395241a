"""tgt_amd/gemm.py (csrc/gemm_dispatch.cpp): the step's library GEMMs from cached plans.  The plans make the SAME hipBLASLt / rocBLAS
call torch makes (its handle, its workspace, the algorithm of the shipped TunableOp table), so the acceptance test is bit-identity
with torch.mm / torch.addmm / `@` / torch.bmm on the BASELINE shapes -- and that the process still holds ONE hipBLASLt."""
import time

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rnd(g, *shape, dtype=torch.bfloat16, scale=1.0):
    return (torch.randn(*shape, device='cuda', generator=g) * scale).to(dtype)


@pytest.mark.parametrize('dtype', [torch.bfloat16])
def test_plans_reproduce_torch_bit_for_bit_on_the_baseline_shapes(dtype):
    from tgt_amd import gemm
    assert torch.cuda.tunable.is_enabled()                  # (tests/conftest.py: the shipped table, offline)
    g = torch.Generator(device='cuda').manual_seed(3)
    M = 262144
    before = dict(gemm.stats)
    covered = 0
    # forward Linears (TN, bias): node channel (8192 rows) and the edge channel's library shapes
    for rows, K, N in ((8192, 768, 2304), (8192, 768, 768), (M, 256, 1536), (M, 256, 1600), (M, 128, 256), (M, 64, 256)):
        x, w, b = _rnd(g, rows, K), _rnd(g, N, K, scale=K ** -0.5), _rnd(g, N, scale=0.1)
        for bias in (b, None):
            own = gemm.stats['own']
            y = gemm.linear_tn(x, w, bias)
            ref = torch.addmm(bias, x, w.t()) if bias is not None else torch.mm(x, w.t())
            assert torch.equal(y, ref), (rows, K, N, bias is not None)
            covered += gemm.stats['own'] - own
    # data gradients (NN)
    for rows, Nout, Kin in ((M, 1600, 256), (M, 256, 256), (M, 512, 256), (8192, 768, 768), (8192, 2304, 768), (M, 128, 256)):
        dy, w = _rnd(g, rows, Nout), _rnd(g, Nout, Kin, scale=Nout ** -0.5)
        own = gemm.stats['own']
        assert torch.equal(gemm.matmul_nn(dy, w), dy @ w), (rows, Nout, Kin)
        covered += gemm.stats['own'] - own
    # weight gradients: batched row chunks with float32 partials, dy possibly a column slice
    for rows, Nout, Kin, P, cols in ((M, 256, 256, 128, None), (M, 1600, 256, 32, (0, 1536)), (M, 1600, 256, 128, (1536, 1600)),
                                     (M, 256, 512, 64, None), (8192, 768, 768, 8, None)):
        dy, x = _rnd(g, rows, Nout), _rnd(g, rows, Kin)
        d = dy if cols is None else dy[:, cols[0]:cols[1]]
        own = gemm.stats['own']
        part = gemm.wgrad_chunks(d, x, P)
        ref = torch.bmm(d.unflatten(0, (P, rows // P)).transpose(1, 2), x.view(P, rows // P, Kin), out_dtype=torch.float32)
        assert torch.equal(part, ref), (rows, Nout, Kin, P, cols)
        covered += gemm.stats['own'] - own
    torch.cuda.synchronize()
    assert gemm.stats['dropped'] == before['dropped'], 'a plan differed from torch and was dropped'
    assert covered >= 10, (covered, gemm.stats, gemm._BMM_PLAN)      # the table's hipBLASLt / rocBLAS entries are actually taken
    # ONE hipBLASLt / rocBLAS in the process: torch's own copies (the tuned indices are theirs)
    maps = open('/proc/self/maps').read()
    lt = {l.split()[-1] for l in maps.splitlines() if 'libhipblaslt' in l}
    rb = {l.split()[-1] for l in maps.splitlines() if 'librocblas' in l}
    assert len(lt) == 1 and 'torch/lib' in next(iter(lt)), lt
    assert len(rb) == 1 and 'torch/lib' in next(iter(rb)), rb


def test_host_time_per_call_is_lower_than_through_torch():
    """the point of the plans: host time.  GPU kept busy-free with a tiny problem; reported, and asserted only loosely"""
    from tgt_amd import gemm
    g = torch.Generator(device='cuda').manual_seed(4)
    x, w, b = _rnd(g, 8192, 768), _rnd(g, 768, 768, scale=0.03), _rnd(g, 768)
    out = torch.empty(8192, 768, dtype=torch.bfloat16, device='cuda')

    def t(fn, n=400):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        dt = (time.perf_counter() - t0) / n * 1e6
        torch.cuda.synchronize()
        return dt
    own0 = gemm.stats['own']
    t_own = t(lambda: gemm.linear_tn(x, w, b, out=out))
    took_plan = gemm.stats['own'] > own0
    t_torch = t(lambda: torch.addmm(b, x, w.t(), out=out))
    print(f'host us per call: plan {t_own:.1f}  torch {t_torch:.1f}  (plan taken: {took_plan})')
    if took_plan:
        assert t_own < t_torch, (t_own, t_torch)
