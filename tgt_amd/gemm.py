"""Cached-descriptor dispatch of the step's library GEMMs (csrc/gemm_dispatch.cpp).

torch.addmm / torch.mm / `a @ b` / torch.bmm cost 20-37 us of HOST time per call on this stack (TunableOp signature + look-up, three
matrix layouts and a matmul descriptor created and destroyed, a support query: ATen/cuda/tunable/GemmHipblaslt.h) -- 611 calls and
16 ms of the ~70 ms it takes to queue a TGT-At step.  The functions here make the SAME library call -- torch's own hipBLASLt / rocBLAS
handle and workspace, the algorithm / solution index of the shipped TunableOp table (tgt_amd/tuning/tunableop_gfx950.csv) -- from a
plan built once per problem.  Same kernel, same arguments: bit-identical results.  That is CHECKED, once per plan, on its first use:
the plan's result is compared with torch's own call on the same operands (torch.equal); a plan that differs (another library build,
another table) is dropped with a warning and the call goes through torch from then on.  A shape that is not in the table, a table
entry `Default`, a dtype / layout the plans do not cover: torch.  Nothing here computes anything itself.

A/B knob: TGT_OWN_GEMM=0 (knobs.K.own_gemm) -- every call through torch.
"""
import ctypes as C
import os
import warnings

import torch

from .knobs import K

_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}
_NAME = {torch.float32: 'float', torch.bfloat16: 'BFloat16', torch.float16: 'Half'}
_ENABLED = K.own_gemm
_lib = None
_table = None
_plans = {}            # key -> plan id | None (None: this problem goes through torch)
_checked = set()
stats = dict(own=0, torch=0, dropped=0)


def _load():
    global _lib
    if _lib is None:
        from . import torch_ops
        torch_ops.load()
        L = C.CDLL(torch_ops.OPS_LIB_PATH)
        L.tgt_gemm_plan.restype = C.c_int
        L.tgt_gemm_plan.argtypes = [C.c_int, C.c_int, C.c_char, C.c_char] + [C.c_int64] * 6 + [C.c_int] + [C.c_int64] * 3 + [C.c_int] * 3
        L.tgt_gemm_run.restype = C.c_int
        L.tgt_gemm_run.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        L.tgt_gemm_last_error.restype = C.c_char_p
        _lib = L
        global _run_fn
        _run_fn = L.tgt_gemm_run
    return _lib


def _tuning_table():
    """{(op signature, params signature): (backend, index)} of the shipped TunableOp file; the table torch itself loaded must be
    this one (tgt_amd.training.gemm_tuning.enable_gemm_tuning), or the first-use check drops the plans that differ"""
    global _table
    if _table is None:
        from .training.gemm_tuning import TUNING_FILE
        t = {}
        if os.path.exists(TUNING_FILE):
            for line in open(TUNING_FILE):
                f = line.strip().split(',')
                if len(f) < 3 or f[0] == 'Validator':
                    continue
                sol = f[2]
                if sol.startswith('Gemm_Hipblaslt_'):
                    t[(f[0], f[1])] = (0, int(sol[len('Gemm_Hipblaslt_'):]))
                elif sol.startswith('Gemm_Rocblas_'):
                    t[(f[0], f[1])] = (1, int(sol[len('Gemm_Rocblas_'):]))
        _table = t
    return _table


def _tunable_on():
    try:
        return torch.cuda.tunable.is_enabled()
    except Exception:
        return False


def _plan(op, ta, tb, m, n, k, lda, ldb, ldc, dt, bias, batch=1, strides=(0, 0, 0), out_dt=None, heuristic=None):
    """plan id for the TunableOp problem, or None (-> torch).  heuristic = (backend, index) for problems TunableOp does not cover."""
    key = (op, ta, tb, m, n, k, lda, ldb, ldc, dt, bias, batch, strides, out_dt)
    pid = _plans.get(key, -1)
    if pid != -1:
        return pid
    pid = None
    sel = heuristic
    if sel is None and _tunable_on():
        params = f'{ta}{tb}_{m}_{n}_{k}' + (f'_B_{batch}' if batch > 1 else '') + f'_ld_{lda}_{ldb}_{ldc}'
        sel = _tuning_table().get((f'{op}_{_NAME[dt]}_{(ta + tb).upper()}', params))
    if sel is not None and not (sel[0] == 1 and sel[1] < 0 and heuristic is None):
        L = _load()
        r = L.tgt_gemm_plan(sel[0], sel[1], ta.encode(), tb.encode(), m, n, k, lda, ldb, ldc, batch, *strides, _DT[dt],
                            _DT[out_dt or dt], 1 if bias else 0)
        if r >= 0:
            pid = r
        else:
            warnings.warn(f'tgt_amd.gemm: no plan for {key}: {L.tgt_gemm_last_error().decode()} -- this problem stays on torch')
    _plans[key] = pid
    return pid


def _run(pid, key_for_check, a, b, c, bias, reference):
    """run plan `pid` into c; on its first use compare with `reference()` (torch's own call) bit for bit"""
    L = _load()
    st = torch.cuda.current_stream(c.device).cuda_stream
    r = L.tgt_gemm_run(pid, a.data_ptr(), b.data_ptr(), c.data_ptr(), None if bias is None else bias.data_ptr(), 1.0, 0.0, st)
    if r != 0:
        raise RuntimeError(f'tgt_gemm_run: {L.tgt_gemm_last_error().decode()}')
    if pid not in _checked:
        _checked.add(pid)
        if not torch.cuda.is_current_stream_capturing():
            ref = reference()
            if not torch.equal(ref, c):
                stats['dropped'] += 1
                for k_, v in list(_plans.items()):
                    if v == pid:
                        _plans[k_] = None
                warnings.warn(f'tgt_amd.gemm: plan {key_for_check} is not bit-identical to torch on this stack '
                              f'(max |diff| {float((ref.float() - c.float()).abs().max()):.3e}) -- dropped, torch takes this problem')
                c.copy_(ref)
    stats['own'] += 1
    return c


_fast = {}             # (form, shapes / strides / dtype ...) -> plan id | None: the whole eligibility + table look-up, once per problem
_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream_of(t):
    return _raw_stream(t.device.index) if _raw_stream is not None else torch.cuda.current_stream(t.device).cuda_stream


def _go(pid, a, b, c, bias):
    """the per-call path of a plan that passed its first-use check"""
    if _run_fn(pid, a.data_ptr(), b.data_ptr(), c.data_ptr(), None if bias is None else bias.data_ptr(), 1.0, 0.0, _stream_of(c)) != 0:
        raise RuntimeError(f'tgt_gemm_run: {_load().tgt_gemm_last_error().decode()}')
    stats['own'] += 1
    return c


_run_fn = None


def _plans_alive(pid):
    """pid while the plan is still in use, None once the first-use check dropped it"""
    return pid if any(v == pid for v in _plans.values()) else None


def _ok2(*ts):
    return all(t.is_cuda and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1] for t in ts)


def linear_tn(x2, w, bias=None, out=None):
    """out (M, N) = x2 (M, K) @ w (N, K)^T [+ bias (N)]: torch.addmm(bias, x2, w.t(), out=out) / torch.mm(x2, w.t(), out=out)"""
    M, Kd = x2.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=x2.dtype, device=x2.device)
    fk = (0, M, N, Kd, x2.stride(), w.stride(), out.stride(), x2.dtype, w.dtype, bias is None or (bias.dtype, bias.stride()), x2.device.index)
    pid = _fast.get(fk, -1)
    if pid is not None and pid >= 0 and pid in _checked:
        return _go(pid, w, x2, out, bias)
    if pid is None:
        stats['torch'] += 1
        return torch.mm(x2, w.t(), out=out) if bias is None else torch.addmm(bias, x2, w.t(), out=out)
    _fast[fk] = None
    if _ENABLED and x2.dtype in (torch.bfloat16, torch.float16) and w.dtype == x2.dtype and _ok2(x2, w, out) and \
            (bias is None or (bias.dtype == x2.dtype and bias.is_contiguous())):
        pid = _plan('GemmAndBiasTunableOp' if bias is not None else 'GemmTunableOp', 't', 'n', N, M, Kd, w.stride(0), x2.stride(0),
                    out.stride(0), x2.dtype, bias is not None)
        if pid is not None:
            _run(pid, ('tn', N, M, Kd), w, x2, out, bias, lambda: torch.mm(x2, w.t()) if bias is None else torch.addmm(bias, x2, w.t()))
            _fast[fk] = _plans_alive(pid)
            return out
    stats['torch'] += 1
    if bias is None:
        return torch.mm(x2, w.t(), out=out)
    return torch.addmm(bias, x2, w.t(), out=out)


def matmul_nn(a, b):
    """a (M, K) @ b (K, N): the data gradient dY @ W"""
    M, Kd = a.shape
    N = b.shape[1]
    fk = (1, M, N, Kd, a.stride(), b.stride(), a.dtype, b.dtype, a.device.index)
    pid = _fast.get(fk, -1)
    if pid is not None and pid >= 0 and pid in _checked:
        return _go(pid, b, a, torch.empty(M, N, dtype=a.dtype, device=a.device), None)
    if pid is None:
        stats['torch'] += 1
        return a @ b
    _fast[fk] = None
    if _ENABLED and a.dtype in (torch.bfloat16, torch.float16) and b.dtype == a.dtype and _ok2(a, b):
        out = torch.empty(M, N, dtype=a.dtype, device=a.device)
        pid = _plan('GemmTunableOp', 'n', 'n', N, M, Kd, b.stride(0), a.stride(0), N, a.dtype, False)
        if pid is not None:
            _run(pid, ('nn', N, M, Kd), b, a, out, None, lambda: a @ b)
            _fast[fk] = _plans_alive(pid)
            return out
    stats['torch'] += 1
    return a @ b


def wgrad_chunks(dy2, x2, P):
    """(P, N, K) float32 = per row chunk c: dy2[c]^T @ x2[c], dy2 (M, N) (rows may be strided: a column slice), x2 (M, K) contiguous,
    16-bit: torch.bmm(dy2.view(P, M/P, N).transpose(1, 2), x2.view(P, M/P, K), out_dtype=torch.float32)"""
    M, N = dy2.shape
    Kd = x2.shape[1]
    m = M // P

    def reference():
        return torch.bmm(dy2.unflatten(0, (P, m)).transpose(1, 2), x2.view(P, m, Kd), out_dtype=torch.float32)
    fk = (2, M, N, Kd, P, dy2.stride(), x2.stride(), dy2.dtype, x2.dtype, dy2.device.index)
    pid = _fast.get(fk, -1)
    if pid is not None and pid >= 0 and pid in _checked:
        return _go(pid, x2, dy2, torch.empty(P, N, Kd, dtype=torch.float32, device=dy2.device), None)
    if pid is None:
        stats['torch'] += 1
        return reference()
    if _bmm_calibrated[0] or not dy2.is_cuda or torch.cuda.is_current_stream_capturing():
        _fast[fk] = None                     # (decided below; before the calibration has run nothing is recorded)
    if _ENABLED and not _bmm_calibrated[0] and dy2.is_cuda and not torch.cuda.is_current_stream_capturing():
        calibrate_bmm(dy2.device)
    if _ENABLED and _BMM_PLAN[0] is not None and dy2.dtype in (torch.bfloat16, torch.float16) and x2.dtype == dy2.dtype and \
            _ok2(dy2, x2) and x2.is_contiguous() and M % P == 0:
        # column-major: C_c^T (K x N) = X_c^T (K x m, no transpose, lda = K) . dY_c (m x N as stored = N x m column-major: transposed)
        pid = _plan('bmm_f32', 'n', 't', Kd, N, m, Kd, dy2.stride(0), Kd, dy2.dtype, False, batch=P,
                    strides=(m * Kd, m * dy2.stride(0), N * Kd), out_dt=torch.float32, heuristic=_BMM_PLAN[0])
        if pid is not None:
            out = torch.empty(P, N, Kd, dtype=torch.float32, device=dy2.device)
            _run(pid, ('bmm', Kd, N, m, P), x2, dy2, out, None, reference)
            _fast[fk] = _plans_alive(pid)
            return out
    _fast[fk] = None
    stats['torch'] += 1
    return reference()


# how torch itself runs a 16-bit bmm with a float32 result on this stack is not in the TunableOp table: (backend, index) found by
# tgt_amd.gemm.calibrate_bmm() -- the first candidate whose result equals torch's bit for bit -- or None (torch keeps these calls)
_BMM_PLAN = [None]
_bmm_calibrated = [False]


def calibrate_bmm(device='cuda'):
    """try the library calls torch could be making for bmm(16-bit, 16-bit) -> float32 on one weight-gradient problem and keep the
    one that reproduces torch's bits (hipBLASLt's first heuristic choice; rocBLAS's standard algorithm); None if neither does"""
    _bmm_calibrated[0] = True
    if not _ENABLED:
        return None
    g = torch.Generator(device=device).manual_seed(7)
    P, m, N, Kd = 8, 2048, 256, 256
    dy = torch.randn(P * m, N, device=device, generator=g).to(torch.bfloat16)
    x = torch.randn(P * m, Kd, device=device, generator=g).to(torch.bfloat16)
    ref = torch.bmm(dy.view(P, m, N).transpose(1, 2), x.view(P, m, Kd), out_dtype=torch.float32)
    for cand in ((0, -1), (1, -1)):
        L = _load()
        pid = L.tgt_gemm_plan(cand[0], cand[1], b'n', b't', Kd, N, m, Kd, N, Kd, P, m * Kd, m * N, N * Kd, 1, 0, 0)
        if pid < 0:
            continue
        out = torch.empty(P, N, Kd, dtype=torch.float32, device=device)
        if L.tgt_gemm_run(pid, x.data_ptr(), dy.data_ptr(), out.data_ptr(), None, 1.0, 0.0,
                          torch.cuda.current_stream(out.device).cuda_stream) == 0 and torch.equal(out, ref):
            _BMM_PLAN[0] = cand
            return cand
    _BMM_PLAN[0] = None
    return None
