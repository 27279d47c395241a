"""torch.autograd bindings of the HIP kernels (C ABI in include/tgt_hip.h).

These functions are the only way the tgt_amd modules do the hot-path
arithmetic.  They require HIP device tensors and libtgt_hip.so; there is no
eager / CPU fallback (a missing library or a CPU tensor raises).
"""
import collections
import contextlib
import ctypes as C
import os
import weakref

import torch

from . import _lib
from . import gemm as _gemm
from .knobs import K

_DT = {torch.float32: _lib.TGT_F32, torch.bfloat16: _lib.TGT_BF16, torch.float16: _lib.TGT_F16}


# optional per-kernel timing with HIP events on the launch stream (bench.py's roofline leg)
_PROFILE = None


_TRI_SPLIT = K.tri_split        # A/B knobs: tgt_amd/knobs.py reads the environment once; tests patch these names
_TRI_PROJ = K.tri_proj
_TRI_COLSUM = K.tri_colsum
# graph_scale (DropPath-dropped graphs skipped by the triplet kernels) reaches the BACKWARD kernel only with TGT_TRI_SKIP=2: at the
# BASELINE shapes the backward is 1024 workgroups in exactly four rounds on 256 CUs (one workgroup per CU), and with ~10 % of them
# finishing early the last round is still a round -- measured 0.474 vs 0.476 ms -- so by default it keeps computing every graph (the
# forward, two workgroups per CU, gains 4 %)
_TRI_SKIP_BWD = K.tri_skip == 2


_PROFILE_ONLY = None     # names to time (None: every kernel that offers itself)
_PROFILE_STRIDE = 1      # every k-th launch of a name is timed
_profile_seen = {}


def profile_kernels(enable=True, only=None, stride=1):
    """Start (returns the dict that will fill with name -> [(start,end) events]) or stop.  only: an iterable of kernel names --
    every other launch goes out without events; stride: of those, every stride-th launch per name is timed.  An event pair is not
    free: two marker packets in the queue and ~3 us of host time per launch -- with all 1400 launches of a step timed the step
    measured 81.66 ms, with the roofline kernels only 79.53 (same box, alternating: profiles/r06e_ab_host.txt); bench.py therefore
    times just the kernels its roofline leg reports, one launch in five (a stride coprime with the 24 layers)."""
    global _PROFILE, _PROFILE_ONLY, _PROFILE_STRIDE
    _PROFILE = {} if enable else None
    _PROFILE_ONLY = frozenset(only) if (enable and only is not None) else None
    _PROFILE_STRIDE = max(1, int(stride)) if enable else 1
    _profile_seen.clear()
    return _PROFILE


def profile_launch_counts():
    """name -> launches seen since profile_kernels(True, stride > 1) (timed or not); empty with stride 1"""
    return dict(_profile_seen)


def _timed(name):
    """is this launch of `name` one the running profile wants events around?"""
    if _PROFILE is None or (_PROFILE_ONLY is not None and name not in _PROFILE_ONLY):
        return False
    if _PROFILE_STRIDE == 1:
        return True
    n = _profile_seen.get(name, 0)
    _profile_seen[name] = n + 1
    return n % _PROFILE_STRIDE == 0


def kernel_times_ms(prof):
    """After torch.cuda.synchronize(): name -> list of per-launch durations (ms)."""
    return {k: [s.elapsed_time(e) for s, e in v] for k, v in prof.items()}


def _call(name, fn, args):
    if _PROFILE is None or not _timed(name):
        _lib.check(fn(C.byref(args), _stream()), name)
        return
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    _lib.check(fn(C.byref(args), _stream()), name)
    e.record()
    _PROFILE.setdefault(name, []).append((s, e))


def _dev(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('tgt_amd ops run on the MI355X only: got a CPU tensor '
                               '(there is no CPU fallback; the CPU restatement lives in oracle/ and is test-only)')


def _stream():
    """the current HIP stream of the current device as a raw handle (the private accessors skip the Stream object and the
    device-index plumbing of torch.cuda.current_stream(): ~1 us instead of ~8 us, 600 times a step: host enqueue time
    72 -> 60 ms per step; the step itself is GPU-bound either way, 2504 vs 2484-2500 graphs/s same-box)"""
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _pair(cls, a, b):
    return (cls * 2)(a, b)


def as_mask3(mask, B, N):
    """(B,N,N,1) additive mask of any float dtype -> contiguous float32 (B,N,N).  The 48 calls of a
    24-layer step pass the same tensor: a real conversion is remembered ON that tensor object (attribute
    `_tgt_mask3`, keyed on its version counter) -- no process-global state, so two models or two threads
    never see each other's masks, and the image dies with the mask."""
    memo = getattr(mask, '_tgt_mask3', None)
    if memo is not None and memo[0] == mask._version and memo[1].shape == (B, N, N):
        return memo[1]
    m = mask.reshape(B, N, N)
    if m.dtype != torch.float32:
        m = m.float()
    m = m.contiguous()
    if m.data_ptr() != mask.data_ptr():           # a real conversion / copy: worth remembering
        mask._tgt_mask3 = (mask._version, m)
    return m


# ---------------------------------------------------------------------------
# triplet attention
# ---------------------------------------------------------------------------
class TripletLayout:
    """Column offsets (elements) inside the fused projection row
    [Q_in|K_in|V_in|Q_out|K_out|V_out|E_in|G_in|E_out|G_out] (head-major Q/K/V)."""

    def __init__(self, C_, H, gated=True, biased=True):
        self.C, self.H, self.D = C_, H, C_ // H
        self.gated, self.biased = gated, biased
        self.q = (0, 3 * C_)
        self.k = (C_, 4 * C_)
        self.v = (2 * C_, 5 * C_)
        nb = (2 if gated else 1) * H if biased else 0
        self.e = (6 * C_, 6 * C_ + nb)
        self.g = (6 * C_ + H, 6 * C_ + nb + H) if gated else (0, 0)
        self.used = 6 * C_ + 2 * nb
        self.width = -(-self.used // 8) * 8           # rows stay 16-byte aligned for every dtype
        self.flags = (_lib.TRI_BIASED if biased else 0) | (_lib.TRI_GATED if gated else 0)


def _tri_args(fused, mask3, out, L, d_out=None, d_fused=None, colsum=None, dropout=(0.0, 0), eg=None, graph_scale=None):
    """eg given: `fused` holds only the Q/K/V channels (rows of 6C) and `eg` the third-arm E/G channels
    (rows of L.used - 6C); d_fused / colsum stay ONE fused row of L.width (ld_dqkv / ld_deg)."""
    B, N = fused.shape[0], fused.shape[1]
    a = _lib.TripletAttentionArgs()
    a.B, a.N, a.H, a.D = B, N, L.H, L.D
    a.dtype, a.flags, a.scale = _DT[fused.dtype], L.flags, float(L.D) ** -0.5
    p = fused.data_ptr()
    split = eg is not None
    wq = 6 * L.C if split else L.width                    # row length of the Q/K/V tensor
    we, eo = (L.used - 6 * L.C, 6 * L.C) if split else (L.width, 0)      # E/G rows, and where they start in a fused row
    a.qkv = _pair(C.c_void_p, p, p)
    a.ld_qkv = _pair(C.c_int64, wq, wq)
    a.q_off, a.k_off, a.v_off = _pair(C.c_int32, *L.q), _pair(C.c_int32, *L.k), _pair(C.c_int32, *L.v)
    pe = eg.data_ptr() if split else p
    a.eg = _pair(C.c_void_p, pe, pe)
    a.ld_eg = _pair(C.c_int64, we, we)
    a.e_off, a.g_off = _pair(C.c_int32, L.e[0] - eo, L.e[1] - eo), _pair(C.c_int32, L.g[0] - eo, L.g[1] - eo)
    a.mask = mask3.data_ptr()
    a.out, a.ld_out = out.data_ptr(), 2 * L.C
    a.o_off = _pair(C.c_int32, 0, L.C)
    a.dropout_p, a.dropout_seed = float(dropout[0]), int(dropout[1]) & 0xFFFFFFFFFFFFFFFF
    if graph_scale is not None:           # per-graph DropPath factor of the branch: graphs at exactly 0 are not computed
        if graph_scale.dtype != torch.float32 or graph_scale.numel() != B or not graph_scale.is_contiguous() or not graph_scale.is_cuda:
            raise RuntimeError('triplet attention: graph_scale must be a contiguous float32 device tensor with one value per graph')
        a.graph_scale = graph_scale.data_ptr()
    if d_out is not None:
        a.d_out = d_out.data_ptr()
        dp = d_fused.data_ptr()
        esz = d_fused.element_size()
        a.d_qkv = _pair(C.c_void_p, dp, dp)
        a.d_eg = _pair(C.c_void_p, dp + eo * esz, dp + eo * esz)
        if split:
            a.ld_dqkv = _pair(C.c_int64, L.width, L.width)
            a.ld_deg = _pair(C.c_int64, L.width, L.width)
        if colsum is not None:            # (B, width) fp32: per-graph column sums of d_fused
            cp = colsum.data_ptr()
            a.d_qkv_colsum = _pair(C.c_void_p, cp, cp)
            if L.biased:
                a.d_eg_colsum = _pair(C.c_void_p, cp + eo * 4, cp + eo * 4)
    return a


# ---- randomness that survives a hipGraph capture (training/graphed.py) --------------------------------------------------
# A captured step replays the kernel ARGUMENTS of the capture: a dropout seed drawn on the host would give every replay the
# same drop pattern, and a vector taken from a pre-drawn pool the same DropPath draw.  In graph-safe mode
#   * host seeds are a fixed function of the call's position inside the step (begin_step() resets the position), and the step
#     itself enters through a 64-bit DEVICE counter the kernels mix into the seed (tgt_set_seed_counter, ABI 29) -- the captured
#     graph increments it once per replay;
#   * DropPath factors / source-dropout masks are drawn by torch where they are needed (torch's generator is capture-aware: the
#     Philox offset of a replay comes from device memory), not from the pools.
# An eager step in this mode computes exactly what a replay computes (tests/test_hip_trainer.py).
_GRAPH_SAFE = [False]
_site = [0]
_seed_counter = [None]            # the registered device counter (kept alive here)


def graph_safe_rng(on):
    _GRAPH_SAFE[0] = bool(on)
    _site[0] = 0
    if on:
        reset_random_pools()


def begin_step():
    """start of a training step in graph-safe mode: the per-call seed positions start over"""
    _site[0] = 0


def set_seed_counter(counter):
    """counter: a 1-element int64 device tensor (or None); see graph_safe_rng"""
    if counter is not None:
        assert counter.is_cuda and counter.dtype == torch.int64 and counter.numel() == 1
    _seed_counter[0] = counter
    _lib.check(_lib.lib().tgt_set_seed_counter(_ptr(counter)), 'tgt_set_seed_counter')


def bump_seed_counter():
    """counter += 1 on the registered device counter (what a captured step does first); a no-op without one"""
    if _seed_counter[0] is not None:
        _seed_counter[0].add_(1)


def _host_seed():
    """a 64-bit dropout seed: from torch's CPU generator (no device sync; torch.manual_seed makes it reproducible), or -- graph-safe
    mode -- the position of this call inside the step, spread over 64 bits"""
    if _GRAPH_SAFE[0]:
        _site[0] += 1
        x = (_site[0] * 0x9E3779B97F4A7C15 + 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
        x ^= x >> 31
        return (x * 0xBF58476D1CE4E5B9) & 0x7FFFFFFFFFFFFFFF
    return int(torch.empty((), dtype=torch.int64).random_().item())


def draw_dropout(p, training):
    """(p, seed) of a counter-based in-kernel dropout: p forced to 0 outside training, the seed
    drawn from torch's CPU generator (no device sync; torch.manual_seed makes it reproducible)"""
    p = float(p) if training else 0.0
    return (p, _host_seed() if p > 0 else 0)


class _TripletAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fused, mask3, L, dropout=(0.0, 0), graph_scale=None):
        _dev(fused, mask3)
        fused = fused.contiguous()
        B, N = fused.shape[0], fused.shape[1]
        assert fused.shape == (B, N, N, L.width), (fused.shape, L.width)
        out = torch.empty(B, N, N, 2 * L.C, dtype=fused.dtype, device=fused.device)
        a = _tri_args(fused, mask3, out, L, dropout=dropout, graph_scale=graph_scale)
        _call('tgt_triplet_attention_fwd', _lib.lib().tgt_triplet_attention_fwd, a)
        ctx.save_for_backward(fused, mask3, out)
        ctx.L, ctx.dropout, ctx.graph_scale = L, dropout, (graph_scale if _TRI_SKIP_BWD else None)
        return out

    @staticmethod
    def backward(ctx, d_out):
        fused, mask3, out = ctx.saved_tensors
        d_out = d_out.contiguous()
        d_fused = torch.empty_like(fused)          # every used column is written by the kernel
        if ctx.L.width > ctx.L.used:
            d_fused[..., ctx.L.used:] = 0
        a = _tri_args(fused, mask3, out, ctx.L, d_out, d_fused, dropout=ctx.dropout, graph_scale=ctx.graph_scale)
        _call('tgt_triplet_attention_bwd', _lib.lib().tgt_triplet_attention_bwd, a)
        return d_fused, None, None, None, None


def triplet_attention(fused, mask3, layout, dropout=(0.0, 0), graph_scale=None):
    """fused: (B,N,N,layout.width) fused projections (head-major Q/K/V), mask3:
    (B,N,N) float32.  Returns Va (B,N,N,2C) with channel = dir*C + h*D + d.
    dropout: (p, seed) of the attention dropout on the gated weights (draw_dropout).
    graph_scale (B,) float32: the DropPath factor of the residual branch the result feeds (drawn by the caller, applied by
    the caller): graphs whose factor is exactly 0 are skipped -- zeros out, zero gradients -- which equals computing them
    and multiplying by that zero (the incoming gradient of such a graph MUST be zero, as it is behind that multiplication).
    Reference arithmetic: lib/tgt/layers/triplet.py:213-246."""
    return _TripletAttention.apply(fused, mask3, layout, dropout, graph_scale)


def _colsum_workspace(B, width, used, device):
    """(B, width) fp32 per-graph column sums written by the backward kernels: a fresh block from the caching
    allocator per call (owned by that call's autograd node -- no shared scratch whose reuse would rest on stream
    order).  The kernels write every used column of every graph; only layouts with row padding (width > used) need
    the zero fill."""
    if width > used:
        return torch.zeros(B, width, dtype=torch.float32, device=device)
    return torch.empty(B, width, dtype=torch.float32, device=device)


_ATEN_PLANE_SUM = False       # settled (+0.3 % for the kernel; tests patch this): ATen's reduction instead of tgt_sum_planes


# ---------------------------------------------------------------------------
# The backward's launch diet (round 4): the closing sums of a layer as ONE launch -- built, measured, OPT-IN.
# A backward layer ends 13 split-M weight gradients and 8 column-sum / LayerNorm partial buffers with a 4-9 us kernel each
# (sum_planes / sum_rows).  Nothing on the step's chain reads their results: they are parameter gradients, read by the Trainer's
# gradient collection.  With TGT_DEFER_SUMS=1, inside a Trainer's backward (ops.trainer_backward) sum_planes / sum_rows only
# REGISTER (partials, result) and return the -- not yet filled -- result tensor; flush_deferred() runs everything registered so
# far as one tgt_sum_many launch per <= 64 sums, followed by the few kernels that read those results (the column permutation of
# lin_O's weight gradient, the scatter of the fused projection's gradient: _after_sums).  The Trainer flushes before it reads
# gradients (FlatState.collect_grads, i.e. also before every bucket's all-reduce) and when the backward ends; a queue flushes
# itself at TGT_DEFER_MAX entries.  One queue per stream: work registered from the node side stream is summed on that stream.
# Bit-identical to the immediate path (tgt_sum_many does per item exactly what tgt_sum_planes does;
# tests/test_hip_trainer.py::test_deferred_closing_sums_change_nothing).
# Why it is off: ~370 launches fewer per step, and the step gets SLOWER -- per-step medians over 30 steps, same box, alternating
# (profiles/r05g_ab_defer_sweep2.txt): 84.15 / 84.17 / 84.19 / 84.12 ms immediate against 84.76 / 84.79 (flush at 8) and 84.90 /
# 84.84 (flush at 16).  The immediate sums read partials the GEMM has just written (L2 / MALL-resident, 4-5 us each, and their
# dispatch gaps largely overlap the neighbouring kernels' tails); the collected launch reads them from HBM with little memory
# parallelism.  The launch COUNT was the wrong target; what the boundary costs is ~1-2 us here, not the 5 us the trace suggests.
# Contract of the deferral (why the queue holds aliases): autograd's AccumulateGrad adopts a gradient only when it is its single
# owner and CLONES it otherwise; a clone taken before the flush copies unfilled memory.  Registered results therefore reach
# autograd as fresh tensor objects (t.detach() where an object is also referenced elsewhere: _take_colsum), weight-shared stacks
# and anything outside a Trainer's backward never defer.
# ---------------------------------------------------------------------------
_DEFER_SUMS = K.defer_sums           # A/B knob (measured slower: see above)
_DEFER_MAX = K.defer_max            # sums per queue before it flushes itself (<= 64: one launch)
_deferred = {}                     # stream handle -> [ [ (part, out), ... ], [post callables] ]
_deferred_stats = [0, 0]           # [sums registered, tgt_sum_many launches]  (tests / diagnostics)


def _deferring():
    return _DEFER_SUMS and _trainer_backward[0] > 0 and not _WGRAD_STREAM


def _queue():
    key = torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
    q = _deferred.get(key)
    if q is None:
        q = _deferred[key] = [[], [], torch.cuda.current_stream()]
    return q


def _run_sums(pairs):
    L = _lib.lib()
    for i in range(0, len(pairs), _lib.SUM_MANY_MAX):
        chunk = pairs[i:i + _lib.SUM_MANY_MAX]
        items = (_lib.SumItem * len(chunk))()
        for it, (part, out) in zip(items, chunk):
            it.src, it.dst, it.planes, it.n = part.data_ptr(), out.data_ptr(), part.shape[0], out.numel()
        _lib.check(L.tgt_sum_many(items, len(chunk), _stream()), 'tgt_sum_many')
        _deferred_stats[1] += 1


def _flush_queue(q):
    pairs, posts = q[0], q[1]
    q[0], q[1] = [], []
    if pairs:
        _run_sums(pairs)
    for fn in posts:
        fn()


def flush_deferred():
    """run every registered closing sum (and what was queued behind them), each queue on the stream it was registered from"""
    if not _deferred:
        return
    cur = torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
    for key, q in list(_deferred.items()):
        if not (q[0] or q[1]):
            continue
        if key == cur:
            _flush_queue(q)
        else:
            with torch.cuda.stream(q[2]):
                _flush_queue(q)


def _after_sums(fn):
    """fn() reads results of sum_planes / sum_rows: now when nothing is pending on this stream, else right behind the sums"""
    if _deferring():
        q = _queue()
        if q[0] or q[1]:
            q[1].append(fn)
            return
    fn()


def _check_planes(part, out):
    if part.dtype != torch.float32 or out.dtype != torch.float32 or not part.is_contiguous() or not out.is_contiguous() or \
            part.shape[1:] != out.shape:
        raise RuntimeError(f'sum_planes: contiguous float32 (P, ...) -> (...) expected, got {tuple(part.shape)} {part.dtype} -> '
                           f'{tuple(out.shape)} {out.dtype}')


def sum_planes(part, out, defer=True):
    """out <- part.sum(0) for contiguous fp32 part (P, ...) and out (...), fixed order (tgt_sum_planes).  Inside a Trainer's
    backward the sum is only registered (see above): `out` is filled by the next flush_deferred() -- callers that read the
    result themselves pass defer=False."""
    _dev(part, out)
    if _ATEN_PLANE_SUM:
        return torch.sum(part, 0, out=out)
    _check_planes(part, out)
    if defer and _deferring():
        q = _queue()
        # (an ALIAS of out keeps its memory: holding the tensor object itself would make autograd's AccumulateGrad see a second
        # owner of the gradient it is handed and CLONE it -- a copy of memory the flush has not filled yet -- instead of adopting it)
        q[0].append((part, out.detach()))
        _deferred_stats[0] += 1
        if len(q[0]) >= _DEFER_MAX:
            _flush_queue(q)
        return out
    _lib.check(_lib.lib().tgt_sum_planes(_ptr(part), part.shape[0], out.numel(), _ptr(out), _stream()), 'tgt_sum_planes')
    return out


def sum_rows(x, defer=True):
    """fp32 (rows, C) -> (C,) column sums in the fixed order of tgt_sum_planes (rows = planes), deferrable like sum_planes."""
    _dev(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2
    return sum_planes(x, torch.empty(x.shape[1], dtype=torch.float32, device=x.device), defer)


class ParamTable:
    """Which source parameter row feeds each row of a fused kernel-order projection
    (tgt_fuse_rows_args.row_src / row_idx); device copies are made once per device."""

    def __init__(self, row_src, row_idx, n_cols, n_src):
        self.row_src = torch.as_tensor(row_src, dtype=torch.int32).contiguous()
        self.row_idx = torch.as_tensor(row_idx, dtype=torch.int32).contiguous()
        self.n_rows, self.n_cols, self.n_src = int(self.row_src.numel()), int(n_cols), int(n_src)
        self._dev = {}

    def on(self, device):
        t = self._dev.get(device)
        if t is None:
            t = self._dev[device] = (self.row_src.to(device), self.row_idx.to(device))
        return t


def _fuse_args(table, ws, bs, fused_w, fused_b):
    a = _lib.FuseRowsArgs()
    a.n_rows, a.n_cols, a.n_src = table.n_rows, table.n_cols, table.n_src
    a.src_dtype, a.fused_dtype = _DT[ws[0].dtype], _DT[fused_w.dtype]
    rs, ri = table.on(fused_w.device)
    a.row_src, a.row_idx = rs.data_ptr(), ri.data_ptr()
    for i, (w, b) in enumerate(zip(ws, bs)):
        assert w.is_contiguous() and b.is_contiguous() and w.dtype == ws[0].dtype and b.dtype == ws[0].dtype
        a.src[i], a.src_bias[i] = w.data_ptr(), b.data_ptr()
    a.fused, a.fused_bias = fused_w.data_ptr(), fused_b.data_ptr()
    return a


def _fuse_params(table, params, cd):
    """(weight, bias) of the fused projection in dtype cd from the module's nn.Linear parameters
    (w0, b0, w1, b1, ...) in ONE launch; reads the 16-bit shadows when the Trainer keeps them."""
    _dev(*params)
    srcs = [_as_dtype_view(p, cd) for p in params]
    if any(t.dtype != srcs[0].dtype for t in srcs):
        srcs = [p.detach() for p in params]
    w = torch.empty(table.n_rows, table.n_cols, dtype=cd, device=params[0].device)
    b = torch.empty(table.n_rows, dtype=cd, device=params[0].device)
    a = _fuse_args(table, srcs[0::2], srcs[1::2], w, b)
    _lib.check(_lib.lib().tgt_fuse_rows(C.byref(a), _stream()), 'tgt_fuse_rows')
    return w, b


def _unfuse_grads(table, params, dW, db):
    """per-parameter gradients (dtype of each parameter) from the fused projection's fp32
    gradients in ONE launch"""
    grads = []
    for p in params:                      # (inside a Trainer's backward: straight into the flat gradient buffer, see _grad_dst)
        dst = _grad_dst(p.data_ptr(), p.shape, p.dtype)
        grads.append(dst if dst is not None else torch.empty_like(p))
    if not (dW.is_contiguous() and db.dtype == dW.dtype and db.is_contiguous()):
        flush_deferred()                  # (a cast / copy reads them: they must not be pending sums)
        dW, db = dW.contiguous(), db.to(dW.dtype).contiguous()
    a = _fuse_args(table, grads[0::2], grads[1::2], dW, db)
    keep = [t.detach() for t in (dW, db, *grads)]       # aliases: the memory stays, the gradient objects keep ONE owner (see sum_planes)

    def run():
        _lib.check(_lib.lib().tgt_unfuse_rows(C.byref(a), _stream()), 'tgt_unfuse_rows')
        keep.clear()
    _after_sums(run)
    return grads


_SPLIT_MIN_ROWS = 65536          # below this the single GEMM is as good (tests lower it)


def _split_projection_ok(x, L):
    """Q/K/V and E/G projected by two GEMMs into two tensors (big inputs, biased layouts whose E/G row
    is 16-byte aligned and unpadded); TGT_TRI_SPLIT=0: one fused GEMM (A/B knob)"""
    nb = L.used - 6 * L.C
    return (_TRI_SPLIT and L.biased and L.width == L.used and nb > 0 and nb % 8 == 0 and
            x.numel() // L.C >= _SPLIT_MIN_ROWS)


_slow_path_said = set()


def _slow_path_notice(key, msg):
    """Say ONCE per process (and reason) that a BASELINE-sized call takes a kernel family that is 2-3x off the hot one, instead of
    doing so silently.  Small problems (tests) never reach the callers' size thresholds."""
    if key not in _slow_path_said:
        _slow_path_said.add(key)
        import warnings
        warnings.warn('tgt_amd: ' + msg, RuntimeWarning, stacklevel=3)


def _proj_fused_ok(x, N, L, cd):
    """the projection-fused forward kernel (tgt_triplet_attention_proj_fwd, wave roles: DESIGN.md section 4, profiles/HISTORY_rounds_1-4.md 4.1a): Q/K/V are
    projected inside the attention kernel (still written once, for the backward); TGT_TRI_PROJ=0 is the A/B knob"""
    return (_TRI_PROJ and N <= 32 and L.D == 16 and L.H % 8 == 0 and L.width == L.used and
            cd in (torch.bfloat16, torch.float16) and L.C == 256 and x.numel() // L.C >= _SPLIT_MIN_ROWS)


# A/B knob (round 5): keep the node side stream's backward chain of a layer off the HBM-bound triplet backward
# kernel of that layer.  The host enqueues a layer's node-FFN backward AFTER the layer's triplet backward (lower sequence numbers),
# but on the GPU the chain starts as soon as dh arrives from the layer above -- i.e. under the edge FFN's backward and, with its
# tail, under tri_att_bwd2 (0.40 ms alone, 0.45-0.47 inside the step).  Gated, the side stream waits for an event recorded behind
# the triplet backward kernel (1) or behind the projection's data-gradient GEMM that follows it (2), so the chain runs under the
# projection's GEMMs instead.  Consumed by the first Linear backward that runs on the side stream afterwards.
_GATE_NODE_BWD = K.gate_node_bwd
_tri_gate = {}                    # device index -> event recorded on the step's stream behind the latest triplet backward


def _gate_record(dev):
    if _GATE_NODE_BWD and side_stream.enabled and side_stream._owners > 0 and _trainer_backward[0] > 0:
        ev = torch.cuda.Event()
        ev.record()
        _tri_gate[dev.index] = ev


def _gate_wait(t):
    if _tri_gate and t.is_cuda:
        side = _side_streams.get(t.device)
        if side is not None and torch._C._cuda_getCurrentRawStream(t.device.index) == side.cuda_stream:
            ev = _tri_gate.pop(t.device.index, None)
            if ev is not None:
                side.wait_event(ev)


class _ProjectedTripletAttention(torch.autograd.Function):
    """fused projection GEMM + triplet attention core as ONE autograd node, so that the backward
    kernel can hand the projection its bias gradient (column sums of d_fused, accumulated while
    the gradient rows are written) instead of a separate 0.84 GB reduction pass.
    wb: the pre-fused (weight, bias), or -- with a ParamTable -- the module's nn.Linear
    parameters (w0, b0, w1, b1, ...), fused/unfused here with one launch each way."""

    @staticmethod
    def forward(ctx, x, mask3, L, cd, table, dropout, graph_scale, *wb):
        _dev(x, mask3)
        B, N = x.shape[0], x.shape[1]
        weight, bias = wb if table is None else _fuse_params(table, wb, cd)
        out = torch.empty(B, N, N, 2 * L.C, dtype=cd, device=x.device)
        eg = None
        proj_skip = None
        if x.numel() // L.C >= _SPLIT_MIN_ROWS and not (dropout[0] == 0 and _proj_fused_ok(x, N, L, cd)) and _TRI_PROJ:
            why = ('N > 32' if N > 32 else 'attention dropout > 0' if dropout[0] else 'head dim != 16' if L.D != 16 else
                   'heads not a multiple of 8' if L.H % 8 else 'edge width != 256' if L.C != 256 else
                   'fp32' if cd not in (torch.bfloat16, torch.float16) else 'padded / unbiased layout')
            _slow_path_notice(('tri_proj', why), f'triplet attention: the projection-fused forward / round-4 backward do not take this shape ({why}); '
                              'running the projection as library GEMMs + the general attention kernels (N in 33..64: 16-wide tiles at 0.39-0.45 of '
                              'HBM instead of 0.52; see DESIGN.md section 4)')
        if dropout[0] == 0 and _proj_fused_ok(x, N, L, cd):
            # Q/K/V projected inside the attention kernel (it still writes them once, for the
            # backward); only the narrow E/G third-arm projection stays a library GEMM, written
            # straight into its columns of the fused row
            x2 = x.reshape(-1, L.C)
            x2 = (x2 if x2.dtype == cd else x2.to(cd)).contiguous()
            w, b = _as_dtype(weight, cd).contiguous(), _as_dtype(bias, cd).contiguous()
            # the Q/K/V rows (written by the kernel, for the backward) and the narrow third-arm E/G projection as two tensors,
            # as in the split-GEMM path below
            fused = torch.empty(B, N, N, 6 * L.C, dtype=cd, device=x.device)
            we, be = w[6 * L.C:L.used], b[6 * L.C:L.used]
            if not L.biased:
                eg = None                      # (axial: no third arm)
            elif we.shape[0] <= 128 and _edge_kernel_ok(x2, we.shape[0], cd):
                eg = edge_linear_raw(x2, we, be.contiguous()).view(B, N, N, L.used - 6 * L.C)
            else:
                eg = torch.addmm(be, x2, we.t()).view(B, N, N, L.used - 6 * L.C)
            a = _tri_args(fused, mask3, out, L, eg=eg, graph_scale=graph_scale)
            s0, s1 = _prof_begin('tgt_triplet_attention_proj_fwd')
            _lib.check(_lib.lib().tgt_triplet_attention_proj_fwd(C.byref(a), _ptr(x2), L.C, _ptr(w), _ptr(b), _stream()),
                       'tgt_triplet_attention_proj_fwd')
            _prof_end('tgt_triplet_attention_proj_fwd', s0, s1)
            proj_skip = graph_scale        # (dropped graphs have NO Q/K/V rows: the backward must skip them too)
        elif _split_projection_ok(x, L):
            # two GEMMs: Q/K/V (6C = 1536 channels: six full 256-wide tile columns, 294 us) and the
            # narrow E/G (39 us) instead of one ragged 1600-wide GEMM (376 us; tools/gemm_probe.py);
            # the backward still writes ONE fused gradient row (ld_dqkv / ld_deg)
            x2 = x.reshape(-1, L.C)
            x2 = x2 if x2.dtype == cd else x2.to(cd)
            w, b = _as_dtype(weight, cd), _as_dtype(bias, cd)
            fused = torch.addmm(b[:6 * L.C], x2, w[:6 * L.C].t()).view(B, N, N, 6 * L.C)
            we, be = w[6 * L.C:L.used], b[6 * L.C:L.used]
            if we.shape[0] <= 128 and we.is_contiguous() and _edge_kernel_ok(x2, we.shape[0], cd):
                # the narrow third-arm projection on the weight-resident slice kernel (29 vs 37 us against the tuned library GEMM)
                eg = edge_linear_raw(x2, we, be.contiguous()).view(B, N, N, L.used - 6 * L.C)
            else:
                eg = torch.addmm(be, x2, we.t()).view(B, N, N, L.used - 6 * L.C)
            a = _tri_args(fused, mask3, out, L, dropout=dropout, eg=eg, graph_scale=graph_scale)
            _call('tgt_triplet_attention_fwd', _lib.lib().tgt_triplet_attention_fwd, a)
        else:
            x2, w, fused = _linear_forward(x, weight, bias, cd)
            a = _tri_args(fused, mask3, out, L, dropout=dropout, graph_scale=graph_scale)
            _call('tgt_triplet_attention_fwd', _lib.lib().tgt_triplet_attention_fwd, a)
        ctx.save_for_backward(x2, w, fused, mask3, out, eg if eg is not None else fused.new_empty(0),
                              *(wb if table is not None else ()))
        ctx.L, ctx.table, ctx.dropout = L, table, dropout
        ctx.graph_scale = proj_skip if proj_skip is not None else (graph_scale if _TRI_SKIP_BWD else None)
        ctx.meta = (x.shape, x.dtype, weight.dtype, bias.dtype)
        return out

    @staticmethod
    def backward(ctx, d_out):
        x2, w, fused, mask3, out, eg = ctx.saved_tensors[:6]
        params = ctx.saved_tensors[6:]
        eg = eg if eg.numel() else None
        L, table = ctx.L, ctx.table
        xs, xdt, wdt, bdt = ctx.meta
        d_out = d_out.contiguous()
        d_fused = torch.empty(*fused.shape[:3], L.width, dtype=fused.dtype, device=fused.device)   # every used column is written
        if L.width > L.used:
            d_fused[..., L.used:] = 0
        colsum = _colsum_workspace(fused.shape[0], L.width, L.used, fused.device)
        a = _tri_args(fused, mask3, out, L, d_out, d_fused, colsum, dropout=ctx.dropout, eg=eg, graph_scale=ctx.graph_scale)
        _call('tgt_triplet_attention_bwd', _lib.lib().tgt_triplet_attention_bwd, a)
        if _GATE_NODE_BWD == 1:
            _gate_record(d_fused.device)
        need_p = any(ctx.needs_input_grad[7:])
        d2 = d_fused.view(-1, L.width)
        if eg is not None and need_p:
            ws0 = _terminal_fork(d2.shape[0], colsum)
            with _on_stream(ws0):
                db = sum_rows(colsum)
        else:
            db = sum_rows(colsum) if need_p else None
        if eg is not None and need_p:
            # weight gradient as two batched GEMMs as well: the 1536-row block has no ragged tile row
            # (213 us with 32 chunks) and the 64-row E/G block is cheap (35 us), against 303 us for the
            # 1600-row product (tools/wgrad_chunk_probe.py); dx stays one GEMM over the fused row
            ws = _wgrad_fork(d2, x2, db)
            dx, _, _ = _linear_backward(x2, w, d2, xs, xdt, torch.float32, None, ctx.needs_input_grad[0], False, False)
            if _GATE_NODE_BWD == 2:
                _gate_record(d_fused.device)
            with _on_stream(ws):
                dw = torch.empty(L.width, L.C, dtype=torch.float32, device=d2.device)
                _wgrad_into(dw[:6 * L.C], d2[:, :6 * L.C], x2, 32)
                _wgrad_into(dw[6 * L.C:], d2[:, 6 * L.C:], x2, 128)
        else:
            ws = None
            dx, dw, _ = _linear_backward(x2, w, d2, xs, xdt, torch.float32, None,
                                         ctx.needs_input_grad[0], need_p, False)
        if not need_p:
            return (dx, None, None, None, None, None, None) + (None,) * (len(ctx.needs_input_grad) - 7)
        if table is None:
            return dx, None, None, None, None, None, None, _param_grad(dw, wdt), _param_grad(db, bdt)
        with _on_stream(ws):           # (the parameter gradients leave on the stream their weight gradient was computed on)
            grads = _unfuse_grads(table, params, dw, db)
        return (dx, None, None, None, None, None, None, *grads)


def projected_triplet_attention(x, weight, bias, mask3, layout, table=None, dropout=(0.0, 0), graph_scale=None):
    """triplet_attention(linear(x, weight, bias), mask3, layout) with the bias gradient of the
    projection produced inside the backward kernel.  weight/bias: the fused (layout.width, C)
    projection in kernel order (see TripletLayout) -- or, with a ParamTable, `weight` is the
    tuple of the module's nn.Linear parameters (w0, b0, w1, b1, ...) and bias is None."""
    if not _TRI_COLSUM and table is None:      # A/B knob: separate bias-gradient pass
        return triplet_attention(linear(x, weight, bias), mask3, layout, dropout, graph_scale)
    cd = torch.get_autocast_dtype('cuda') if (x.is_cuda and torch.is_autocast_enabled('cuda')) else x.dtype
    wb = (weight, bias) if table is None else tuple(weight)
    return _ProjectedTripletAttention.apply(x, mask3, layout, cd, table, dropout, graph_scale, *wb)


# ---------------------------------------------------------------------------
# triplet aggregate
# ---------------------------------------------------------------------------
class AggregateLayout:
    """[V_in|V_out|E_in|G_in|E_out|G_out] gated, [V_in|V_out|E_in|E_out] ungated."""

    def __init__(self, C_, H, gated=True):
        self.C, self.H, self.D, self.gated = C_, H, C_ // H, gated
        self.v = (0, C_)
        if gated:
            self.e = (2 * C_, 2 * C_ + 2 * H)
            self.g = (2 * C_ + H, 2 * C_ + 3 * H)
            self.used = 2 * C_ + 4 * H
            self.flags = _lib.TRI_BIASED | _lib.TRI_GATED
        else:
            self.e = (2 * C_, 2 * C_ + H)
            self.g = (0, 0)
            self.used = 2 * C_ + 2 * H
            self.flags = _lib.TRI_BIASED | _lib.TRI_MASK_OUT
        self.width = -(-self.used // 8) * 8


def _agg_args(fused, mask3, out, L, d_out=None, d_fused=None, dropout=(0.0, 0)):
    B, N = fused.shape[0], fused.shape[1]
    a = _lib.TripletAggregateArgs()
    a.B, a.N, a.H, a.D = B, N, L.H, L.D
    a.dtype, a.flags = _DT[fused.dtype], L.flags
    p = fused.data_ptr()
    a.v = _pair(C.c_void_p, p, p)
    a.ld_v = _pair(C.c_int64, L.width, L.width)
    a.v_off = _pair(C.c_int32, *L.v)
    a.eg = _pair(C.c_void_p, p, p)
    a.ld_eg = _pair(C.c_int64, L.width, L.width)
    a.e_off, a.g_off = _pair(C.c_int32, *L.e), _pair(C.c_int32, *L.g)
    a.mask = mask3.data_ptr()
    a.out, a.ld_out = out.data_ptr(), 2 * L.C
    a.o_off = _pair(C.c_int32, 0, L.C)
    a.dropout_p, a.dropout_seed = float(dropout[0]), int(dropout[1]) & 0xFFFFFFFFFFFFFFFF
    if d_out is not None:
        a.d_out = d_out.data_ptr()
        dp = d_fused.data_ptr()
        a.d_v = _pair(C.c_void_p, dp, dp)
        a.d_eg = _pair(C.c_void_p, dp, dp)
    return a


class _TripletAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fused, mask3, L, dropout=(0.0, 0)):
        _dev(fused, mask3)
        fused = fused.contiguous()
        B, N = fused.shape[0], fused.shape[1]
        assert fused.shape == (B, N, N, L.width), (fused.shape, L.width)
        out = torch.empty(B, N, N, 2 * L.C, dtype=fused.dtype, device=fused.device)
        a = _agg_args(fused, mask3, out, L, dropout=dropout)
        _call('tgt_triplet_aggregate_fwd', _lib.lib().tgt_triplet_aggregate_fwd, a)
        ctx.save_for_backward(fused, mask3, out)
        ctx.L, ctx.dropout = L, dropout
        return out

    @staticmethod
    def backward(ctx, d_out):
        fused, mask3, out = ctx.saved_tensors
        d_out = d_out.contiguous()
        d_fused = torch.empty_like(fused)
        if ctx.L.width > ctx.L.used:
            d_fused[..., ctx.L.used:] = 0
        a = _agg_args(fused, mask3, out, ctx.L, d_out, d_fused, dropout=ctx.dropout)
        _call('tgt_triplet_aggregate_bwd', _lib.lib().tgt_triplet_aggregate_bwd, a)
        return d_fused, None, None, None


def triplet_aggregate(fused, mask3, layout, dropout=(0.0, 0)):
    """Reference arithmetic: lib/tgt/layers/triplet.py:56-70 / :107-123; dropout as triplet_attention."""
    return _TripletAggregate.apply(fused, mask3, layout, dropout)


# ---------------------------------------------------------------------------
# triangular update
# ---------------------------------------------------------------------------
class _TriangularUpdate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, e4, v4, mask3, H):
        _dev(e4, v4, mask3)
        e4, v4 = e4.contiguous(), v4.contiguous()
        B, N = e4.shape[0], e4.shape[1]
        out = torch.empty(B, N, N, 2 * H, dtype=e4.dtype, device=e4.device)
        _lib.check(_lib.lib().tgt_triangular_update_fwd(_ptr(e4), _ptr(v4), _ptr(mask3), _ptr(out), B, N, H,
                                                        _DT[e4.dtype], _stream()), 'tgt_triangular_update_fwd')
        ctx.save_for_backward(e4, v4, mask3)
        ctx.H = H
        return out

    @staticmethod
    def backward(ctx, d_out):
        e4, v4, mask3 = ctx.saved_tensors
        d_out = d_out.contiguous()
        B, N = e4.shape[0], e4.shape[1]
        d_e4, d_v4 = torch.empty_like(e4), torch.empty_like(v4)
        _lib.check(_lib.lib().tgt_triangular_update_bwd(_ptr(e4), _ptr(v4), _ptr(mask3), _ptr(d_out), _ptr(d_e4),
                                                        _ptr(d_v4), B, N, ctx.H, _DT[e4.dtype], _stream()),
                   'tgt_triangular_update_bwd')
        return d_e4, d_v4, None, None


def triangular_update(e4, v4, mask3, num_heads):
    """e4, v4: (B,N,N,4H) lin_E / lin_V outputs -> (B,N,N,2H).  Reference lib/tgt/layers/triplet.py:156-172."""
    if v4.dtype != e4.dtype:
        v4 = v4.to(e4.dtype)
    return _TriangularUpdate.apply(e4, v4, mask3, num_heads)


# ---------------------------------------------------------------------------
# node attention (EGT_Attention core) and EdgeUpdate logits
# ---------------------------------------------------------------------------
def _node_args(qkv, eg, mask3, H, scale_degree, logits_only):
    B, N = qkv.shape[0], qkv.shape[1]
    W = qkv.shape[2] // (2 if logits_only else 3)
    a = _lib.NodeAttentionArgs()
    a.B, a.N, a.H, a.D = B, N, H, W // H
    a.dtype, a.scale_degree, a.logits_only = _DT[qkv.dtype], int(bool(scale_degree)), int(bool(logits_only))
    a.scale = float(W // H) ** -0.5
    a.qkv, a.ld_qkv = qkv.data_ptr(), qkv.shape[2]
    a.q_off, a.k_off, a.v_off = 0, W, (0 if logits_only else 2 * W)
    a.eg, a.ld_eg = eg.data_ptr(), eg.shape[3]
    a.e_off, a.g_off = 0, (0 if logits_only else H)
    a.mask = None if mask3 is None else mask3.data_ptr()
    return a, W


class _NodeAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, eg, mask3, H, scale_degree, want_edges, hhat_scale=None):
        _dev(qkv, eg, mask3, hhat_scale)
        qkv, eg = qkv.contiguous(), eg.contiguous()
        if eg.dtype != qkv.dtype:
            eg = eg.to(qkv.dtype)
        B, N = qkv.shape[0], qkv.shape[1]
        a, W = _node_args(qkv, eg, mask3, H, scale_degree, False)
        if B * N * N >= 65536 and (N > 64 or qkv.dtype == torch.float32 or H % 8 or (W // H) not in (8, 12, 16)):
            why = 'N > 64' if N > 64 else 'fp32' if qkv.dtype == torch.float32 else 'heads not a multiple of 8' if H % 8 else 'head dim not in {8, 12, 16}'
            # (N > 64 with H a multiple of 32, 16-bit: the FORWARD still runs the key-blocked matrix-core kernel, N <= 1024; only the backward
            # is lane-per-head there)
            which = 'the backward runs' if (why == 'N > 64' and H % 32 == 0 and N <= 1024 and (W // H) in (8, 12, 16)) else 'running'
            _slow_path_notice(('node_mfma', why), f'node attention: the matrix-core kernels do not take this shape ({why}); {which} the lane-per-head '
                              'kernels (0.15-0.20 of HBM at N = 48 against 0.28-0.39; DESIGN.md section 4)')
        vatt = torch.empty(B, N, W, dtype=qkv.dtype, device=qkv.device)
        hhat = torch.empty(B, N, N, H, dtype=qkv.dtype, device=qkv.device) if want_edges else None
        lse = torch.empty(B, N, H, dtype=torch.float32, device=qkv.device)
        gsum = torch.empty(B, N, H, dtype=torch.float32, device=qkv.device)
        a.vatt, a.hhat, a.lse, a.gsum = vatt.data_ptr(), _ptr(hhat), lse.data_ptr(), gsum.data_ptr()
        a.hhat_scale = _ptr(hhat_scale)
        _call('tgt_node_attention_fwd', _lib.lib().tgt_node_attention_fwd, a)
        ctx.save_for_backward(qkv, eg, mask3, lse, gsum, vatt, hhat_scale)
        ctx.cfg = (H, scale_degree, want_edges)
        if want_edges:
            return vatt, hhat
        return vatt, None

    @staticmethod
    def backward(ctx, d_vatt, d_hhat):
        qkv, eg, mask3, lse, gsum, vatt, hhat_scale = ctx.saved_tensors
        H, scale_degree, want_edges = ctx.cfg
        a, W = _node_args(qkv, eg, mask3, H, scale_degree, False)
        d_vatt = torch.zeros_like(qkv[..., :W]).contiguous() if d_vatt is None else d_vatt.contiguous()
        if d_hhat is not None:
            d_hhat = d_hhat.contiguous()
        d_qkv, d_eg = torch.empty_like(qkv), torch.empty_like(eg)
        a.lse, a.gsum, a.vatt = lse.data_ptr(), gsum.data_ptr(), vatt.data_ptr()
        a.d_vatt, a.d_hhat, a.d_qkv, a.d_eg = d_vatt.data_ptr(), _ptr(d_hhat), d_qkv.data_ptr(), d_eg.data_ptr()
        a.hhat_scale = _ptr(hhat_scale)
        _call('tgt_node_attention_bwd', _lib.lib().tgt_node_attention_bwd, a)
        return d_qkv, d_eg, None, None, None, None, None


def node_attention(qkv, eg, mask3, num_heads, scale_degree=True, want_edges=True, hhat_scale=None):
    """qkv (B,N,3W) and eg (B,N,N,2H) in the reference's head-minor layout (channel = d*H + h);
    returns V_att (B,N,W) and H_hat (B,N,N,H) (or None).
    hhat_scale (B,) float32: H_hat is returned multiplied by hhat_scale[b] (the DropPath factor of the edge branch it
    feeds, folded in: linear_residual_layer_norm(prescaled=True)).
    Reference arithmetic: lib/tgt/layers/layers.py:62-77."""
    return _NodeAttention.apply(qkv, eg, mask3, num_heads, scale_degree, want_edges, hhat_scale)


class _EdgeLogits(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qk, e_bias, H):
        _dev(qk, e_bias)
        qk, e_bias = qk.contiguous(), e_bias.contiguous()
        if e_bias.dtype != qk.dtype:
            e_bias = e_bias.to(qk.dtype)
        B, N = qk.shape[0], qk.shape[1]
        a, W = _node_args(qk, e_bias, None, H, False, True)
        hhat = torch.empty(B, N, N, H, dtype=qk.dtype, device=qk.device)
        a.hhat = hhat.data_ptr()
        _call('tgt_node_attention_fwd(logits)', _lib.lib().tgt_node_attention_fwd, a)
        ctx.save_for_backward(qk, e_bias)
        ctx.H = H
        return hhat

    @staticmethod
    def backward(ctx, d_hhat):
        qk, e_bias = ctx.saved_tensors
        a, W = _node_args(qk, e_bias, None, ctx.H, False, True)
        d_hhat = d_hhat.contiguous()
        d_qk, d_e = torch.empty_like(qk), torch.empty_like(e_bias)
        a.d_hhat, a.d_qkv, a.d_eg = d_hhat.data_ptr(), d_qk.data_ptr(), d_e.data_ptr()
        _call('tgt_node_attention_bwd(logits)', _lib.lib().tgt_node_attention_bwd, a)
        return d_qk, d_e, None


def edge_logits(qk, e_bias, num_heads):
    """EdgeUpdate logits: s*QK^T + E.  Reference lib/tgt/layers/layers.py:120-124."""
    return _EdgeLogits.apply(qk, e_bias, num_heads)


# ---------------------------------------------------------------------------
# flat Adam
# ---------------------------------------------------------------------------
def adam_step_(param, grad, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8,
               weight_decay=0.0, grad_scale=1.0, shadow=None, clip_value=0.0, ctl=None):
    """In-place Adam on flat float32 buffers (replaces apex FusedAdam,
    reference lib/training/training.py:159-171).  clip_value > 0: clip_grad_value_ folded in
    (training.py:455-460).  ctl: the device control block of grad_scaler_step_ (skip flag, gradient
    multiplier, applied-step count); `step` / `grad_scale` are then ignored."""
    _dev(param, grad, exp_avg, exp_avg_sq, ctl)
    for t in (param, grad, exp_avg, exp_avg_sq):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == param.numel()
    if shadow is not None:
        assert shadow.is_cuda and shadow.numel() == param.numel() and shadow.dtype in (torch.bfloat16, torch.float16)
    if ctl is not None:
        assert ctl.dtype == torch.float32 and ctl.is_contiguous() and ctl.numel() >= CTL_SIZE
    _lib.check(_lib.lib().tgt_adam_step(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq),
                                        param.numel(), lr, betas[0], betas[1], eps, weight_decay,
                                        int(step), grad_scale, float(clip_value or 0.0), _ptr(ctl), _ptr(shadow),
                                        0 if shadow is None else _DT[shadow.dtype], _stream()), 'tgt_adam_step')


# layout of the optimizer control block (include/tgt_hip.h TGT_CTL_*)
CTL_SCALE, CTL_TRACKER, CTL_FOUND_INF, CTL_STEPS, CTL_MULT, CTL_COEF, CTL_NORM, CTL_SKIPPED = range(8)
CTL_LOSS, CTL_SAMPLES, CTL_NAN, CTL_LOSS_LO, CTL_PAIR, CTL_SAMPLES_LO, CTL_LR, CTL_SIZE = 8, 9, 10, 11, 12, 14, 15, 16


def grad_scaler_step_(grad, ctl, world=1, clip_value=0.0, clip_norm=0.0, dynamic=False, growth_factor=2.0,
                      backoff_factor=0.5, growth_interval=2000):
    """One pass over the flat (all-reduced, still loss-scaled) gradient + a one-thread decision kernel:
    GradScaler's found_inf / skip / scale update and clip_grad_norm_'s coefficient, all left on the
    device in `ctl` for adam_step_ (reference training.py:451-469 does this with a host sync)."""
    _dev(grad, ctl)
    assert grad.dtype == torch.float32 and grad.is_contiguous() and ctl.dtype == torch.float32 and ctl.numel() >= CTL_SIZE
    L = _lib.lib()
    partial = torch.empty(L.tgt_grad_stats_parts(), dtype=torch.float32, device=grad.device)
    _lib.check(L.tgt_grad_scaler_step(_ptr(grad), grad.numel(), _ptr(ctl), _ptr(partial), int(world), float(clip_value or 0.0),
                                      float(clip_norm or 0.0), int(bool(dynamic)), float(growth_factor), float(backoff_factor),
                                      int(growth_interval), _stream()), 'tgt_grad_scaler_step')


def loss_accumulate_(loss, samples, ctl, mixed, mode=3):
    """update_losses on the device (reference tgt_training.py:141-171): mode 1 writes
    (loss*samples, samples) to ctl[12:14] (all-reduce those between the halves), mode 2 accumulates
    them into ctl[8:10] with the reference's NaN-skipping rule, mode 3 does both."""
    _dev(ctl, loss)
    if loss is not None:
        assert loss.numel() == 1 and loss.dtype in (torch.float32, torch.float64)
    _lib.check(_lib.lib().tgt_loss_accumulate(_ptr(loss), int(loss is not None and loss.dtype == torch.float64), float(samples),
                                              _ptr(ctl), int(bool(mixed)), int(mode), _stream()), 'tgt_loss_accumulate')


# ---------------------------------------------------------------------------
# LayerNorm (reads x in its storage dtype, writes the consumer's dtype)
# ---------------------------------------------------------------------------
class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_dtype):
        _dev(x, weight, bias)
        x = x.contiguous()
        C_ = x.shape[-1]
        rows = x.numel() // C_
        w = weight.detach().float().contiguous()
        b = bias.detach().float().contiguous()
        y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        L = _lib.lib()
        s, e = _prof_begin()
        _lib.check(L.tgt_layer_norm_fwd(_ptr(x), _DT[x.dtype], _ptr(w), _ptr(b), _ptr(y), _DT[out_dtype],
                                        _ptr(mean), _ptr(rstd), rows, C_, float(eps), _stream()), 'tgt_layer_norm_fwd')
        _prof_end('tgt_layer_norm_fwd', s, e)
        ctx.save_for_backward(x, w, mean, rstd)
        ctx.wdtype = weight.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, rstd = ctx.saved_tensors
        dy = dy.contiguous()
        C_ = x.shape[-1]
        rows = x.numel() // C_
        L = _lib.lib()
        dx = torch.empty_like(x)
        # separate tensors (not two views of one): autograd can then adopt them as .grad without a copy
        dgb = (torch.empty(C_, dtype=torch.float32, device=x.device), torch.empty(C_, dtype=torch.float32, device=x.device))
        partial = torch.empty(L.tgt_layer_norm_parts() * 2 * C_, dtype=torch.float32, device=x.device)
        s, e = _prof_begin()
        _lib.check(L.tgt_layer_norm_bwd(_ptr(dy), _DT[dy.dtype], _ptr(x), _DT[x.dtype], _ptr(w), _ptr(mean), _ptr(rstd),
                                        _ptr(dx), _DT[dx.dtype], _ptr(dgb[0]), _ptr(dgb[1]), _ptr(partial),
                                        rows, C_, _stream()), 'tgt_layer_norm_bwd')
        _prof_end('tgt_layer_norm_bwd', s, e)
        return dx, dgb[0].to(ctx.wdtype), dgb[1].to(ctx.wdtype), None, None


def _prof_begin(name=None):
    if _PROFILE is None or not _timed(name):
        return None, None
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    return s, e


def _prof_end(name, s, e):
    if s is not None:
        e.record()
        _PROFILE.setdefault(name, []).append((s, e))


def layer_norm(x, weight, bias, eps=1e-5, out_dtype=None):
    """LayerNorm over the last axis.  out_dtype defaults to the autocast dtype when
    autocast is on (what the consuming GEMM would cast to anyway), else x.dtype."""
    if out_dtype is None:
        out_dtype = torch.get_autocast_dtype('cuda') if torch.is_autocast_enabled('cuda') else x.dtype
    return _LayerNorm.apply(x, weight, bias, eps, out_dtype)


# ---------------------------------------------------------------------------
# glue that is a single device op each
# ---------------------------------------------------------------------------
class _Permute(torch.autograd.Function):
    """index_select along `dim` by a PERMUTATION; the backward is the gather by the inverse
    permutation (one kernel) instead of the sort-based index_put of advanced indexing."""

    @staticmethod
    def forward(ctx, w, perm, inv, dim):
        ctx.save_for_backward(inv)
        ctx.dim = dim
        return w.index_select(dim, perm)

    @staticmethod
    def backward(ctx, g):
        inv, = ctx.saved_tensors
        return g.index_select(ctx.dim, inv), None, None, None


def permute(w, perm, inv, dim=0):
    return _Permute.apply(w, perm, inv, dim)


class _ScalePool:
    """Per-sample DropPath factors drawn 64 vectors at a time (one Bernoulli launch per refill
    instead of one per residual branch)."""

    def __init__(self):
        self.buf, self.cur, self.key = None, 0, None

    def _take(self, B, keep, device):
        key = (B, keep, device)
        if self.key != key or self.buf is None or self.cur >= self.buf.shape[0]:
            self.buf = torch.empty(64, B, dtype=torch.float32, device=device).bernoulli_(keep).div_(keep)
            self.cur, self.key = 0, key
        out = self.buf[self.cur]
        self.cur += 1
        return out

    _pools = {}

    @classmethod
    def take(cls, B, keep, device):
        """one pool per stream (a refill and the reads of its rows stay on one stream) AND per keep probability: the DropPath
        rate ramps over the layers, and a pool keyed by the stream alone was refilled by every layer (two launches each)"""
        if _GRAPH_SAFE[0]:
            return torch.empty(B, dtype=torch.float32, device=device).bernoulli_(keep).div_(keep)
        sid = torch.cuda.current_stream(device).cuda_stream if device.type == 'cuda' else 0
        pool = cls._pools.get((sid, B, keep))
        if pool is None:
            pool = cls._pools[(sid, B, keep)] = cls()
        return pool._take(B, keep, device)


_scale_pool = _ScalePool


def scaled_add_(x, residual, scale):
    """residual + x*scale with a per-sample scale (B,) float32 or None (then in place on x)"""
    if scale is not None:
        sc = scale.view([x.size(0)] + [1] * (x.ndim - 1))
        return torch.addcmul(residual.to(x.dtype) if residual.dtype != x.dtype else residual, x, sc.to(x.dtype))
    return x.add_(residual)


class _DropMaskPool:
    """Source-dropout masks (B,1,N) = Bernoulli(p)*finfo.min drawn 32 layers at a time per stream."""
    _pools = {}

    @classmethod
    def take(cls, B, N, p, fill, device):
        if _GRAPH_SAFE[0]:
            return torch.empty(B, 1, N, dtype=torch.float32, device=device).bernoulli_(p).mul_(fill)
        sid = torch.cuda.current_stream(device).cuda_stream if device.type == 'cuda' else 0
        key = (sid, B, N, p, fill, device)
        st = cls._pools.get(key)
        if st is None or st[1] >= st[0].shape[0]:
            st = cls._pools[key] = [torch.empty(32, B, 1, N, dtype=torch.float32, device=device).bernoulli_(p).mul_(fill), 0]
        out = st[0][st[1]]
        st[1] += 1
        return out


def reset_random_pools():
    """forget the pre-drawn DropPath / source-dropout vectors: call after torch.manual_seed() when a
    run has to be reproducible from that seed (the pools otherwise carry draws over)"""
    _ScalePool._pools.clear()
    _DropMaskPool._pools.clear()


def source_drop_mask(B, N, p, fill, device):
    """additive per-key-node mask of EGT_Attention's source dropout (reference
    lib/tgt/layers/layers.py:55-59): (B,1,N) float32, `fill` (finfo.min) where the key is dropped"""
    return _DropMaskPool.take(B, N, float(p), float(fill), device)


def drop_path_add_(x, residual, drop_prob, training):
    """residual + DropPath(x)  (reference lib/tgt/layers/layers.py:169-174 followed by
    the in-place add_ of :270-290) in ONE pass over the tensors."""
    if drop_prob > 0 and training:
        scale = _scale_pool.take(x.size(0), 1.0 - drop_prob, x.device).view([x.size(0)] + [1] * (x.ndim - 1))
        return torch.addcmul(residual.to(x.dtype) if residual.dtype != x.dtype else residual, x, scale.to(x.dtype))
    return x.add_(residual)


class _GeluDropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed, sample_scale):
        _dev(x, sample_scale)
        x = x.contiguous()
        y = torch.empty_like(x)
        eps_ = x.numel() // x.shape[0] if sample_scale is not None else 0
        s, e = _prof_begin()
        _lib.check(_lib.lib().tgt_gelu_dropout_scaled_fwd(_ptr(x), _ptr(y), x.numel(), _DT[x.dtype], float(p), seed,
                                                          _ptr(sample_scale), eps_, _stream()), 'tgt_gelu_dropout_fwd')
        _prof_end('tgt_gelu_dropout_fwd', s, e)
        ctx.save_for_backward(x, sample_scale)
        ctx.p, ctx.seed, ctx.eps_ = float(p), seed, eps_
        return y

    @staticmethod
    def backward(ctx, dy):
        x, sample_scale = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        s, e = _prof_begin()
        _lib.check(_lib.lib().tgt_gelu_dropout_scaled_bwd(_ptr(x), _ptr(dy), _ptr(dx), x.numel(), _DT[x.dtype], ctx.p, ctx.seed,
                                                          _ptr(sample_scale), ctx.eps_, _stream()), 'tgt_gelu_dropout_bwd')
        _prof_end('tgt_gelu_dropout_bwd', s, e)
        return dx, None, None, None


def gelu_dropout(x, p, training, sample_scale=None):
    """dropout(gelu(x), p) in one pass each way (reference FFN, lib/tgt/layers/layers.py:157-158).
    The drop pattern comes from a per-call seed drawn from torch's CPU generator (no device sync).
    sample_scale (B,) float32: the result is multiplied by sample_scale[b] -- the DropPath factor of the residual
    branch, folded in here (see linear_residual_layer_norm(prescaled=True))."""
    p = float(p) if training else 0.0
    seed = _host_seed() if p > 0 else 0
    if sample_scale is not None and (x.numel() // x.shape[0]) % 8:
        raise RuntimeError('gelu_dropout: sample_scale needs a multiple of 8 elements per sample')
    return _GeluDropout.apply(x, p, seed, sample_scale)


class _MultiHotEmbed(torch.autograd.Function):
    """sum_f W[idx[..., f]]  as  counts(idx) @ W: both directions are small GEMMs
    instead of a gather and a sort-based scatter (the ATen embedding backward spends
    ~5 ms per call on these few-hundred-row tables).  Row `padding_idx` gets no
    gradient, like nn.Embedding(padding_idx=...).  Sums and weight gradients are float32 whatever
    autocast says: nn.Embedding is not an autocast op, the reference's embeddings stay fp32
    (lib/models/pcqm/layers.py:62-76)."""

    @staticmethod
    def forward(ctx, idx, weight, padding_idx, out_dtype):
        lead, V = idx.shape[:-1], weight.shape[0]
        flat = idx.reshape(-1, idx.shape[-1])
        counts = torch.zeros(flat.shape[0], V, dtype=out_dtype, device=weight.device)
        counts.scatter_add_(1, flat, torch.ones_like(flat, dtype=out_dtype))
        ctx.save_for_backward(counts)
        ctx.padding_idx, ctx.wdtype = padding_idx, weight.dtype
        return (counts @ weight.to(out_dtype)).view(*lead, weight.shape[1])

    @staticmethod
    def backward(ctx, g):
        counts, = ctx.saved_tensors
        g2 = g.reshape(counts.shape[0], -1).to(counts.dtype)
        P = _wgrad_chunks(counts.shape[0]) if (g2.is_cuda and g2.dtype == torch.float32 and g2.is_contiguous()) else 1
        if P > 1:
            # the contraction runs over all B*N*N pair rows into a (59, C) result: as ONE GEMM the library puts it on a
            # handful of workgroups (0.33 ms at the BASELINE batch); row chunks + a fixed-order sum are byte-bound
            gw = torch.empty(counts.shape[1], g2.shape[1], dtype=torch.float32, device=g2.device)
            _wgrad_into(gw, counts, g2, P)
        else:
            gw = counts.t() @ g2
        if ctx.padding_idx is not None:
            gw[ctx.padding_idx] = 0
        return None, gw.to(ctx.wdtype), None, None


class _GatherEmbed(torch.autograd.Function):
    """W[idx] for ONE index per row (the per-node tables of the Gaussian 3-D embedding: 1536 types x 2 columns): the forward is
    a gather; the weight gradient is counts(idx)^T @ g with the count matrix built in the backward only, over the index range
    the batch uses, and never saved (the count-matrix forward of _MultiHotEmbed kept a (rows, 1536) fp32 matrix alive per call).
    Fixed summation order (a GEMM, no atomics), no host read (nn.Embedding's sort-based backward has one)."""

    @staticmethod
    def forward(ctx, idx, weight, padding_idx):
        ctx.save_for_backward(idx)
        ctx.padding_idx, ctx.wshape, ctx.wdtype = padding_idx, weight.shape, weight.dtype
        return weight.index_select(0, idx.reshape(-1)).view(*idx.shape, weight.shape[1])

    @staticmethod
    def backward(ctx, g):
        idx, = ctx.saved_tensors
        V, C_ = ctx.wshape
        flat = idx.reshape(-1, 1)
        g2 = g.reshape(flat.shape[0], C_).float()
        counts = torch.zeros(flat.shape[0], V, dtype=torch.float32, device=g.device)
        counts.scatter_(1, flat, 1.0)
        gw = counts.t() @ g2
        if ctx.padding_idx is not None:
            gw[ctx.padding_idx] = 0
        return None, gw.to(ctx.wdtype), None


def gather_embed(idx, weight, padding_idx=None):
    """idx (...) long with values < weight.shape[0] -> (..., C) = weight[idx]; see _GatherEmbed"""
    _dev(weight)
    return _GatherEmbed.apply(idx, weight, padding_idx)


def multi_hot_embed(idx, weight, padding_idx=None, out_dtype=None):
    """idx: (..., F) long with values < weight.shape[0]; returns (..., C) = sum over F of rows,
    in weight.dtype unless out_dtype says otherwise."""
    if out_dtype is None:
        out_dtype = weight.dtype
    with torch.autocast('cuda', enabled=False):
        return _MultiHotEmbed.apply(idx, weight, padding_idx, out_dtype)


# ---------------------------------------------------------------------------
# Linear with a split-M weight gradient
# ---------------------------------------------------------------------------
def column_sum(x2):
    """float32 column sums of a contiguous (rows, C) tensor (bias gradients): HIP kernel when the
    shape qualifies (device tensor, >= 1024 rows, C a multiple of 8, <= 4096), else the library reduction."""
    rows, C_ = x2.shape
    if not (x2.is_cuda and x2.is_contiguous() and C_ % 8 == 0 and C_ <= 4096 and x2.dtype in _DT and rows >= 1024):
        return x2.sum(0, dtype=torch.float32)
    L = _lib.lib()
    out = torch.empty(C_, dtype=torch.float32, device=x2.device)
    partial = torch.empty(L.tgt_layer_norm_parts() * C_, dtype=torch.float32, device=x2.device)
    s, e = _prof_begin()
    _lib.check(L.tgt_colsum(_ptr(x2), _DT[x2.dtype], rows, C_, _ptr(out), _ptr(partial), _stream()), 'tgt_colsum')
    _prof_end('tgt_colsum', s, e)
    return out


# A kernel that writes a gradient tensor can accumulate its column sums on the way; the Linear
# whose output that gradient belongs to needs exactly those as its bias gradient.  The sums ride
# on the gradient tensor object (autograd hands the same object to the next node) together with
# the tensor's version counter: any in-place change (e.g. autograd accumulating a second
# gradient into it) or a new tensor object silently falls back to the separate reduction.
_colsum_handoffs = [0, 0]            # [offered, used]  (tests / diagnostics)


def _hand_colsum(grad, colsum):
    grad._tgt_colsum = (colsum, grad._version)
    _colsum_handoffs[0] += 1


def _take_colsum(grad, C_):
    tag = getattr(grad, '_tgt_colsum', None)
    if tag is None or tag[1] != grad._version or tag[0].numel() != C_ or grad.shape[-1] != C_:
        return None
    _colsum_handoffs[1] += 1
    # a FRESH alias: the tagged gradient tensor (and with it this tuple) can outlive the node -- e.g. d_res also flows down the
    # residual stream -- and a second owner of the object would make AccumulateGrad clone the sums instead of adopting them;
    # inside a Trainer's backward they may be registered sums that the flush has not computed yet (flush_deferred)
    return tag[0].detach()


# LayerNorm backward as the epilogue of the data-gradient GEMM that produces its dy (tgt_edge_linear, TGT_EPI_LN_BWD): the Linear
# that consumes a LayerNorm output does NOT compute dx = dz W in its backward when the LayerNorm entry that produced its input
# has said it will (`_tgt_lazy_ok` on the forward tensor): it returns a TOKEN -- a ZERO scalar expanded to the gradient's shape (no
# memory) carrying (dz, W) -- and the entry's backward runs GEMM + LayerNorm backward + stream-gradient add + dgamma / dbeta /
# bias-gradient sums as ONE launch: the 134 MB dy is neither written nor read back (0.126 ms against 0.060 + 0.135 at the BASELINE
# shape).  Same object-identity hand-over as _hand_colsum.
# The hand-over is safe for ANY use of y (round 4; the round-3 token was a NaN that a second consumer of y turned into a silent NaN
# gradient): the entry hangs a tensor hook on y (_LazyRec.hook) that sees the ACCUMULATED gradient of y before the entry's
# backward does.  One consumer, token untouched: it passes, and the entry fuses the product.  Anything else -- two Linears on the same
# y, a second non-lazy consumer, retain_grad, another hook -- and autograd has summed the (zero) tokens with whatever real gradients
# there were: the hook adds the products the tokens stand for, so the entry receives the complete gradient (unfused, correct).
_EPI_LN_BWD = K.epi_ln_bwd           # A/B knob
_zero_scalars = {}
_lazy_dgrads = [0, 0, 0]           # [offered, fused, materialized by the hook]  (tests / diagnostics)


class _LazyRec:
    """the tokens issued against one LayerNorm output y during a backward pass"""
    __slots__ = ('issued', 'y')

    def __init__(self, y):
        self.issued = []
        self.y = weakref.ref(y)

    def hook(self, grad):
        issued, self.issued = self.issued, []
        if not issued:
            return None
        y = self.y()
        watched = y is not None and (y.retains_grad or len(y._backward_hooks or ()) > 1)      # someone else looks at y's gradient
        if len(issued) == 1 and grad is issued[0] and _take_lazy_dgrad(grad) is not None and not watched:
            return None                         # the untouched token of the only consumer: the entry's backward fuses it
        total = grad
        for tok in issued:
            dz2, w, _ = tok._tgt_lazy
            prod = (dz2 @ w).view(grad.shape).to(grad.dtype)
            total = prod if total is None else total + prod
        _lazy_dgrads[2] += len(issued)
        return total


def _lazy_ok(x, weight, cd):
    """forward-time decision of a Linear: may its backward leave dx to the LayerNorm entry that produced x?  Returns the entry's
    _LazyRec (the backward issues its token against it) or False."""
    rec = getattr(x, '_tgt_lazy_ok', None)
    if (_EPI_LN_BWD and rec is not None and x.is_cuda and cd in (torch.bfloat16, torch.float16) and
            x.dtype == cd and weight.shape[1] == 256 and weight.shape[0] in (64, 128, 256) and
            x.numel() // 256 >= _EDGE_MIN_ROWS):
        return rec
    return False


def _lazy_dgrad(dz2, w, shape, rec=None):
    """the un-computed gradient dz2 @ w of `shape` (see above); rec: the _LazyRec of the tensor it is the gradient of"""
    key = (dz2.device, dz2.dtype)
    zero = _zero_scalars.get(key)
    if zero is None:
        zero = _zero_scalars[key] = torch.zeros((), dtype=dz2.dtype, device=dz2.device)
    token = zero.expand(shape)
    token._tgt_lazy = (dz2, w, token._version)
    if rec is not None:
        rec.issued.append(token)
    _lazy_dgrads[0] += 1
    return token


def _offer_lazy(s, y):
    """(s, y) of a residual + LayerNorm entry, y marked: this entry's backward accepts a lazy data gradient for y (_ln_backward)"""
    if _EPI_LN_BWD and y.is_cuda and y.requires_grad and y.shape[-1] == 256 and y.dtype == s.dtype and y.dtype in (torch.bfloat16, torch.float16):
        rec = _LazyRec(y)
        y._tgt_lazy_ok = rec
        y.register_hook(rec.hook)
    return s, y


def _take_lazy_dgrad(dy):
    """(dz2, w) when dy is a token of _lazy_dgrad, else None"""
    tag = getattr(dy, '_tgt_lazy', None)
    if tag is None:
        return None
    if tag[2] != dy._version:
        raise RuntimeError('a lazy data-gradient token was written to in place')
    return tag[0], tag[1]


def _materialize_dgrad(dy):
    """dy itself, or the product a token stands for"""
    lazy = _take_lazy_dgrad(dy)
    return dy if lazy is None else (lazy[0] @ lazy[1]).view(dy.shape)


def _ln_backward(dy, s, g, mean, rstd, ds, scale, rps, want_dz):
    """(d_res, d_z, dgamma, dbeta, colsum of d_z) of the residual entry  s = res + scale z, y = LayerNorm(s; g):
    d_res = ds + LN_bwd(dy), d_z = scale * d_res (a separate tensor only when want_dz; else d_res stands for it and only the
    column sums carry the factor).  dy a lazy token of a K <= 256 data gradient: one fused launch; else tgt_add_layer_norm_bwd."""
    N = s.shape[-1]
    rows = s.numel() // N
    L = _lib.lib()
    ds = None if ds is None else ds.contiguous()
    d_res = torch.empty_like(s)
    d_z = torch.empty_like(s) if want_dz else None
    lazy = _take_lazy_dgrad(dy)
    if lazy is not None and N == 256 and lazy[0].dtype == s.dtype and (ds is None or ds.dtype == s.dtype) and lazy[0].is_contiguous():
        dz2, w = lazy
        partial = torch.empty(L.tgt_edge_linear_parts(rows, N), 3 * N, dtype=torch.float32, device=s.device)
        edge_linear_raw(dz2, weight_t(w), None, _lib.EPI_LN_BWD, ln=(g, None, 1e-5), stats=(mean, rstd), res=s.view(rows, N),
                        ds_in=None if ds is None else ds.view(rows, N), out=d_res.view(rows, N),
                        out2=None if d_z is None else d_z.view(rows, N), row_scale=scale, rows_per_sample=rps if scale is not None else 0,
                        colsum_partial=partial)
        with _on_stream(_terminal_fork(rows, partial)):
            tot = sum_planes(partial, torch.empty(3 * N, dtype=torch.float32, device=s.device))
        _lazy_dgrads[1] += 1
        return d_res, d_z, tot[:N], tot[N:2 * N], tot[2 * N:]
    dy = _materialize_dgrad(dy).contiguous()
    dg = torch.empty(N, dtype=torch.float32, device=s.device)
    db_cs = torch.empty(2 * N, dtype=torch.float32, device=s.device)
    partial = torch.empty(L.tgt_layer_norm_parts() * 3 * N, dtype=torch.float32, device=s.device)
    p0, p1 = _prof_begin()
    _lib.check(L.tgt_add_layer_norm_bwd(_ptr(dy), _DT[dy.dtype], _ptr(s), _DT[s.dtype], _ptr(ds),
                                        0 if ds is None else _DT[ds.dtype], _ptr(scale), rps, _ptr(g), _ptr(mean),
                                        _ptr(rstd), _ptr(d_res), _ptr(d_z), _DT[s.dtype], _ptr(dg), _ptr(db_cs),
                                        db_cs.data_ptr() + 4 * N, _ptr(partial), rows, N, _stream()),
               'tgt_add_layer_norm_bwd')
    _prof_end('tgt_add_layer_norm_bwd', p0, p1)
    return d_res, d_z, dg, db_cs[:N], db_cs[N:]


_WGRAD_BIG = 131072      # outputs at least this large (lin_O 256x512, the fused projection) take 64 chunks: +0.3 % same-box over 262144
_WGRAD_MAXP = K.wgrad_maxp      # settled (in-step sweep 32..256, profiles/HISTORY_rounds_1-4.md 4.6): cap on the row chunks of a weight gradient


def _wgrad_chunks(M, out_in=0):
    """number of row chunks for dW = sum_c dY_c^T X_c (rows per chunk >= 1024, <= 128 chunks; 64
    for the 1600x256 fused projection, whose fp32 partials are 1.6 MB each:
    tools/wgrad_chunk_probe.py)"""
    for P in ((64, 32, 16, 8, 4, 2) if out_in >= _WGRAD_BIG else (256, 128, 64, 32, 16, 8, 4, 2)):
        if P <= _WGRAD_MAXP and M % P == 0 and M // P >= 1024:
            return P
    return 1


def _wgrad_into(out, dy2, x2, chunks):
    """out (rows(dy2^T), in) fp32 <- dy2^T x2 as `chunks` batched partial products + their sum; dy2 may
    be a column slice of a wider row-major matrix"""
    M = x2.shape[0]
    P = chunks
    while P > 1 and (M % P or M // P < 1024):
        P //= 2
    if P > 1:
        if dy2.dtype != torch.float32:
            part = _gemm.wgrad_chunks(dy2, x2, P)          # (torch.bmm(..., out_dtype=float32) from a cached plan: tgt_amd/gemm.py)
        else:
            part = torch.bmm(dy2.unflatten(0, (P, M // P)).transpose(1, 2), x2.view(P, M // P, -1))
        sum_planes(part, out)
    else:
        out.copy_(dy2.t() @ x2)


def _as_dtype(p, cd):
    """p in dtype cd; a Trainer-maintained low-precision shadow (`p._lp`, refreshed by the Adam
    kernel) is used instead of a cast kernel when it matches."""
    if p.dtype == cd:
        return p
    lp = getattr(p, '_lp', None)
    if lp is not None and lp.dtype == cd:
        return lp
    return p.to(cd)


def _as_dtype_view(p, cd):
    """the tensor to READ parameter p from when a kernel converts to cd itself: the 16-bit shadow
    when it is there and matches, else the parameter"""
    lp = getattr(p, '_lp', None)
    return lp if (lp is not None and lp.dtype == cd and p.dtype != cd) else p.detach()


# opt-in (TGT_WGRAD_STREAM=1): weight gradients of the edge Linears on a third stream.  Measured over 20 same-box pairs on five boxes:
# +1.7 % on one, -1.9 % on another, occasional runs 5-10 % low -- two equal-priority queues of HBM-bound kernels; the mean is ~0
_WGRAD_STREAM = K.wgrad_stream
_WGRAD_KEEP = K.wgrad_keep            # forked operands kept referenced in a window instead of Tensor.record_stream
_WGRAD_DEPTH = K.wgrad_depth
_wgrad_streams = {}
_wgrad_window = {}                    # device -> deque of [event on the forked stream | None, operands, origin stream]
_trainer_backward = [0]           # > 0 while a Trainer runs its backward: the only caller that joins the forked stream afterwards


# Gradients written where the optimizer reads them.  A Trainer keeps one flat float32 gradient buffer; autograd hands every
# parameter its own gradient tensor, and the Trainer's gradient collection then copies all of them into the buffer (414 MB each
# way, 0.34 ms of multi-tensor copies and ~1 ms of host time per step).  Inside a Trainer's backward the kernels that END a
# weight gradient (the fixed-order plane sum of a split-M product, the scatter of a fused projection's gradient) write into the
# parameter's slice of that buffer instead and return a fresh alias of it: AccumulateGrad adopts the alias (single owner, same
# layout), the collection finds the gradient already in place and skips it.  Only there: any other backward (tests,
# torch.autograd.grad) gets ordinary tensors, and a Trainer whose parameters enter the graph more than once never opens the
# context (step.py: _parameter_reused) -- autograd would add into the slice a second time.
_FLAT_GRAD = {}                   # parameter data_ptr -> (float32 view of the flat gradient buffer shaped like the parameter)
_grad_dst_seen = set()            # parameter data_ptrs handed a destination during the current Trainer backward (see _grad_dst)
_FLAT_GRAD_ON = K.flat_grad_dst


def register_flat_grads(params, views):
    for p_, v in zip(params, views):
        _FLAT_GRAD[p_.data_ptr()] = v


def unregister_flat_grads(ptrs):
    for q in ptrs:
        _FLAT_GRAD.pop(q, None)


def _grad_dst(ptr, shape, dtype):
    """a fresh alias of the flat-buffer slice that belongs to the parameter at `ptr`, or None (not inside a Trainer's backward,
    not registered, another shape / dtype)"""
    if not (_FLAT_GRAD_ON and _trainer_backward[0] > 0 and ptr is not None):
        return None
    v = _FLAT_GRAD.get(ptr)
    if v is None or v.dtype != dtype or tuple(v.shape) != tuple(shape):
        return None
    # One gradient per parameter and backward: the Trainer checked the graph of its FIRST step for parameters that enter it twice
    # (step.py: _parameter_reused); a graph that changes later (a module applied twice in some steps, weights tied after step 1)
    # would have two kernels write this slice and autograd sum two aliases of it -- silently wrong.  Caught here, every step.
    if ptr in _grad_dst_seen:
        raise RuntimeError('tgt_amd: a parameter received a second weight gradient inside one Trainer backward (it enters the '
                           'autograd graph more than once: weight sharing that the first step did not show).  Construct the '
                           'Trainer after tying the weights, or set TGT_FLAT_GRAD_DST=0')
    _grad_dst_seen.add(ptr)
    return v.detach()


class trainer_backward:
    """`with ops.trainer_backward():` around loss.backward() -- inside, parameter gradients may be produced on a forked stream
    (autograd does not know about it: the stream switch happens inside a node).  The caller MUST order its reads of the
    gradients after ops.wait_side_streams(); the Trainer's gradient collection does.  Anyone else (torch.autograd.grad in a
    test, another optimizer) gets every gradient on the current stream."""

    def __enter__(self):
        if _trainer_backward[0] == 0:
            _grad_dst_seen.clear()
        _trainer_backward[0] += 1

    def __exit__(self, *exc):
        if _trainer_backward[0] == 1:
            flush_deferred()               # (before the count drops: nothing stays registered past the backward)
        _trainer_backward[0] -= 1
        return False


def _wgrad_fork(*tensors, rows=None):
    """The stream a weight gradient may run on -- nothing consumes it before the step ends (the Trainer's gradient collection waits
    for every stream, wait_side_streams) -- ordered after the work queued so far on the current stream; None = stay.  Only inside a
    Trainer's backward (ops.trainer_backward), only for the edge rows, never from the node side stream."""
    t0 = tensors[0]
    if not (_WGRAD_STREAM and _trainer_backward[0] > 0 and t0.is_cuda and (rows if rows is not None else t0.shape[0]) >= _EDGE_MIN_ROWS):
        return None
    dev = t0.device
    cur = torch.cuda.current_stream(dev)
    if _side_streams.get(dev) == cur:
        return None
    ws = _wgrad_streams.get(dev)
    if ws is None:
        ws = _wgrad_streams[dev] = torch.cuda.Stream(dev, priority=0)
    ws.wait_stream(cur)
    _main_streams.setdefault(dev, cur)
    if not _WGRAD_KEEP:
        for t in tensors:
            t.record_stream(ws)
        return ws
    # The operands must outlive the forked work.  record_stream would do it, but then the allocator holds every freed operand
    # until an event on the forked stream completes, the host runs many layers ahead of the GPU and asks for fresh memory
    # instead: the reserve grew from 69 to 130-200 GB and one run in eight lost 10 % (profiles/r05y_ab_wgrad_stream.txt).
    # Instead the operands of the last _WGRAD_DEPTH forks stay referenced here; the stream they came from is made to wait for
    # the work of the fork that falls out of the window and only then is that fork's reference dropped: whatever reuses the
    # memory on that stream is ordered behind the forked kernels that read it.
    q = _wgrad_window.setdefault(dev, collections.deque())
    if q and q[-1][0] is None:               # close the previous fork: an event on ws behind everything it was given
        ev = torch.cuda.Event()
        ev.record(ws)
        q[-1][0] = ev
    q.append([None, tensors, cur])
    while len(q) > _WGRAD_DEPTH:
        ev, _old, origin = q.popleft()
        origin.wait_event(ev)
    return ws


def _on_stream(ws):
    return torch.cuda.stream(ws) if ws is not None else contextlib.nullcontext()


_TERMINAL_SUMS = K.terminal_sums     # A/B knob: the closing sums of parameter gradients on the forked stream too


def _terminal_fork(rows, *tensors):
    """as _wgrad_fork, for the small closing sums (column-sum partials -> a bias / LayerNorm parameter gradient) behind a kernel
    that is already queued: nothing on the step's own chain reads their result"""
    return _wgrad_fork(*tensors, rows=rows) if _TERMINAL_SUMS else None


def _param_grad(t, dtype):
    """t in the parameter's dtype.  A gradient computed on the forked stream is only ever RETURNED to autograd; if it needs a
    cast kernel first (parameters that are not float32), that kernel runs on the current stream and has to wait."""
    if t is None or t.dtype == dtype:
        return t
    flush_deferred()                      # (t may be a registered, not yet computed sum)
    wait_side_streams(t.device)
    return t.to(dtype)


_SIDE_PRIO = K.side_prio           # A/B knob: HIP priority of the node side stream (-1 = high: its short kernels are
#                                                                    dispatched ahead of the edge kernels' next workgroups; +0.5 % over 5 same-box pairs)


class WeightTransposes:
    """W^T (contiguous) of the 16-bit weight shadows a Trainer maintains, for the data-gradient launches of tgt_edge_linear.
    The first request for a weight computes and registers its transpose; `refresh()` -- called by the Trainer right after every
    optimizer step / shadow refresh, the only places the shadow changes -- rewrites all registered ones in ONE launch
    (tgt_transpose_many) instead of one copy kernel per Linear and backward launch (96 a step, each with its dispatch gap).
    Only tensors INSIDE the registered shadow buffer are cached (looked up by address: saved tensors come back from autograd as
    new Python objects); everything else gets a fresh transpose."""
    live = []

    def __init__(self, shadow):
        self.shadow = shadow
        self.lo, self.hi = shadow.data_ptr(), shadow.data_ptr() + shadow.numel() * shadow.element_size()
        self.entries = {}               # address -> (W^T, shape of W)
        self.table = None
        WeightTransposes.live.append(self)

    def close(self):
        if self in WeightTransposes.live:
            WeightTransposes.live.remove(self)
        self.entries.clear()
        self.table = None

    def get(self, w):
        e = self.entries.get(w.data_ptr())
        if e is not None and e[1] == tuple(w.shape):
            return e[0]
        t = w.t().contiguous()
        self.entries[w.data_ptr()] = (t, tuple(w.shape))
        self.table = None               # (rebuilt at the next refresh)
        return t

    def refresh(self):
        if not self.entries:
            return
        if self.table is None:
            rows = []
            for ptr, (t, shape) in self.entries.items():
                rows += [ptr, t.data_ptr(), shape[0] | (shape[1] << 32)]        # {src, dst, int32 rows, int32 cols}
            self.table = torch.tensor(rows, dtype=torch.int64).to(self.shadow.device)
        _lib.check(_lib.lib().tgt_transpose_many(_ptr(self.table), len(self.entries), 16, _stream()), 'tgt_transpose_many')


_WT_CACHE = K.wt_cache          # A/B knob


def weight_t(w):
    """w.t().contiguous() for a 2-D 16-bit weight; served from a Trainer's WeightTransposes when w lives in its shadow"""
    if _WT_CACHE and w.dim() == 2 and w.is_contiguous() and w.element_size() == 2:
        ptr = w.data_ptr()
        for c in WeightTransposes.live:
            if c.lo <= ptr < c.hi and w.dtype == c.shadow.dtype:
                return c.get(w)
    return w.t().contiguous()


_EDGE_GEMM = K.edge_gemm          # A/B knob: the weight-resident slice kernel where it wins
_EDGE_N512 = K.edge_n512          # A/B knob: lin_O's (256 -> 512) and the narrow (-> <= 128) data gradients on the weight-resident kernels
_EDGE_MIN_ROWS = 65536                                            # (tests lower it)


def _edge_kernel_ok(x2, N, cd, ks=(64, 128, 256)):
    """edge-row GEMMs the kernels of csrc/edge_gemm.hip take: device, 16-bit, K in {64,128,256} (512: the residual entry of lin_O), many rows"""
    return (_EDGE_GEMM and x2.is_cuda and cd in (torch.bfloat16, torch.float16) and x2.shape[1] in ks and
            N % 8 == 0 and x2.shape[0] >= _EDGE_MIN_ROWS and x2.stride(-1) == 1 and (x2.stride(0) * 2) % 16 == 0)


def _linear_forward(x, weight, bias, cd):
    """(x2, w, y): the operands saved for the backward and y = x W^T + b in dtype cd"""
    xs = x.shape
    x2 = x.reshape(-1, xs[-1])
    if x2.dtype != cd:
        x2 = x2.to(cd)
    w = _as_dtype(weight, cd)
    b = None if bias is None else _as_dtype(bias, cd)
    # the result must own its storage (not be a view): the layer adds the residual in place
    y = torch.empty(*xs[:-1], weight.shape[0], dtype=cd, device=x.device)
    y2 = y.view(-1, weight.shape[0])
    if weight.shape[0] <= 128 and w.is_contiguous() and _edge_kernel_ok(x2, weight.shape[0], cd):
        # narrow outputs (lin_EG: 128, third-arm E/G: 64): 41 vs 47 us and 29 vs 37 us against the tuned library GEMM
        edge_linear_raw(x2, w, b, out=y2)
        return x2, w, y
    _gemm.linear_tn(x2, w, b, out=y2)                      # (torch.mm / torch.addmm(out=) from a cached plan)
    return x2, w, y


def _linear_backward(x2, w, dy2, xs, xdt, wdt, bdt, need_dx, need_dw, need_db, lazy_dx=False, dw_post=None, fork=True, dw_ptr=None):
    """(dx, dW, db) of y = x W^T + b.  dW = dY^T X contracts over M = B*N*N = 262144 rows into a
    tiny (out,in) result; as one GEMM the library runs it on a handful of workgroups
    (0.4-0.8 ms), as 64-128 independent chunk products + an fp32 sum it is HBM-bound (57 us for
    256x256, measured; tools/wgrad_probe.py)."""
    if dy2.dtype != w.dtype:
        dy2 = dy2.to(w.dtype)
    dx = dw = db = None
    _gate_wait(dy2)
    # dw_post: what the caller still does to dW (it must run where dW was computed); fork=False: the caller reads dW itself
    ws = _wgrad_fork(dy2, x2) if (need_dw and fork) else None  # (forked BEFORE the data gradient is queued: it waits for dy only)
    if need_dx:
        # lazy_dx: the LayerNorm entry that produced x runs this GEMM itself, fused with its own backward (_lazy_dgrad)
        if lazy_dx and xdt == dy2.dtype:
            dx = _lazy_dgrad(dy2.contiguous(), w, xs, lazy_dx if isinstance(lazy_dx, _LazyRec) else None)
        elif _EDGE_N512 and _edge_kernel_ok(dy2, w.shape[1], dy2.dtype) and \
                ((w.shape[0] == 256 and w.shape[1] == 512) or w.shape[1] <= 128):
            # data gradients the weight-resident kernels win: lin_O's (256 -> 512 channels: 3 E at ~5 TB/s, library 3.5) on
            # edge_wide512_kernel, narrow ones (lin_O_e: 256 -> 64) on the slice kernel
            dx = edge_linear_raw(dy2.contiguous(), weight_t(w)).view(xs).to(xdt)
        else:
            dx = _gemm.matmul_nn(dy2, w).view(xs).to(xdt)
    if need_dw:
        M = x2.shape[0]
        P = _wgrad_chunks(M, dy2.shape[1] * x2.shape[1])
        with _on_stream(ws):
            if P > 1:
                part = _gemm.wgrad_chunks(dy2, x2, P) if dy2.dtype != torch.float32 else \
                    torch.bmm(dy2.view(P, M // P, -1).transpose(1, 2), x2.view(P, M // P, -1))
                # (dw_ptr: the parameter this gradient belongs to -- inside a Trainer's backward the sum lands in its slice of the
                #  flat gradient buffer, see _grad_dst)
                dst = _grad_dst(dw_ptr, part.shape[1:], torch.float32) if (dw_post is None and wdt == torch.float32) else None
                dw = sum_planes(part, dst if dst is not None else torch.empty(part.shape[1:], dtype=torch.float32, device=part.device))
                if wdt != torch.float32:
                    flush_deferred()
                    dw = dw.to(wdt)
            elif P == 1:
                dst = _grad_dst(dw_ptr, (dy2.shape[1], x2.shape[1]), wdt) if dw_post is None else None
                dw = (dy2.t() @ x2).to(wdt) if dst is None else dst.copy_(dy2.t() @ x2)      # (the same cast, into the slice)
            if dw_post is not None:
                dw = dw_post(dw)
    if need_db:
        with _on_stream(ws):             # (a parameter gradient as well: nothing reads it before the step ends)
            db = column_sum(dy2).to(bdt)
    return dx, dw, db


class _Linear(torch.autograd.Function):
    """y = x W^T + b on the library GEMM, with the weight gradient computed as a BATCHED GEMM
    over row chunks (see _linear_backward)."""

    @staticmethod
    def forward(ctx, x, weight, bias, cd, lazy=False):
        x2, w, y = _linear_forward(x, weight, bias, cd)
        ctx.save_for_backward(x2, w)
        ctx.meta = (x.shape, x.dtype, weight.dtype, None if bias is None else bias.dtype)
        ctx.lazy = lazy
        ctx.wptr = weight.data_ptr()
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        xs, xdt, wdt, bdt = ctx.meta
        need_db = bdt is not None and ctx.needs_input_grad[2]
        cs = _take_colsum(dy, dy.shape[-1]) if need_db else None
        dx, dw, db = _linear_backward(x2, w, dy.reshape(-1, dy.shape[-1]), xs, xdt, wdt, bdt,
                                      ctx.needs_input_grad[0], ctx.needs_input_grad[1], need_db and cs is None, ctx.lazy,
                                      dw_ptr=ctx.wptr)
        if cs is not None:
            db = _param_grad(cs, bdt)
        return dx, dw, db, None, None


def _permute_cols(src, idx, dtype, after_sums=False):
    """src[:, idx] as `dtype`; after_sums: src may be a pending sum (see flush_deferred): the launch queues behind it"""
    out = torch.empty(src.shape, dtype=dtype, device=src.device)
    keep = [src.detach(), out.detach()]                 # aliases (see sum_planes): `out` itself must keep a single owner
    args = (_ptr(src), _DT[src.dtype], _ptr(idx), _ptr(out), _DT[dtype], src.shape[0], src.shape[1])

    def run():
        _lib.check(_lib.lib().tgt_permute_cols(*args, _stream()), 'tgt_permute_cols')
        keep.clear()
    if after_sums:
        _after_sums(run)
    else:
        run()
    return out


class _LinearPermutedCols(torch.autograd.Function):
    """y = x W[:, idx]^T + b: the weight's input columns re-ordered (and cast) by one launch,
    the weight gradient re-ordered back by one launch (lin_O of the triplet modules)."""

    @staticmethod
    def forward(ctx, x, weight, bias, cd, idx, inv):
        _dev(x, weight)
        w = _permute_cols(_as_dtype_view(weight, cd).contiguous(), idx, cd)
        x2, w, y = _linear_forward(x, w, bias, cd)
        ctx.save_for_backward(x2, w, inv)
        ctx.meta = (x.shape, x.dtype, weight.dtype, None if bias is None else bias.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, w, inv = ctx.saved_tensors
        xs, xdt, wdt, bdt = ctx.meta
        need_db = bdt is not None and ctx.needs_input_grad[2]
        cs = _take_colsum(dy, dy.shape[-1]) if need_db else None
        dx, dw, db = _linear_backward(x2, w, dy.reshape(-1, dy.shape[-1]), xs, xdt, torch.float32, bdt,
                                      ctx.needs_input_grad[0], ctx.needs_input_grad[1], need_db and cs is None,
                                      dw_post=lambda t: _permute_cols(t.contiguous(), inv, wdt, after_sums=True))
        if cs is not None:
            db = _param_grad(cs, bdt)
        return dx, dw, db, None, None, None


def linear_permuted_cols(x, weight, bias, idx, inv):
    """linear(x, weight[:, idx], bias); idx / inv: int32 device tensors (a permutation and its inverse)"""
    cd = torch.get_autocast_dtype('cuda') if (x.is_cuda and torch.is_autocast_enabled('cuda')) else x.dtype
    return _LinearPermutedCols.apply(x, weight, bias, cd, idx, inv)


class _FusedLinear(torch.autograd.Function):
    """y = x W^T + b where (W, b) are rows gathered from several nn.Linear (ParamTable): fuse,
    GEMM / dgrad, wgrad, unfuse -- one launch each for the parameter plumbing."""

    @staticmethod
    def forward(ctx, x, cd, table, *params):
        _dev(x)
        weight, bias = _fuse_params(table, params, cd)
        x2, w, y = _linear_forward(x, weight, bias, cd)
        ctx.save_for_backward(x2, w, *params)
        ctx.table, ctx.meta = table, (x.shape, x.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors[:2]
        params = ctx.saved_tensors[2:]
        xs, xdt = ctx.meta
        need_p = any(ctx.needs_input_grad[3:])
        dx, dw, db = _linear_backward(x2, w, dy.reshape(-1, dy.shape[-1]), xs, xdt, torch.float32, torch.float32,
                                      ctx.needs_input_grad[0], need_p, need_p, fork=False)
        grads = _unfuse_grads(ctx.table, params, dw, db) if need_p else (None,) * len(params)
        return (dx, None, None, *grads)


def fused_linear(x, table, params):
    """linear(x, W, b) with (W, b) assembled from `params` = (w0, b0, w1, b1, ...) by `table`"""
    cd = torch.get_autocast_dtype('cuda') if (x.is_cuda and torch.is_autocast_enabled('cuda')) else x.dtype
    return _FusedLinear.apply(x, cd, table, *params)


def linear(x, weight, bias=None):
    """F.linear replacement for the TGT modules (autocast-aware: computes in the autocast
    dtype when autocast is on, else in x.dtype)."""
    cd = torch.get_autocast_dtype('cuda') if (x.is_cuda and torch.is_autocast_enabled('cuda')) else x.dtype
    return _Linear.apply(x, weight, bias, cd, _lazy_ok(x, weight, cd))


# ---------------------------------------------------------------------------
# edge-channel Linear with fused LayerNorm prologue / GELU / residual / backward epilogues
# ---------------------------------------------------------------------------
def edge_linear_raw(a, w, bias=None, epilogue=_lib.EPI_BIAS, *, ln=None, y=None, out=None, out2=None, res=None, ds_in=None,
                    row_scale=None, out_scale=None, rows_per_sample=0, dropout=(0.0, 0), stats=None, colsum_partial=None,
                    flags=0, dw_partial=None):
    """One launch of tgt_edge_linear on 2-D operands (rows may be strided views with a contiguous last axis).
    a (M,K), w (N,K), bias (N) in one 16-bit dtype; ln = (gamma, beta, eps) float32 for the LayerNorm prologue /
    the LN_BWD epilogue; stats = (mean, rstd) float32 (M) (written by the prologue, read by LN_BWD).
    Returns out (allocated when not given).  No fallback: an unsupported shape raises."""
    _dev(a, w, bias, out, out2, res, ds_in, y)
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=a.dtype, device=a.device)
    g = _lib.EdgeLinearArgs()
    g.M, g.K, g.N, g.dtype, g.epilogue = M, K, N, _DT[a.dtype], epilogue

    def mat(t, name):
        if t is None:
            return None, 0
        if t.dtype != a.dtype or t.stride(-1) != 1 or t.shape[0] != M:
            raise RuntimeError(f'edge_linear: operand {name} must be ({M}, ..) {a.dtype} with a contiguous last axis, got '
                               f'{tuple(t.shape)} {t.dtype} stride {t.stride()}')
        return t.data_ptr(), t.stride(0)

    g.a, g.lda = mat(a, 'a')
    if w.dtype != a.dtype or w.stride(-1) != 1 or w.shape[1] != K:
        raise RuntimeError(f'edge_linear: weight must be (N, {K}) {a.dtype}, got {tuple(w.shape)} {w.dtype}')
    g.w, g.ldw = w.data_ptr(), w.stride(0)
    if bias is not None:
        if bias.dtype != a.dtype or not bias.is_contiguous() or bias.numel() != N:
            raise RuntimeError('edge_linear: bias must be a contiguous (N,) tensor of the operand dtype')
        g.bias = bias.data_ptr()
    if ln is not None:
        gamma, beta, eps = ln
        assert gamma.dtype == torch.float32 and gamma.is_contiguous()
        g.gamma, g.eps = gamma.data_ptr(), float(eps)
        if beta is not None:
            assert beta.dtype == torch.float32 and beta.is_contiguous()
            g.beta = beta.data_ptr()
    if stats is not None:
        g.mean, g.rstd = stats[0].data_ptr(), stats[1].data_ptr()
    g.y, g.ldy = mat(y, 'y')
    g.out, g.ldo = mat(out, 'out')
    g.out2, g.ldo2 = mat(out2, 'out2')
    g.res, g.ldr = mat(res, 'res')
    g.ds_in, g.ld_ds = mat(ds_in, 'ds_in')
    if row_scale is not None:
        g.row_scale = row_scale.data_ptr()
    if out_scale is not None:
        g.out_scale = out_scale.data_ptr()
    g.rows_per_sample = int(rows_per_sample)
    g.dropout_p, g.dropout_seed = float(dropout[0]), int(dropout[1]) & 0xFFFFFFFFFFFFFFFF
    g.flags = int(flags)
    if colsum_partial is not None:
        g.colsum_partial = colsum_partial.data_ptr()
        g.colsum_rows = colsum_partial.shape[0]          # (N = 256: the launch uses exactly this many workgroups, one row each)
    if dw_partial is not None:
        # fused weight gradient (ABI 30): one (N, K) fp32 plane per persistent workgroup, every plane written
        if dw_partial.dtype != torch.float32 or not dw_partial.is_contiguous() or tuple(dw_partial.shape[1:]) != (N, K) or \
                (colsum_partial is not None and colsum_partial.shape[0] != dw_partial.shape[0]):
            raise RuntimeError(f'edge_linear: dw_partial must be contiguous float32 (parts, {N}, {K}) with as many planes as colsum_partial has rows')
        g.dw_partial = dw_partial.data_ptr()
        g.colsum_rows = dw_partial.shape[0]
    _call('tgt_edge_linear', _lib.lib().tgt_edge_linear, g)
    return out


def edge_linear_supported(K, N, dtype, epilogue=_lib.EPI_BIAS, ln=False, row_scale=False):
    if dtype not in (torch.bfloat16, torch.float16):
        return False
    g = _lib.EdgeLinearArgs()
    g.M, g.K, g.N, g.dtype, g.epilogue = 128, K, N, _DT[dtype], epilogue
    one = C.c_void_p(16)
    if row_scale:
        g.row_scale, g.rows_per_sample = one, 1
    if ln or epilogue == _lib.EPI_LN_BWD:
        g.gamma = g.beta = g.mean = g.rstd = g.res = one
    return bool(_lib.lib().tgt_edge_linear_supported(C.byref(g)))


# A/B knob: lin_W1's bias gradient out of the activation's backward pass (tgt_gelu_dropout_bwd_colsum, ABI 24).  Parity-green; in the step
# it LOSES 0.25 % (2496 / 2498 vs 2501 / 2505 graphs/s same-box): the 4096-workgroup grid-stride form and the eight accumulators per
# vector cost the streaming kernel more than the 22 us column-sum pass they replace.  Off by default.
_GELU_BWD_COLSUM = False       # settled off (-0.25 % in the step); tests patch this to cover the ABI entry
_FFN_GELU_EPI = K.ffn_gelu_epi     # A/B knob: lin_W1 + GELU + dropout as one launch on the edge rows
# A/B knob: GELU backward as the epilogue of lin_W2's data-gradient GEMM (the closing node takes the activation detached and returns the
# gradient of the pre-activation).  Round 2: neutral (the row phase of that epilogue was instruction-bound, 0.122 ms against 0.068 + 0.072
# for GEMM + activation pass).  Round 3, after the row-phase rewrite (0.101 ms): +1.5 % graphs/s same-box (2749.7 / 2750.0 against
# 2709.9 / 2708.8, profiles/r03e_ab.txt): on by default
_FFN_GELU_BWD_EPI = K.ffn_gelu_bwd_epi


class _LinearGeluDropout(torch.autograd.Function):
    """act = dropout(gelu(x W^T + b), p) * sample_scale[b] as ONE launch (tgt_edge_linear, TGT_EPI_GELU on the row-phase
    kernel): the pre-activation is written once for the backward and never re-read by a separate activation pass
    (reference FFN, lib/tgt/layers/layers.py:155-158).  Same drop pattern and arithmetic as tgt_gelu_dropout_scaled_fwd on
    the stored pre-activation; the backward is that kernel's backward followed by the Linear's."""

    @staticmethod
    def forward(ctx, x, weight, bias, cd, p, seed, sample_scale, lazy=False):
        _dev(x, weight, sample_scale)
        ctx.lazy = lazy
        xs = x.shape
        x2 = x.reshape(-1, xs[-1])
        if x2.dtype != cd:
            x2 = x2.to(cd)
        w = _as_dtype(weight, cd)
        b = None if bias is None else _as_dtype(bias, cd)
        N = weight.shape[0]
        pre = torch.empty(*xs[:-1], N, dtype=cd, device=x.device)
        act = torch.empty_like(pre)
        rps = (x2.shape[0] // sample_scale.numel()) if sample_scale is not None else 0
        edge_linear_raw(x2, w, b, _lib.EPI_GELU, out=act.view(-1, N), out2=pre.view(-1, N), dropout=(p, seed),
                        row_scale=sample_scale, rows_per_sample=rps)
        ctx.save_for_backward(x2, w, pre, sample_scale)
        ctx.meta = (xs, x.dtype, weight.dtype, None if bias is None else bias.dtype, float(p), seed, rps * N)
        ctx.wptr = weight.data_ptr()
        ctx.set_materialize_grads(False)
        # TWO outputs: the activation, and the pre-activation as a differentiable value of its own -- a consumer that owns the
        # activation's derivative (linear_residual_layer_norm: GELU backward as the epilogue of its data-gradient GEMM) reads
        # the activation detached and sends its gradient to `pre` directly
        return act, pre

    @staticmethod
    def backward(ctx, d_act, d_pre_in):
        x2, w, pre, sample_scale = ctx.saved_tensors
        xs, xdt, wdt, bdt, p, seed, eps_ = ctx.meta
        d_pre = None
        need_db = bdt is not None and ctx.needs_input_grad[2]
        cs = None
        if d_act is not None:
            d_act = d_act.contiguous()
            d_pre = torch.empty_like(pre)
            L = _lib.lib()
            N = pre.shape[-1]
            vec = 16 // pre.element_size()
            s, e = _prof_begin()
            if _GELU_BWD_COLSUM and need_db and d_pre_in is None and N % vec == 0 and (256 * vec) % N == 0:
                # the bias gradient of lin_W1 (column sums of d_pre) rides on the activation's backward pass: no separate
                # reduction pass over the 134 MB gradient
                cs = torch.empty(N, dtype=torch.float32, device=pre.device)
                partial = torch.empty(L.tgt_gelu_colsum_parts() * N, dtype=torch.float32, device=pre.device)
                _lib.check(L.tgt_gelu_dropout_bwd_colsum(_ptr(pre), _ptr(d_act), _ptr(d_pre), pre.numel(), _DT[pre.dtype], p, seed,
                                                         _ptr(sample_scale), eps_, N, _ptr(partial), _ptr(cs), _stream()),
                           'tgt_gelu_dropout_bwd_colsum')
            else:
                _lib.check(L.tgt_gelu_dropout_scaled_bwd(_ptr(pre), _ptr(d_act), _ptr(d_pre), pre.numel(), _DT[pre.dtype], p, seed,
                                                         _ptr(sample_scale), eps_, _stream()), 'tgt_gelu_dropout_bwd')
            _prof_end('tgt_gelu_dropout_bwd', s, e)
        if d_pre_in is not None:
            if d_pre is None and need_db:
                cs = _take_colsum(d_pre_in, d_pre_in.shape[-1])        # (the GELU_BWD epilogue that wrote it summed its columns)
            d_pre = d_pre_in.contiguous() if d_pre is None else d_pre + d_pre_in
        if d_pre is None:
            return None, None, None, None, None, None, None, None
        dx, dw, db = _linear_backward(x2, w, d_pre.view(-1, d_pre.shape[-1]), xs, xdt, wdt, bdt, ctx.needs_input_grad[0],
                                      ctx.needs_input_grad[1], need_db and cs is None, ctx.lazy, dw_ptr=ctx.wptr)
        if need_db and cs is not None:
            db = _param_grad(cs, bdt)
        return dx, dw, db, None, None, None, None, None


def linear_gelu_dropout_ok(x, weight, sample_scale=None):
    """whether linear_gelu_dropout takes this call: edge rows (many of them), 16-bit compute dtype, 256 outputs, K in {64,128,256}"""
    if not (_FFN_GELU_EPI and x.is_cuda):
        return False
    cd = torch.get_autocast_dtype('cuda') if torch.is_autocast_enabled('cuda') else x.dtype
    rows = x.numel() // x.shape[-1]
    if sample_scale is not None and (rows % sample_scale.numel() or sample_scale.dtype != torch.float32):
        return False
    return weight.shape[0] == 256 and _edge_kernel_ok(x.reshape(-1, x.shape[-1]), 256, cd) and \
        edge_linear_supported(x.shape[-1], 256, cd, _lib.EPI_GELU, row_scale=sample_scale is not None)


def linear_gelu_dropout(x, weight, bias, p, training, sample_scale=None):
    """dropout(gelu(linear(x, weight, bias)), p) [* sample_scale per graph] in one launch; see linear_gelu_dropout_ok"""
    cd = torch.get_autocast_dtype('cuda') if torch.is_autocast_enabled('cuda') else x.dtype
    p = float(p) if training else 0.0
    seed = _host_seed() if p > 0 else 0
    act, pre = _LinearGeluDropout.apply(x, weight, bias, cd, p, seed, sample_scale, _lazy_ok(x, weight, cd))
    # rides on the tensor object (as _tgt_colsum does): what a consumer needs to take over the activation's backward
    act._tgt_gelu = (pre, p, seed, sample_scale)
    return act


# ---------------------------------------------------------------------------
# residual entry of a pre-norm block: s = res + x*scale ; y = LN(s)   (one pass each way)
# ---------------------------------------------------------------------------
class _AddLayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res, scale, weight, bias, eps, out_dtype):
        _dev(x, res, weight, bias)
        x, res = x.contiguous(), res.contiguous()
        C_ = x.shape[-1]
        rows = x.numel() // C_
        w, b = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        s = torch.empty_like(x)
        y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        rps = rows // x.shape[0]
        L = _lib.lib()
        p0, p1 = _prof_begin()
        _lib.check(L.tgt_add_layer_norm_fwd(_ptr(x), _DT[x.dtype], _ptr(res), _DT[res.dtype], _ptr(scale), rps, _ptr(s),
                                            _ptr(w), _ptr(b), _ptr(y), _DT[out_dtype], _ptr(mean), _ptr(rstd),
                                            rows, C_, float(eps), _stream()), 'tgt_add_layer_norm_fwd')
        _prof_end('tgt_add_layer_norm_fwd', p0, p1)
        ctx.save_for_backward(s, w, mean, rstd, scale)
        ctx.meta = (weight.dtype, res.dtype, rps)
        return s, y

    @staticmethod
    def backward(ctx, ds, dy):
        s, w, mean, rstd, scale = ctx.saved_tensors
        wdt, rdt, rps = ctx.meta
        C_ = s.shape[-1]
        rows = s.numel() // C_
        L = _lib.lib()
        if dy is None:                                  # the LN branch was not used
            d_res = ds
            d_x = ds if scale is None else ds * scale.view(-1, *([1] * (ds.ndim - 1))).to(ds.dtype)
            return d_x, d_res.to(rdt), None, None, None, None, None
        # (column sums of d_x: the bias gradient of the Linear that produced x (lin_O / lin_W2 / lin_O_e), handed over on the
        # gradient tensor, see _hand_colsum)
        d_res, d_x, dg, dbeta, cs = _ln_backward(dy, s, w, mean, rstd, ds, scale, rps, scale is not None)
        if d_x is None:
            d_x = d_res
        _hand_colsum(d_x, cs)
        return d_x, (d_res if rdt == d_res.dtype else d_res.to(rdt)), None, _param_grad(dg, wdt), _param_grad(dbeta, wdt), None, None


class _LinearResidualLN(torch.autograd.Function):
    """s = res + scale * (x W^T + b);  y = LayerNorm(s): the Linear that closes a sub-block (lin_O_e, lin_W2), the
    residual add (+ DropPath) and the LayerNorm that opens the next sub-block (reference layers.py:270-290) in ONE
    launch (tgt_edge_linear, TGT_EPI_RESID with the LayerNorm epilogue) instead of GEMM -> x -> add+LN pass.
    Backward: the add+LN backward kernel (it also yields the bias gradient), then the Linear's gradients."""

    @staticmethod
    def forward(ctx, x, weight, bias, res, scale, ln_w, ln_b, eps, cd, out_dtype, prescaled=False, pre=None, gelu=None, col_perm=None):
        # pre / gelu = (p, seed, sample_scale): x is dropout(gelu(pre)) * sample_scale, handed in DETACHED; this node then owns the
        # activation's derivative and returns the gradient of `pre` (GELU backward as the epilogue of the data-gradient GEMM)
        _dev(x, weight, res, ln_w, ln_b)
        xs = x.shape
        x2 = x.reshape(-1, xs[-1])
        if x2.dtype != cd:
            x2 = x2.to(cd)
        if col_perm is not None:
            # the weight's input columns in the order the producing kernel emits its channels (lin_O of the triplet modules:
            # one launch, cast included); the weight gradient goes back through the inverse permutation
            w = _permute_cols(_as_dtype_view(weight, cd).contiguous(), col_perm[0], cd)
        else:
            w = _as_dtype(weight, cd).contiguous()
        ctx.col_inv = None if col_perm is None else col_perm[1]
        b = None if bias is None else _as_dtype(bias, cd).contiguous()
        N, rows = weight.shape[0], x2.shape[0]
        res2 = res.reshape(rows, N)
        if res2.dtype != cd:
            res2 = res2.to(cd)
        g, be = ln_w.detach().float().contiguous(), ln_b.detach().float().contiguous()
        s = torch.empty(*xs[:-1], N, dtype=cd, device=x.device)
        y = torch.empty(*xs[:-1], N, dtype=out_dtype, device=x.device)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        rps = rows // xs[0]
        edge_linear_raw(x2, w, b, _lib.EPI_RESID, out=s.view(rows, N), res=res2, row_scale=scale, rows_per_sample=rps,
                        ln=(g, be, eps), stats=(mean, rstd), y=y.view(rows, N),
                        flags=_lib.EDGE_BIAS_SCALED if (prescaled and scale is not None) else 0)
        ctx.gelu = None
        if pre is not None:
            ctx.gelu = (float(gelu[0]), gelu[1], gelu[2] is not None)
            ctx.save_for_backward(x2, w, s, g, mean, rstd, scale, pre, gelu[2])
        else:
            ctx.save_for_backward(x2, w, s, g, mean, rstd, scale)
        ctx.meta = (xs, x.dtype, weight.dtype, None if bias is None else bias.dtype, ln_w.dtype, res.dtype, rps)
        ctx.prescaled = bool(prescaled and scale is not None)
        ctx.wptr = weight.data_ptr()
        return s, y

    @staticmethod
    def backward(ctx, ds, dy):
        x2, w, s, g, mean, rstd, scale = ctx.saved_tensors[:7]
        xs, xdt, wdt, bdt, lndt, rdt, rps = ctx.meta
        N = s.shape[-1]
        rows = s.numel() // N
        L = _lib.lib()
        pres = ctx.prescaled       # x arrived pre-scaled: the gradient of x W^T IS the stream gradient, only the bias carries the factor
        if dy is None:                                  # the LayerNorm branch was not used
            d_res = ds
            d_z = ds if (scale is None or pres) else ds * scale.view(-1, *([1] * (ds.ndim - 1))).to(ds.dtype)
            dg = dbeta = None
            cs = None
            if pres and bdt is not None and ctx.needs_input_grad[2]:
                cs = (ds.reshape(scale.numel(), -1, N).float().sum(1) * scale.view(-1, 1)).sum(0)
        else:
            d_res, d_z, dg, dbeta, cs = _ln_backward(dy, s, g, mean, rstd, ds, scale, rps, scale is not None and not pres)
            if d_z is None:
                d_z = d_res

        need_db = bdt is not None and ctx.needs_input_grad[2]
        d_pre = None
        fused_dw = None
        need_dx = ctx.needs_input_grad[0]
        if ctx.gelu is not None:
            # d_pre = ((d_z W) * sample_scale) * gelu'(pre) * keep / (1 - p): the data gradient and the activation's backward in
            # one launch (TGT_EPI_GELU_BWD); x was handed in detached, its slot gets no gradient
            pre, g_scale = ctx.saved_tensors[7], (ctx.saved_tensors[8] if ctx.gelu[2] else None)
            need_dx = False
            if ctx.needs_input_grad[11]:
                d_pre = torch.empty_like(pre)
                # (256 outputs = the row-phase kernel: it also returns the column sums of d_pre, lin_W1's bias gradient)
                # the weight gradient of THIS Linear rides on the same launch (TGT_EDGE_WGRAD): dW = d_z^T x with x = the activation
                # recomputed from `pre` in the row phase -- neither d_z nor x is read again by a weight-gradient GEMM
                fuse_dw = (_EDGE_WGRAD and ctx.needs_input_grad[1] and ctx.col_inv is None and N == 256 and pre.shape[-1] == 256 and
                           x2.shape[1] == 256 and wdt == torch.float32)
                parts = _lib.lib().tgt_edge_linear_parts(rows, 256)
                if fuse_dw and parts > 2 * _EDGE_WGRAD_SPARE > 0:
                    parts -= _EDGE_WGRAD_SPARE          # (the fused launch owns every CU it runs on -- all 160 KB of LDS: leave some to the node side stream)
                part = torch.empty(parts, 256, dtype=torch.float32, device=pre.device) \
                    if (pre.shape[-1] == 256 and _GELU_BWD_EPI_COLSUM) else None
                dw_part = torch.empty(parts, 256, 256, dtype=torch.float32, device=pre.device) if fuse_dw else None
                edge_linear_raw(d_z.reshape(rows, N), weight_t(w), None, _lib.EPI_GELU_BWD, out=d_pre.view(rows, -1),
                                res=pre.view(rows, -1), out_scale=g_scale, rows_per_sample=rps, dropout=(ctx.gelu[0], ctx.gelu[1]),
                                colsum_partial=part, dw_partial=dw_part)
                if part is not None:
                    with _on_stream(_terminal_fork(rows, part)):
                        _hand_colsum(d_pre, sum_rows(part))
                if dw_part is not None:
                    dst = _grad_dst(ctx.wptr, (256, 256), torch.float32)
                    with _on_stream(_terminal_fork(rows, dw_part)):
                        fused_dw = sum_planes(dw_part, dst if dst is not None else torch.empty(256, 256, dtype=torch.float32, device=pre.device))
        dx, dw, db = _linear_backward(x2, w, d_z.reshape(rows, N), xs, xdt, torch.float32 if ctx.col_inv is not None else wdt, bdt,
                                      need_dx, ctx.needs_input_grad[1] and fused_dw is None, need_db and cs is None,
                                      dw_post=None if ctx.col_inv is None else (lambda t: _permute_cols(t.contiguous(), ctx.col_inv, wdt, after_sums=True)),
                                      dw_ptr=ctx.wptr)
        if need_db and cs is not None:
            db = _param_grad(cs, bdt)
        if fused_dw is not None:
            dw = fused_dw
        return (dx, dw, db, d_res if rdt == d_res.dtype else d_res.to(rdt), None,
                _param_grad(dg, lndt), _param_grad(dbeta, lndt), None, None, None, None, d_pre, None, None)


_GELU_BWD_EPI_COLSUM = K.gelu_bwd_epi_colsum      # A/B knob: lin_W1's bias gradient from the GELU_BWD epilogue
_EDGE_WGRAD_SPARE = K.edge_wgrad_spare_cus
_EDGE_WGRAD = K.edge_wgrad      # A/B knob: weight gradients of the 256 x 256 edge Linears inside their data-gradient launches (csrc/edge_wgrad.hip)
_EDGE_K512 = K.edge_k512        # A/B knob: lin_O (K = 512) + residual + LayerNorm as one launch


def _residual_fusable(x, weight, res, cd):
    N = weight.shape[0]
    x2 = x.reshape(-1, x.shape[-1])
    ks = (64, 128, 256, 512) if (N == 256 and _EDGE_K512) else (64, 128, 256)
    return N <= 256 and res.is_cuda and _edge_kernel_ok(x2, N, cd, ks) and res.dtype in (cd,) and res.is_contiguous()


_PRESCALE = K.prescale       # A/B knob: DropPath factor folded into the branch's producer


def can_prescale(rows, in_features, out_features, dtype):
    """Will linear_residual_layer_norm(prescaled=True) on (rows, in_features) -> out_features rows of `dtype` (the compute
    dtype) run as the ONE fused launch that implements it?  (The callers decide before they run the producer.)"""
    return (_PRESCALE and _EDGE_GEMM and dtype in (torch.bfloat16, torch.float16) and out_features == 256 and
            in_features in (64, 128, 256) and rows >= _EDGE_MIN_ROWS)


def linear_residual_layer_norm(x, weight, bias, res, scale, ln_weight, ln_bias, eps=1e-5, prescaled=False, col_perm=None):
    """(s, y): s = res + scale[graph] * linear(x, weight, bias) (scale: per-sample DropPath factors or None),
    y = LayerNorm(s).  One fused launch on the MI355X slice kernel when the shape qualifies (16-bit compute dtype,
    in_features in {64,128,256}, out_features <= 256 and a multiple of 8, >= 65536 rows); otherwise the composition of
    ops.linear and ops.add_layer_norm (same arithmetic, two launches + one more pass over the rows).
    col_perm = (idx, inv) int32 device tensors: the Linear is linear(x, weight[:, idx], bias) (lin_O of the triplet modules, whose
    input arrives in the kernels' channel order; in_features 512 runs on the K = 512 form of the fused launch).
    prescaled: x already carries the factor (its producer folded it in: gelu_dropout(sample_scale=), node_attention(
    hhat_scale=)), so s = res + x W^T + scale[graph] * bias -- the same value, and the backward hands the stream gradient
    itself to the Linear's gradient GEMMs instead of writing a scaled copy of it (one pass over the rows less)."""
    cd = torch.get_autocast_dtype('cuda') if (x.is_cuda and torch.is_autocast_enabled('cuda')) else x.dtype
    # x = the FFN's activation out of linear_gelu_dropout: take over its backward when the data gradient runs on the row-phase kernel
    gelu = getattr(x, '_tgt_gelu', None) if _FFN_GELU_BWD_EPI else None
    if gelu is not None and not (weight.shape[0] == 256 and weight.shape[1] == 256 and gelu[0].dtype == cd and
                                 _residual_fusable(x, weight, res, cd) and edge_linear_supported(256, 256, cd, _lib.EPI_GELU_BWD)):
        gelu = None
    extra = () if gelu is None else (gelu[0], gelu[1:])
    if gelu is not None:
        x = x.detach()
    if col_perm is not None:
        if _residual_fusable(x, weight, res, cd) and weight.shape[0] == 256 and not prescaled:
            return _offer_lazy(*_LinearResidualLN.apply(x, weight, bias, res, scale, ln_weight, ln_bias, eps, cd, cd, False, None, None, col_perm))
        return add_layer_norm(linear_permuted_cols(x, weight, bias, *col_perm), res, scale, ln_weight, ln_bias, eps)
    if prescaled and scale is not None:
        if _residual_fusable(x, weight, res, cd) and weight.shape[0] == 256 and weight.shape[1] <= 256:
            return _offer_lazy(*_LinearResidualLN.apply(x, weight, bias, res, scale, ln_weight, ln_bias, eps, cd, cd, True, *extra))
        # composition with the same arithmetic (shapes the fused launch does not take)
        z = linear(x, weight, None)
        if bias is not None:
            z = z + scale.view([-1] + [1] * (z.ndim - 1)).to(z.dtype) * bias.to(z.dtype)
        return add_layer_norm(z, res, None, ln_weight, ln_bias, eps)
    if _residual_fusable(x, weight, res, cd):
        return _offer_lazy(*_LinearResidualLN.apply(x, weight, bias, res, scale, ln_weight, ln_bias, eps, cd, cd, False, *extra))
    return add_layer_norm(linear(x, weight, bias), res, scale, ln_weight, ln_bias, eps)


_side_streams = {}
_main_streams = {}          # the stream the side stream was last forked from, per device
# Tensors that cross between the step's stream and the node side stream are handed to Tensor.record_stream: the caching allocator
# then holds each freed block until an EVENT on the other stream has completed; the host runs a step ahead of the GPU, so the next
# step's forward asks for some of those blocks before their events are done and the allocator answers with hipMalloc -- 0 to 2
# device allocations per step in steady state, all on the side stream (tools/probes/alloc_dbg.py: 18 segments of 12 / 36 / 54 MB
# in 12 steps; bench.py reports them as memory.device_allocs_in_timed_region).  TGT_STREAM_KEEPALIVE=1 removes them: under a
# Trainer the tensors are kept referenced until the step's two streams have been joined BOTH ways (release_stream_keepalive,
# once per step after the gradient collection), so a freed block goes straight back to the pool of the stream that allocated it.
# Measured (profiles/r05j_ab_keepalive.txt, same box, alternating, 30 steps): device allocations 0 / 0 against 73 / 0, but the
# per-step median is 83.53 / 83.48 ms against 83.12 / 83.06 and single steps stall (max 110 / 116 ms against 96 / 85) -- the
# allocations cost less than holding every cross-stream activation to the end of the step.  Opt-in.
_KEEPALIVE = K.stream_keepalive       # A/B knob (default: Tensor.record_stream)
_stream_keepalive = []


def _cross_stream(t, other):
    """tensor t (allocated on the current allocation stream) is also used on stream `other`"""
    if _KEEPALIVE and side_stream._owners > 0:
        if len(_stream_keepalive) >= 4096:          # (forward-only use under a Trainer: nobody else releases them)
            release_stream_keepalive()
        _stream_keepalive.append(t)
    else:
        t.record_stream(other)


def release_stream_keepalive(device=None):
    """join the step's stream and the node side stream both ways, then drop the references held for cross-stream tensors"""
    if not _stream_keepalive:
        return
    for dev, side in _side_streams.items():
        if device is None or dev == device:
            cur = torch.cuda.current_stream(dev)
            if side != cur:
                cur.wait_stream(side)
                side.wait_stream(cur)
    _stream_keepalive.clear()


class side_stream:
    """`with ops.side_stream(t):` runs the block on a second HIP stream that first waits for
    the work queued so far on the current one; `.join()` makes the current stream wait for the
    block.  Used to run the small node-channel kernels (8192 rows: latency-bound, a few
    workgroups) under the edge-channel kernels of the same layer.  Autograd replays the block's
    backward on the same side stream and inserts the cross-stream waits itself.
    TGT_NODE_STREAM=0 disables it (everything stays on the current stream).

    It is only used while a tgt_amd Trainer owns the step (`owner_present`): the Trainer's gradient
    collection waits for both streams (wait_side_streams) before it reads or all-reduces gradients.
    Anything else that consumes parameter gradients from hooks -- torch DistributedDataParallel's
    reducer when these modules are aliased into the reference's run_training.py -- only orders its
    collectives after the stream of the hook that closed a bucket, and would race with gradients still
    being produced on the side stream; without an owner the block therefore runs on the current stream."""
    enabled = K.node_stream
    _owners = 0

    @classmethod
    def owner_present(cls, present):
        cls._owners = max(0, cls._owners + (1 if present else -1))

    def __init__(self, *inputs):
        self.active = self.enabled and self._owners > 0 and len(inputs) > 0 and all(t.is_cuda for t in inputs)
        if self.active:
            dev = inputs[0].device
            self.main = torch.cuda.current_stream(dev)
            _main_streams[dev] = self.main
            if dev not in _side_streams:
                _side_streams[dev] = torch.cuda.Stream(dev, priority=_SIDE_PRIO)
            self.side = _side_streams[dev]
            self.inputs = inputs

    def __enter__(self):
        if self.active:
            self.side.wait_stream(self.main)
            for t in self.inputs:               # the allocator must not recycle them under the side stream
                _cross_stream(t, self.side)
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.active:
            self.ctx.__exit__(*exc)
        return False

    def resume(self):
        """context manager: more work on the side stream, ordered after what is already on it (no
        new dependency on the current stream)"""
        return torch.cuda.stream(self.side) if self.active else contextlib.nullcontext()

    def join(self, *outputs):
        if self.active:
            torch.cuda.current_stream(self.side.device).wait_stream(self.side)
            for t in outputs:
                _cross_stream(t, torch.cuda.current_stream(self.side.device))


def wait_side_streams(device=None):
    """make the current stream wait for everything queued on the side stream AND on the stream it
    was forked from -- before reading tensors (e.g. gradients inside an autograd hook, which may
    itself be running on either of the two) that blocks on both streams have produced"""
    for dev in set(_side_streams) | set(_wgrad_streams):
        if device is None or dev == device:
            cur = torch.cuda.current_stream(dev)
            for other in (_side_streams.get(dev), _wgrad_streams.get(dev), _main_streams.get(dev)):
                if other is not None and other != cur:
                    cur.wait_stream(other)
            q = _wgrad_window.get(dev)
            if q:
                # everything forked so far is behind `cur` now; operands that came from `cur` may go (the others wait for their turn)
                keep = [e for e in q if e[2] != cur or e[0] is None]      # (an open fork may still be given work)
                q.clear()
                q.extend(keep)


def drop_path_scale(x, drop_prob, training):
    """per-sample DropPath factor (B,) float32 = Bernoulli(keep)/keep, or None when inactive
    (reference lib/tgt/layers/layers.py:169-174)."""
    if drop_prob > 0 and training:
        return _scale_pool.take(x.size(0), 1.0 - drop_prob, x.device)
    return None


def add_layer_norm(x, res, scale, weight, bias, eps=1e-5, out_dtype=None):
    """(s, y) with s = res + x*scale (per-sample scale or None), y = LayerNorm(s)."""
    if out_dtype is None:
        out_dtype = torch.get_autocast_dtype('cuda') if torch.is_autocast_enabled('cuda') else x.dtype
    return _offer_lazy(*_AddLayerNorm.apply(x, res, scale, weight, bias, eps, out_dtype))


# ---------------------------------------------------------------------------
# loss head
# ---------------------------------------------------------------------------
class _CrossEntropyRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        _dev(logits, target)
        if logits.dim() != 2 or not logits.is_contiguous() or target.shape != logits.shape[:1] or target.dtype != torch.int64:
            raise RuntimeError('cross_entropy_rows: contiguous (rows, C) logits and int64 (rows,) targets expected')
        target = target.contiguous()
        rows, C_ = logits.shape
        lse = torch.empty(rows, dtype=torch.float32, device=logits.device)
        xent = torch.empty(rows, dtype=torch.float32, device=logits.device)
        _lib.check(_lib.lib().tgt_cross_entropy_fwd(_ptr(logits), _DT[logits.dtype], _ptr(target), rows, C_, _ptr(lse), _ptr(xent),
                                                    _stream()), 'tgt_cross_entropy_fwd')
        ctx.save_for_backward(logits, target, lse)
        return xent

    @staticmethod
    def backward(ctx, g):
        logits, target, lse = ctx.saved_tensors
        g = g.contiguous().float()
        d = torch.empty_like(logits)
        _lib.check(_lib.lib().tgt_cross_entropy_bwd(_ptr(logits), _DT[logits.dtype], _ptr(target), _ptr(lse), _ptr(g), logits.shape[0],
                                                    logits.shape[1], _ptr(d), _stream()), 'tgt_cross_entropy_bwd')
        return d, None


def cross_entropy_rows(logits, target):
    """F.cross_entropy(logits, target, reduction='none') for (rows, C) logits in their storage dtype ->
    float32 (rows,), without an fp32 image of the logits (reference commons.py:36-38)."""
    return _CrossEntropyRows.apply(logits, target)


# ---------------------------------------------------------------------------
# Gaussian basis of the 3-D distance embedding (reference lib/models/pcqm/layers.py:129-157)
# ---------------------------------------------------------------------------
class _GaussianBasis(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mul, bias, means, stds, out_dtype):
        _dev(x, mul, bias, means, stds)
        K = means.numel()
        xf, mf, bf = (t.detach().reshape(-1).float().contiguous() for t in (x, mul, bias))
        mw, sw = means.detach().reshape(-1).float().contiguous(), stds.detach().reshape(-1).float().contiguous()
        y = torch.empty(*x.shape, K, dtype=out_dtype, device=x.device)
        _lib.check(_lib.lib().tgt_gaussian_basis_fwd(_ptr(xf), _ptr(mf), _ptr(bf), _ptr(mw), _ptr(sw), xf.numel(), K, _DT[out_dtype],
                                                     _ptr(y), _stream()), 'tgt_gaussian_basis_fwd')
        ctx.save_for_backward(xf, mf, bf, mw, sw)
        ctx.meta = (x.shape, mul.shape, bias.shape, means.shape, stds.shape, means.dtype, out_dtype)
        return y

    @staticmethod
    def backward(ctx, g):
        xf, mf, bf, mw, sw = ctx.saved_tensors
        xs, ms, bs, mws, sws, pdt, out_dtype = ctx.meta
        K, pairs = mw.numel(), xf.numel()
        g = g.contiguous()
        if g.dtype != out_dtype:
            g = g.to(out_dtype)
        L = _lib.lib()
        dt = torch.empty(pairs, dtype=torch.float32, device=g.device)
        partial = torch.empty(L.tgt_gaussian_basis_parts(pairs), 2 * K, dtype=torch.float32, device=g.device)
        _lib.check(L.tgt_gaussian_basis_bwd(_ptr(xf), _ptr(mf), _ptr(bf), _ptr(mw), _ptr(sw), pairs, K, _DT[out_dtype], _ptr(g), _ptr(dt),
                                            _ptr(partial), _stream()), 'tgt_gaussian_basis_bwd')
        dms = sum_rows(partial, defer=False)          # (sliced and cast right below)
        dx = (dt * mf).view(xs) if ctx.needs_input_grad[0] else None
        return dx, (dt * xf).view(ms), dt.view(bs), dms[:K].view(mws).to(pdt), dms[K:].view(sws).to(pdt), None


def gaussian_basis(x, mul, bias, means, stds):
    """y[..., k] = exp(-((mul*x + bias - means[k]) / (|stds[k]| + 0.01))^2 / 2) / ((2*3.14159)^0.5 (|stds[k]| + 0.01)); x, mul, bias
    broadcast-free tensors of one shape (mul / bias may carry a trailing 1).  One HIP pass each way; the result comes out
    in the autocast dtype when autocast is on (the reference's float32 result is cast to it by the Linear that follows)."""
    cd = torch.get_autocast_dtype('cuda') if (x.is_cuda and torch.is_autocast_enabled('cuda')) else means.dtype
    return _GaussianBasis.apply(x, mul, bias, means, stds, cd)
