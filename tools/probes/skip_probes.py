"""Timing probes that LEAVE WORK OUT of the training step (probe-only: nothing under tgt_amd/ knows about them).

`python bench.py --timing-probe skip_wgrad` installs one of these on top of tgt_amd.ops before the model is built; bench.py then
prints `"value": null` with `"invalid": "timing probe ..."` next to the measured milliseconds, so that a number produced with work
missing can never be read as a throughput (these used to be environment switches inside tgt_amd/ops.py).

  skip_sums     every tgt_sum_planes closing sum returns without a launch (what any scheme that moves the closing sums can win)
  skip_proj_ln  the projection's standalone LayerNorm backward replaced by a 2 E copy (what a fused epilogue would save)
  skip_wgrad    every split-M weight gradient left out, ZERO gradients (what a wgrad riding on the dgrad kernels could win at most)

The gradients of a probed step are wrong by construction; Adam still runs on them (timing only).
"""
import torch

PROBES = ('skip_sums', 'skip_proj_ln', 'skip_wgrad')


def install(name):
    from tgt_amd import ops, _lib
    if name == 'skip_sums':
        real = ops.sum_planes

        def sum_planes(part, out, defer=True):
            ops._dev(part, out)
            ops._check_planes(part, out)
            return out
        ops.sum_planes = sum_planes
        return real
    if name == 'skip_proj_ln':
        real = ops._ln_backward

        def _ln_backward(dy, s, g, mean, rstd, ds, scale, rps, want_dz):
            N = s.shape[-1]
            rows = s.numel() // N
            if rows < 65536 or ops._take_lazy_dgrad(dy) is not None:
                return real(dy, s, g, mean, rstd, ds, scale, rps, want_dz)
            d_res = torch.empty_like(s)
            d_res.copy_(dy.view_as(d_res))                 # a plain copy (2 E) in place of the 4 E pass keeps the gradients finite
            zeros = torch.zeros(3 * N, dtype=torch.float32, device=s.device)
            return d_res, (torch.zeros_like(s) if want_dz else None), zeros[:N], zeros[N:2 * N], zeros[2 * N:]
        ops._ln_backward = _ln_backward
        return real
    if name == 'skip_wgrad':
        real_into, real_bmm = ops._wgrad_into, torch.bmm

        def _wgrad_into(out, dy2, x2, chunks):
            if x2.shape[0] >= 65536:
                return out.zero_()
            return real_into(out, dy2, x2, chunks)
        ops._wgrad_into = _wgrad_into

        # _linear_backward issues its split-M product itself (torch.bmm over (P, in, rows/P) x (P, rows/P, out) row chunks, fp32
        # partials): the probe answers that one call shape with zeros
        def bmm(a, b, out_dtype=None, **kw):
            if a.dim() == 3 and a.shape[0] > 1 and a.shape[0] * a.shape[2] >= 65536 and out_dtype == torch.float32:
                return torch.zeros(a.shape[0], a.shape[1], b.shape[2], dtype=torch.float32, device=a.device)
            return real_bmm(a, b, **({'out_dtype': out_dtype} if out_dtype is not None else {}), **kw)
        ops.torch = _TorchProxy(torch, bmm)
        return real_into
    raise SystemExit(f'unknown timing probe {name!r}: one of {PROBES}')


class _TorchProxy:
    """`torch` as tgt_amd.ops sees it, with bmm replaced (the probe must not touch the global torch module)"""

    def __init__(self, mod, bmm):
        self._mod, self.bmm = mod, bmm

    def __getattr__(self, k):
        return getattr(self._mod, k)
