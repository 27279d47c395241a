#!/usr/bin/env python
"""Where the cycles of `tgt_triplet_attention_bwd` go (BASELINE shape: B=256, N=32, C=256, Ht=16, bf16, column sums on).

Needs the PROBE build of the library (triplet_attention.hip compiled with -DTGT_PROBES: s_memtime stamps around the
segments of the j-loop in wave 0 of every workgroup, and ablation bits), loaded instead of the shipped one:

    TGT_HIP_LIB=tools/probes/libtgt_hip_probe.so python tools/probes/tri_bwd_probe.py

Prints, per ablation setting (TGT_TRI_BWD_ABLATE: 1 no loads, 2 no stores, 4 no tile math), the kernel time and the mean
cycles per j of each segment.  Results with an ablation bit set are WRONG by construction; this is a timing probe only.
"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tgt_amd import _lib, ops  # noqa: E402

SEG = ['loop', 'commit(wait+4 ds_write)', 'prefetch issue', 'tile math', 'barrier', 'stores(+colsum)', '-', 'kernel total']


def main():
    B, N, Cc, Ht = 256, 32, 256, 16
    dt = torch.bfloat16
    dev = 'cuda'
    torch.manual_seed(0)
    L = ops.TripletLayout(Cc, Ht)
    mask = torch.zeros(B, N, N, device=dev)
    x = torch.randn(B, N, N, Cc, device=dev, dtype=dt).requires_grad_(True)
    w = (torch.randn(L.width, Cc, device=dev) * Cc ** -0.5).to(dt).requires_grad_(True)
    bias = torch.randn(L.width, device=dev).to(dt).requires_grad_(True)
    g = torch.randn(B, N, N, 2 * Cc, device=dev, dtype=dt)
    lib = _lib.lib()
    has_probe = hasattr(lib, 'tgt_probe_read')
    out = {}
    for ab in [int(v) for v in os.environ.get('TGT_PROBE_SET', '0,0,1,2,3,4,7').split(',')]:
        os.environ['TGT_TRI_BWD_ABLATE'] = str(ab)
        prof = ops.profile_kernels(True)
        for _ in range(8):
            torch.autograd.grad(ops.projected_triplet_attention(x, w, bias, mask, L), (x, w, bias), g)
        torch.cuda.synchronize()
        ops.profile_kernels(False)
        t = ops.kernel_times_ms(prof)['tgt_triplet_attention_bwd']
        rec = {'ms': round(sum(t[2:]) / len(t[2:]), 4)}
        if has_probe:
            buf = np.zeros(1024 * 8, dtype=np.uint64)
            lib.tgt_probe_read.restype = C.c_int
            rc = lib.tgt_probe_read(buf.ctypes.data_as(C.c_void_p), C.c_int(buf.size))
            assert rc == 0, rc
            p = buf.reshape(1024, 8).astype(np.float64)
            per_j = p[:, :6].mean(0) / N
            rec['cycles_per_j'] = {SEG[i]: round(float(per_j[i]), 1) for i in range(6)}
            rec['sum_per_j'] = round(float(per_j.sum()), 1)
            rec['kernel_cycles_per_wg'] = round(float(p[:, 7].mean()), 0)
            rec['kernel_cycles_per_wg_minmax'] = [float(p[:, 7].min()), float(p[:, 7].max())]
        out[f'ablate={ab}'] = rec   # (the first setting also warms the box up: listed twice by default)
        print(f'ablate={ab}', json.dumps(rec), flush=True)
    if not has_probe:
        print('NOTE: not the probe build (tgt_probe_read missing): times only')
    return out


if __name__ == '__main__':
    main()
