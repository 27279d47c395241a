#!/usr/bin/env python
"""What runs OUTSIDE the 24 layers in one steady-state step of a rocprofv3 kernel trace (single stream): the kernels from the
previous step's Adam launch to the first layer's first triplet kernel (input embedding), between the last forward layer and
the first backward layer (heads + loss and their backward), and after the last backward layer (embedding backward, gradient
collection, Adam).

  python tools/trace_edges.py /tmp/ps/..._kernel_trace.csv
"""
import csv
import sys


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if 'adam' in r[2]]
    lo, hi = ends[-2] + 1, ends[-1] + 1
    step = rows[lo:hi]
    fwd = [i for i, r in enumerate(step) if 'tri_att_proj_fwd' in r[2] or 'tri_att16_fwd' in r[2] or 'tri_att_fwd' in r[2]]
    bwd = [i for i, r in enumerate(step) if 'tri_att_bwd2' in r[2] or 'tri_att16_bwd' in r[2] or 'tri_att_bwd' in r[2]]
    per_layer = (fwd[-1] - fwd[0]) // max(1, len(fwd) - 1)
    per_layer_b = (bwd[-1] - bwd[0]) // max(1, len(bwd) - 1)
    segs = [('before the first layer', 0, fwd[0]), ('last forward layer .. first backward layer', fwd[-1], bwd[0]),
            ('last backward layer .. Adam', bwd[-1], len(step))]
    t0 = step[0][0]
    print(f'step: {len(step)} launches, {sum(e - s for s, e, _ in step) / 1e6:.2f} ms of kernels, wall {(step[-1][1] - t0) / 1e6:.2f} ms; '
          f'a forward layer ~{per_layer} launches, a backward layer ~{per_layer_b}')
    for name, a, b in segs:
        seg = step[a:b]
        busy = sum(e - s for s, e, _ in seg)
        print(f'--- {name}: {len(seg)} launches, busy {busy / 1e3:.0f} us, wall {(seg[-1][1] - seg[0][0]) / 1e3:.0f} us'
              + (' (includes one layer)' if 'last' in name else ''))
        for s, e, n in seg:
            if e - s >= 15000:
                print(f'  {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}us {n[:150]}')


if __name__ == '__main__':
    main()
