#!/usr/bin/env python
"""Per-kernel statistics of the STEADY-STATE training steps of a rocprofv3 kernel trace.

`rocprofv3 --kernel-trace --stats` summarises the whole process, warm-up (allocator growth, TunableOp's online
GEMM tuning of unseen shapes) included.  This reads the `*_kernel_trace.csv` of the same run, cuts it at the
one-per-step Adam launch (`adam_kernel`) and keeps the last K complete steps.

  python tools/trace_summary.py /tmp/prof/r02_kernel_trace.csv --steps 5 > profiles/r02_bench_kernel_stats.csv
"""
import argparse
import csv
import sys
from collections import defaultdict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('trace')
    ap.add_argument('--steps', type=int, default=5, help='complete steps kept (counted back from the last Adam launch)')
    ap.add_argument('--delimiter', default='adam', help='substring of the kernel that ends a step')
    ap.add_argument('--top', type=int, default=0, help='also print the top N rows as a table on stderr')
    args = ap.parse_args()
    rows = []
    with open(args.trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if args.delimiter in r[2]]
    if len(ends) < args.steps + 1:
        raise SystemExit(f'only {len(ends)} step delimiters in the trace')
    lo, hi = ends[-args.steps - 1] + 1, ends[-1] + 1
    sel = rows[lo:hi]
    agg = defaultdict(list)
    for s, e, n in sel:
        agg[n].append(e - s)
    total = sum(sum(v) for v in agg.values())
    wall = sel[-1][1] - sel[0][0]
    w = csv.writer(sys.stdout)
    w.writerow(['Name', 'Calls', 'CallsPerStep', 'TotalDurationNs', 'AverageNs', 'MsPerStep', 'Percentage', 'MinNs', 'MaxNs'])
    table = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
    for n, v in table:
        w.writerow([n, len(v), round(len(v) / args.steps, 2), sum(v), round(sum(v) / len(v), 1),
                    round(sum(v) / 1e6 / args.steps, 4), round(100.0 * sum(v) / total, 3), min(v), max(v)])
    print(f'# {args.steps} steps: {len(sel) / args.steps:.0f} launches/step, kernel time {total / 1e6 / args.steps:.2f} ms/step, '
          f'wall {wall / 1e6 / args.steps:.2f} ms/step', file=sys.stderr)
    for n, v in table[:args.top]:
        print(f'{n[:96]:96s} {len(v) / args.steps:7.1f} x {sum(v) / len(v) / 1e3:8.1f} us = {sum(v) / 1e6 / args.steps:7.2f} ms/step '
              f'{100.0 * sum(v) / total:5.1f}%', file=sys.stderr)


if __name__ == '__main__':
    main()
