"""Two-queue timeline of ONE backward (or forward) layer out of a rocprofv3 --kernel-trace CSV of the default (two-stream) bench.py
run: every kernel of both HIP queues between two consecutive triplet kernels, in start order, with the idle gaps of the queue
that runs the triplet kernel (the step's own stream).  python tools/timeline2.py trace.csv [bwd|fwd] [index from the end]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
which = sys.argv[2] if len(sys.argv) > 2 else 'bwd'
back = int(sys.argv[3]) if len(sys.argv) > 3 else 14
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], r['Kernel_Name']) for r in rows))
key = 'tri_att_bwd' if which == 'bwd' else 'tri_att_proj_fwd'
idx = [i for i, e in enumerate(ev) if key in e[3]]
a, b = idx[-back], idx[-back + 1]
mainq = ev[a][2]
t0 = ev[a][0]
last_end = None
idle = 0.0
busy = {}
for s, e, q, n in ev[a:b]:
    d = (e - s) / 1e3
    busy[q] = busy.get(q, 0.0) + d
    gap = ''
    if q == mainq:
        if last_end is not None and s - last_end > 3000:
            gap = f'   <-- main queue idle {(s - last_end) / 1e3:.1f} us'
            idle += (s - last_end) / 1e3
        last_end = e if last_end is None else max(last_end, e)
    short = n.split('(')[0]
    short = short[-64:] if len(short) > 64 else short
    print(f"{(s - t0) / 1e3:9.1f} {d:8.1f}us {'M' if q == mainq else ' s'} {short}{gap}")
print(f'layer wall {(ev[b][0] - t0) / 1e3:.0f} us; busy per queue {({("main" if q == mainq else "side"): round(v) for q, v in busy.items()})}; main-queue idle in gaps > 3 us: {idle:.0f} us')
