"""One layer's forward / backward kernel sequence out of a rocprofv3 --kernel-trace CSV of bench.py
(the span between two consecutive triplet-attention kernels).  python tools/trace_layer.py trace.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
fw = [i for i, n in enumerate(names) if 'tri_att_fwd' in n or 'tri_att_proj_fwd' in n]
bw = [i for i, n in enumerate(names) if 'tri_att_bwd' in n]


def show(a, b, brief):
    t0 = int(rows[a]['Start_Timestamp'])
    busy, tiny, ntiny = 0.0, 0.0, 0
    for r in rows[a:b]:
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        busy += d
        if d < 12:
            tiny += d
            ntiny += 1
        if not brief or d >= 12:
            print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {d:8.1f}us q{r['Queue_Id']} {r['Kernel_Name'][:100]}")
    wall = (int(rows[b]['Start_Timestamp']) - t0) / 1e3
    print(f'kernels {b - a}  busy {busy:.0f}us  wall {wall:.0f}us  tiny(<12us) {ntiny} = {tiny:.0f}us')


brief = len(sys.argv) > 2
print('--- forward layer'); show(fw[-14], fw[-13], brief)
print('--- backward layer'); show(bw[-14], bw[-13], brief)
