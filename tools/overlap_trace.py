"""Which kernels of the OTHER queue run beside a given kernel?  From a rocprofv3 --kernel-trace CSV of the two-stream bench.py run.

  python tools/overlap_trace.py kernel_trace.csv tri_att_bwd2 [--last 48]

For the last `--last` launches of the target: its mean duration, the mean time another queue's kernel was running beside it,
and the overlapping kernels by name (count, mean overlap in us).  Answers VERDICT r4 item 3's question -- what costs the
triplet backward its 15 % inside the timed region -- with timestamps instead of event pairs."""
import argparse
import collections
import csv

ap = argparse.ArgumentParser()
ap.add_argument('trace')
ap.add_argument('target')
ap.add_argument('--last', type=int, default=48)
args = ap.parse_args()

rows = list(csv.DictReader(open(args.trace)))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], r['Kernel_Name']) for r in rows))
tg = [e for e in ev if args.target in e[3]][-args.last:]
starts = [e[0] for e in ev]
import bisect
by_name = collections.defaultdict(lambda: [0, 0.0])
tot_dur, tot_ov, with_ov, dur_ov, dur_no = 0.0, 0.0, 0, [], []
for s, e, q, n in tg:
    lo = bisect.bisect_left(starts, s - 5_000_000)
    ov = 0.0
    for s2, e2, q2, n2 in ev[lo:]:
        if s2 >= e:
            break
        if q2 == q or e2 <= s:
            continue
        o = (min(e, e2) - max(s, s2)) / 1e3
        ov += o
        short = n2.split('(')[0][-70:]
        by_name[short][0] += 1
        by_name[short][1] += o
    d = (e - s) / 1e3
    tot_dur += d
    tot_ov += ov
    (dur_ov if ov > 10 else dur_no).append(d)
n = max(1, len(tg))
print(f'{args.target}: {len(tg)} launches, mean {tot_dur / n:.1f} us, mean overlap with other queues {tot_ov / n:.1f} us')
if dur_ov:
    print(f'  with > 10 us of overlap: {len(dur_ov)} launches, mean {sum(dur_ov) / len(dur_ov):.1f} us')
if dur_no:
    print(f'  without:                 {len(dur_no)} launches, mean {sum(dur_no) / len(dur_no):.1f} us')
for k, (c, o) in sorted(by_name.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f'  {c:5d} x {o / c:7.1f} us  {k}')
