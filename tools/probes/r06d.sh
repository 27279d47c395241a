set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06d; mkdir -p $O
export TMPDIR=/tmp
cd $R
ab() { env "$@" python bench.py --no-cpu-baseline --steps 25 --warmup 6 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$*', d['value'], d['ms_per_step'], 'median', d['step_ms']['median'], 'max', d['step_ms']['max'], 'dry', d['step_ms']['steps_stream_ran_dry'], 'lead', d['step_ms']['host_lead_steps'], 'host', d['step_ms']['host_enqueue_ms'])"; }
( ab TGT_EMBED_GEMM=0; ab TGT_EMBED_GEMM=1; ab TGT_EMBED_GEMM=0; ab TGT_EMBED_GEMM=1 ) > $O/ab_embed.txt 2>&1; grep -v "^+" $O/ab_embed.txt
for v in 0 1; do
rm -rf /tmp/ph
( cd /tmp && TGT_EMBED_GEMM=$v timeout 600 rocprofv3 --hip-trace --kernel-trace --output-format csv -d /tmp/ph -o bench -- python $R/bench.py --steps 4 --warmup 2 --settle-steps 4 --roofline-steps 0 --no-cpu-baseline ) > /tmp/ph.log 2>&1
f=$(find /tmp/ph -name "*hip_api_trace.csv" | head -1)
k=$(find /tmp/ph -name "*kernel_trace.csv" | head -1)
python - "$f" "$k" > $O/api_in_step_$v.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ds = [i for i, r in enumerate(rows) if r['Function'] == 'hipDeviceSynchronize']
print('hipDeviceSynchronize at rows', ds)
# timed region: between the 2nd and 3rd device synchronize (fence = sync, [barrier], sync)
a, b = ds[1], ds[2]
reg = rows[a:b]
t0 = int(reg[0]['Start_Timestamp'])
span = (int(reg[-1]['End_Timestamp']) - t0) / 1e6
print(f'timed region: {len(reg)} API calls in {span:.1f} ms (4 steps)')
c = collections.Counter(r['Function'] for r in reg)
tot = collections.Counter()
for r in reg: tot[r['Function']] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for k, v in c.most_common(30): print(f'{v:7d} {tot[k]/1e6:9.2f} ms  {k}')
print('--- every call > 200 us in the region')
for r in reg:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if d > 200: print(f"{(int(r['Start_Timestamp']) - t0) / 1e6:9.3f} ms {d:9.1f} us tid {r['Thread_Id']} {r['Function']}")
print('--- memcpy calls in the region')
for r in reg:
    if 'Memcpy' in r['Function']:
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e6:9.3f} ms {d:9.1f} us tid {r['Thread_Id']} {r['Function']}")
# kernels: idle gaps of the GPU (all queues merged) inside the region > 150 us
ks = list(csv.DictReader(open(sys.argv[2])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in ks)
t1 = int(reg[-1]['End_Timestamp'])
ev = [e for e in ev if t0 <= e[0] <= t1]
end = ev[0][1]; gaps = []
for s, e, n in ev[1:]:
    if s - end > 150_000: gaps.append(((s - t0) / 1e6, (s - end) / 1e3, n.split('(')[0][-50:]))
    end = max(end, e)
print('--- GPU idle gaps > 150 us (all queues merged):', len(gaps), 'total', sum(g[1] for g in gaps) / 1e3, 'ms')
for g in gaps[:60]: print(f'{g[0]:9.3f} ms  idle {g[1]:8.1f} us before {g[2]}')
PY
head -70 $O/api_in_step_$v.txt
done
