set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06c; mkdir -p $O
export TMPDIR=/tmp
cd $R
show() { python -c "
import sys,json; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']
print('$1', d['value'], d['ms_per_step'], 'step_ms', d['step_ms'], 'bwd in-region', r['avg_launch_ms'], r['frac'])"; }
python bench.py --no-cpu-baseline --steps 25 --warmup 6 > $O/bench_a.json 2> $O/bench.err; show $O/bench_a.json
python bench.py --no-cpu-baseline --steps 25 --warmup 6 > $O/bench_b.json 2>> $O/bench.err; show $O/bench_b.json
timeout 900 python -m pytest tests/test_hip_model.py tests/test_hip_trainer.py -m gpu -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
rm -rf /tmp/ph
( cd /tmp && timeout 600 rocprofv3 --hip-trace --stats --output-format csv -d /tmp/ph -o bench -- python $R/bench.py --steps 4 --warmup 2 --settle-steps 4 --roofline-steps 0 --no-cpu-baseline ) > /tmp/ph.log 2>&1
ls -la /tmp/ph/* | head
cp $(find /tmp/ph -name "*hip_api_stats.csv" | head -1) $O/hip_api_stats.csv
cat $O/hip_api_stats.csv | head -40
# sync-like calls with timestamps: the last 3000 of them
f=$(find /tmp/ph -name "*hip_api_trace.csv" | head -1)
python - "$f" > $O/sync_calls.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print(rows[0].keys())
names = collections.Counter(r['Function'] for r in rows)
for k, v in names.most_common(40): print(v, k)
sync = [r for r in rows if any(s in r['Function'] for s in ('Synchronize', 'hipMemcpy', 'hipMalloc', 'hipFree', 'EventQuery', 'StreamQuery', 'hipStreamWaitEvent'))]
t0 = int(rows[0]['Start_Timestamp'])
c2 = collections.Counter(r['Function'] for r in sync)
print(c2)
for r in [r for r in sync if 'Synchronize' in r['Function'] or 'hipMemcpy' in r['Function'] or 'hipMalloc' in r['Function']][-120:]:
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e6:10.3f} ms  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:9.1f} us  {r['Function']}")
PY
head -60 $O/sync_calls.txt
