set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06b; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_hip_trainer.py tests/test_bench_rehearsal.py tests/test_hip_model.py -m gpu -x -q -k "graph or rehearsal or golden or two_ranks" -s > $O/pytest.log 2>&1; tail -15 $O/pytest.log; grep "gap agx2" $O/pytest.log
rm -rf /tmp/pt
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o bench -- python $R/bench.py --steps 3 --warmup 2 --settle-steps 6 --roofline-steps 0 --no-cpu-baseline ) > /tmp/pt.log 2>&1
f=$(find /tmp/pt -name "*kernel_trace.csv" | head -1)
python tools/timeline2.py $f bwd 14 > $O/timeline_bwd.txt 2>&1
python tools/timeline2.py $f bwd 30 > $O/timeline_bwd_b.txt 2>&1
python tools/timeline2.py $f fwd 14 > $O/timeline_fwd.txt 2>&1
tail -3 $O/timeline_bwd.txt $O/timeline_fwd.txt
# the last 1.2 steps of the trace for offline analysis (start / end / queue / short name)
python - "$f" > $O/trace_tail.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-2700:]
t0 = int(rows[0]['Start_Timestamp'])
print('start_us,dur_us,queue,name')
for r in rows:
    n = r['Kernel_Name'].split('(')[0][-60:].replace(',', ';')
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:.1f},{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.1f},{r['Queue_Id']},{n}")
PY
