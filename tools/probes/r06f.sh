set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06f; mkdir -p $O
export TMPDIR=/tmp
cd $R
ab() { env "$@" python bench.py --no-cpu-baseline --steps 25 --warmup 6 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$*', d['value'], d['ms_per_step'], 'median', d['step_ms']['median'], 'max', d['step_ms']['max'], 'dry', d['step_ms']['steps_stream_ran_dry'], 'lead', d['step_ms']['host_lead_steps']['per_step'], 'host', d['step_ms']['host_enqueue_ms'], 'bwd', r['avg_launch_ms'], r['frac'], r['launches'], r['launches_timed'], 'share', r['share_of_step'])"; }
( ab A=0; ab TGT_BENCH_PROFILE_ALL=1; ab A=0; ab TGT_NODE_STREAM=0 ) 2>&1 | grep -v "^+" | grep -v "^import\|^print" > $O/ab.txt; cat $O/ab.txt
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 2500 $O/bench.json
