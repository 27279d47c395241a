set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06g; mkdir -p $O
export TMPDIR=/tmp
cd $R
ab() { env "$@" python bench.py --no-cpu-baseline --steps 30 --warmup 6 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$*', d['value'], d['ms_per_step'], 'median', d['step_ms']['median'], 'max', d['step_ms']['max'], 'dry', d['step_ms']['steps_stream_ran_dry'], 'host', d['step_ms']['host_enqueue_ms']['median'], 'bwd', r['avg_launch_ms'], r['frac'])"; }
( ab A=0; ab TGT_DEFER_SUMS=1; ab TGT_WGRAD_STREAM=1; ab A=0; ab TGT_TRI_SKIP=2; ab TGT_SIDE_PRIO=0; ab TGT_GATE_NODE_BWD=1; ab A=0; ab TGT_STREAM_KEEPALIVE=1; ab TGT_DEFER_SUMS=1 TGT_DEFER_MAX=8 ) 2>&1 | grep -v "^+" | grep -v "^import\|^print" > $O/ab_knobs.txt; cat $O/ab_knobs.txt
python tools/graph_step_bench.py --batch 256 --nodes 32 --steps 30 > $O/graph_step_b256.jsonl 2>$O/err.txt; cat $O/graph_step_b256.jsonl
