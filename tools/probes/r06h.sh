set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06h; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests/test_hip_edge_gemm.py -m gpu -x -q > $O/pytest_edge.log 2>&1; tail -5 $O/pytest_edge.log
timeout 1200 python -m pytest tests/test_hip_model.py tests/test_hip_trainer.py -m gpu -x -q > $O/pytest_model.log 2>&1; tail -5 $O/pytest_model.log
ab() { env "$@" python bench.py --no-cpu-baseline --steps 30 --warmup 6 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$*', d['value'], d['ms_per_step'], 'median', d['step_ms']['median'], 'max', d['step_ms']['max'], 'dry', d['step_ms']['steps_stream_ran_dry'], 'host', d['step_ms']['host_enqueue_ms']['median'], 'bwd', r['avg_launch_ms'], r['frac'], 'loss', d['final_loss'])"; }
( ab TGT_EDGE_SKIP_DEAD=0; ab A=0; ab TGT_EDGE_SKIP_DEAD=0; ab A=0; ab TGT_EDGE_SKIP_DEAD=0; ab A=0 ) 2>&1 | grep -v "^+" | grep -v "^import\|^print" > $O/ab_skip.txt; cat $O/ab_skip.txt
