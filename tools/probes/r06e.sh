set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06e; mkdir -p $O
export TMPDIR=/tmp
cd $R
ab() { env "$@" python bench.py --no-cpu-baseline --steps 25 --warmup 6 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$*', d['value'], d['ms_per_step'], 'median', d['step_ms']['median'], 'max', d['step_ms']['max'], 'dry', d['step_ms']['steps_stream_ran_dry'], 'lead', d['step_ms']['host_lead_steps']['median'], 'host', d['step_ms']['host_enqueue_ms'], 'bwd', r['avg_launch_ms'], r['frac'])"; }
( ab TGT_BENCH_PROFILE_ALL=1; ab A=0; ab TGT_BENCH_PROFILE_ALL=1; ab A=0; ab TGT_BENCH_PROFILE_ALL=1 TGT_EMBED_GEMM=0; ab A=0 ) 2>&1 | grep -v "^+" > $O/ab_host.txt; cat $O/ab_host.txt
python bench.py --no-cpu-baseline --nodes 48 --batch 128 > $O/bench_n48.json 2>$O/err.txt; python -c "
import json; d=json.loads(open('$O/bench_n48.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['traffic'], d['roofline']['offline_pmc'], d['roofline']['mfma_util'])"
timeout 600 python -m pytest tests/test_hip_model.py -m gpu -x -q -k "tiny or task or golden" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
