set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06j; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_full.log 2>&1; tail -4 $O/pytest_full.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py --steps 300 --warmup 10 --no-cpu-baseline > $O/bench_300steps.json 2> $O/bench.err; python -c "
import json; d=json.loads(open('$O/bench_300steps.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['step_ms']['min'], d['step_ms']['median'], d['step_ms']['max'], d['step_ms']['stragglers'], d['step_ms']['host_lead_steps']['per_step'][:64], d['memory'], d['final_loss'])"
