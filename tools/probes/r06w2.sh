# what do the split-M weight gradients (library batched GEMMs + closing sums) cost the step?  (garbage gradients: timing only)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06w; mkdir -p $O
cd $R
for v in 0 1 0 1; do
  TGT_PROBE_SKIP_WGRAD=$v timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_skipw_$v.json
  python -c "
import json; d=json.loads(open('$O/bench_skipw_$v.json').read()); print('skip_wgrad=$v', d['value'], d['ms_per_step'], d['step_ms']['median'], 'loss', d['final_loss'])"
done | tee $O/ab_skip_wgrad.txt
