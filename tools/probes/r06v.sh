# what do the closing sums (tgt_sum_planes: ~364 launches per step) cost the step?  skip them (garbage gradients: timing only)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06v; mkdir -p $O
cd $R
for v in 0 1 0 1; do
  TGT_PROBE_SKIP_SUMS=$v timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_skip_$v.json
  python -c "
import json; d=json.loads(open('$O/bench_skip_$v.json').read()); print('skip_sums=$v', d['value'], d['ms_per_step'], d['step_ms']['median'])"
done | tee $O/ab_skip_sums.txt
