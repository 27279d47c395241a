# what does the projection's standalone LayerNorm backward (ln_bwd_kernel<32,1,true>, 24 launches of ~0.1 ms) cost the step?
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06w; mkdir -p $O
cd $R
for v in 0 1 0 1; do
  TGT_PROBE_SKIP_PROJ_LN=$v timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_skip_$v.json
  python -c "
import json; d=json.loads(open('$O/bench_skip_$v.json').read()); print('skip_proj_ln=$v', d['value'], d['ms_per_step'], d['step_ms']['median'])"
done | tee $O/ab_skip_proj_ln.txt
