# row chunks of the split-M weight gradients: cap 32 / 64 / 128 (default) / 256, re-swept in the GPU-bound regime (new shapes are tuned online in the warm-up)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06y2; mkdir -p $O
cd $R
for rep in 1 2; do for v in 128 64 32 256; do
  TGT_WGRAD_MAXP=$v timeout 900 python bench.py --no-cpu-baseline --steps 20 --warmup 8 2>/dev/null | tail -1 > $O/bench_maxp_$v.json
  python -c "
import json; d=json.loads(open('$O/bench_maxp_$v.json').read()); print('maxp=$v', d['value'], d['ms_per_step'], d['step_ms']['median'])"
done; done | tee $O/ab_wgrad_maxp.txt
