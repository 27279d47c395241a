# runtime environment knobs, same box, alternating
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06x; mkdir -p $O
cd $R
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_$tag.json
  python -c "
import json; d=json.loads(open('$O/bench_$tag.json').read()); s=d['step_ms']; print('$tag', d['value'], d['ms_per_step'], s['median'], 'host', s['host_enqueue_ms']['median'])"; }
for rep in 1 2; do
  run base_$rep X=1
  run devkernarg_$rep HIP_FORCE_DEV_KERNARG=1
  run hwq2_$rep GPU_MAX_HW_QUEUES=2
  run hwq8_$rep GPU_MAX_HW_QUEUES=8
done | tee $O/ab_env.txt
