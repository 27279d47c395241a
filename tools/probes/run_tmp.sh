mkdir -p gpurun_out/r05k
python bench.py --no-cpu-baseline --nodes 48 --batch 128 --steps 20 --warmup 6 2>/dev/null | tail -1 > gpurun_out/r05k/bench_n48_b128.json
python -c "
import json; d=json.load(open('gpurun_out/r05k/bench_n48_b128.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], d['step_ms'], r['kernel'], r['avg_launch_ms'], r['frac'], r['timing']['alone'], {k:v['avg_launch_ms'] for k,v in r['other_kernels'].items()})"
