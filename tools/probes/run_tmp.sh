mkdir -p gpurun_out/r05h
run() { env "$@" python bench.py --no-cpu-baseline --steps 30 --warmup 8 2>gpurun_out/r05h/stderr.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$*', d['value'], d['ms_per_step'], d['step_ms']['median'], r['avg_launch_ms'], r['frac'], d['comm_exposed_ms'], d['memory']['device_allocs_in_timed_region'])"; grep -c "AccumulateGrad" gpurun_out/r05h/stderr.txt; }
for k in "A=1" "TGT_TRI_BWD2_DMA=0" "A=1" "TGT_TRI_BWD2_DMA=0"; do run $k; done | tee gpurun_out/r05h/ab.txt
