mkdir -p gpurun_out/r05o
run() { TGT_HIP_LIB=${1:+$PWD/$1} python bench.py --no-cpu-baseline --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('lib=$1', d['value'], d['step_ms']['median'], r['avg_launch_ms'], {k:v['avg_launch_ms'] for k,v in r['other_kernels'].items()})"; }
for rep in 1 2; do for lib in "" tools/probes/lib_er1.so tools/probes/lib_er2.so tools/probes/lib_b2p.so; do run $lib; done; done | tee gpurun_out/r05o/ab_prio.txt
