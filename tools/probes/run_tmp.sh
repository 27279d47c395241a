mkdir -p gpurun_out/r05j
timeout 1500 python -m pytest tests/test_hip_trainer.py tests/test_hip_model.py -x -q -m gpu 2>&1 | tail -3
run() { env "$@" python bench.py --no-cpu-baseline --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$*', d['value'], d['ms_per_step'], d['step_ms'], d['memory'])"; }
for k in "TGT_STREAM_KEEPALIVE=1" "TGT_STREAM_KEEPALIVE=0" "TGT_STREAM_KEEPALIVE=1" "TGT_STREAM_KEEPALIVE=0"; do run $k; done | tee gpurun_out/r05j/ab_keepalive.txt
