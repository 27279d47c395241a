# same-box A/B of environment knobs inside the training step (boxes of the pool differ by +-2.5 %, one box repeats to +-0.1 % on the
# per-step MEDIAN; the mean moves with allocator / clock transients):
#   bash tools/ab_knobs.sh "A=0" "TGT_EPI_LN_BWD=0" "TGT_FFN_GELU_BWD_EPI=1" "A=0"      -> knob, graphs/s, mean ms, median ms
run() { env "$@" python bench.py --no-cpu-baseline --steps 25 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['ms_per_step'], d['step_ms']['median'])"; }
for k in "$@"; do run $k; done
