# as tools/ab_knobs.sh over 50 timed steps, with the allocator's activity inside the timed region
run() { env "$@" python bench.py --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['ms_per_step'], d['memory'])"; }
for k in "$@"; do run $k; done
