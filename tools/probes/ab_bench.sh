# Same-box A/B of bench.py arms, alternating (boxes of the pool differ by +-2.5 %; one box repeats to +-0.1 % on the per-step median).
#   bash tools/probes/ab_bench.sh <tag> <rounds> "<arm>" ["<arm>" ...]
# An arm is a space-separated list of ENV=value assignments and / or bench.py flags, e.g.
#   bash tools/probes/ab_bench.sh r07a 2 "A=0" "TGT_EDGE_WGRAD=1" "--timing-probe skip_wgrad" "--nodes 48 --batch 128"
# Writes gpurun_out/<tag>/ab.txt (one line per run: arm, graphs/s, mean / median / max ms, roofline in region and alone, the
# node-attention launches, host enqueue time) and the last JSON line of every arm as gpurun_out/<tag>/<n>_<round>.json.
# This one script replaces the per-experiment r06a..r06y2.sh files of round 5 (git history: 3e21c4c).
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; rounds=$2; shift 2
O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
STEPS=${STEPS:-20}; WARMUP=${WARMUP:-6}
for ((i = 0; i < rounds; ++i)); do
  n=0
  for arm in "$@"; do
    envs=(); flags=()
    for w in $arm; do case $w in -*) flags+=($w);; *=*) if [ ${#flags[@]} -eq 0 ]; then envs+=($w); else flags+=($w); fi;; *) flags+=($w);; esac; done
    env "${envs[@]}" timeout 900 python bench.py --no-cpu-baseline --steps $STEPS --warmup $WARMUP "${flags[@]}" 2>/dev/null | tail -1 > $O/${n}_$i.json
    python - "$arm" $O/${n}_$i.json <<'PY'
import json, sys
arm, f = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open(f).read())
except Exception as e:
    print(arm, 'FAILED', e); raise SystemExit
r = d.get('roofline') or {}
ok = r.get('other_kernels', {})
val = d['value'] if d.get('value') is not None else f"(invalid: {d.get('graphs_per_s_with_work_missing')})"
print(arm, '|', val, 'graphs/s', d['ms_per_step'], 'ms; median', d['step_ms']['median'], 'max', d['step_ms']['max'],
      '| roofline', r.get('avg_launch_ms'), r.get('frac'), 'alone', (r.get('timing') or {}).get('alone', {}).get('frac'),
      '| node fwd/bwd', ok.get('tgt_node_attention_fwd', {}).get('avg_launch_ms'), ok.get('tgt_node_attention_bwd', {}).get('avg_launch_ms'),
      '| proj_fwd', ok.get('tgt_triplet_attention_proj_fwd', {}).get('avg_launch_ms'),
      '| host', d['step_ms']['host_enqueue_ms']['median'], 'dry', d['step_ms']['steps_stream_ran_dry'], '| knobs', d.get('knobs_not_default'))
PY
    n=$((n + 1))
  done
done | tee $O/ab.txt
