# node attention for N 33..64 / whole-line forward: parity, then A/B on config 4's shape and on the headline
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06p; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "node_attention" 2>&1 | tail -15 > $O/pytest_node.txt; cat $O/pytest_node.txt
TGT_NODE_KB=2 TGT_NODE_MFMA16=2 timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "node_attention" 2>&1 | tail -15 > $O/pytest_node_mode2.txt; cat $O/pytest_node_mode2.txt
timeout 900 python -m pytest tests/test_hip_model.py -m gpu -x -q -k "n48" 2>&1 | tail -8 > $O/pytest_n48.txt; cat $O/pytest_n48.txt
pr() { python -c "
import json,sys; d=json.loads(open('$1').read()); r=d['roofline']['other_kernels']; print('$2', d['value'], d['ms_per_step'], r.get('tgt_node_attention_fwd'), r.get('tgt_node_attention_bwd'))"; }
for v in 0 1 0 1; do
  TGT_NODE_KB=$v timeout 600 python bench.py --no-cpu-baseline --nodes 48 --batch 128 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_n48_kb_$v.json
  pr $O/bench_n48_kb_$v.json "n48 kb=$v"
done
for v in 1 2 1 2 1 2; do
  TGT_NODE_KB=$v timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_kb_$v.json
  pr $O/bench_kb_$v.json "headline kb=$v"
done
