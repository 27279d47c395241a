# Round 5, first GPU call: full GPU suite with the parity log, the multi-rank rehearsal, the default bench line, the node-gate A/B,
# and a two-stream kernel trace for the overlap question.   bash tools/probes/r06a.sh
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06a; mkdir -p $O
export TMPDIR=/tmp
cd $R
TGT_PARITY_LOG=$O/parity_errors.json timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
ab() { env "$@" python bench.py --no-cpu-baseline --steps 25 --warmup 6 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$*', d['value'], d['ms_per_step'], 'median', d['step_ms']['median'], 'max', d['step_ms']['max'], 'bwd in-region', r['avg_launch_ms'], r['frac'], 'alone', r['timing']['alone']['avg_launch_ms'], 'proj_fwd', r['other_kernels']['tgt_triplet_attention_proj_fwd']['avg_launch_ms'], 'stragglers', d['step_ms']['stragglers'], 'dry', d['step_ms']['steps_stream_ran_dry'], 'host', d['step_ms']['host_enqueue_ms'])"; }
( ab A=0; ab TGT_GATE_NODE_BWD=1; ab TGT_GATE_NODE_BWD=2; ab A=0; ab TGT_GATE_NODE_BWD=1; ab TGT_GATE_NODE_BWD=2; ab TGT_SIDE_PRIO=0 ) > $O/ab_gate.txt 2>&1; cat $O/ab_gate.txt
rm -rf /tmp/pt
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o bench -- python $R/bench.py --steps 4 --warmup 2 --settle-steps 6 --roofline-steps 0 --no-cpu-baseline ) > /tmp/pt.log 2>&1
f=$(find /tmp/pt -name "*kernel_trace.csv" | head -1)
( python tools/overlap_trace.py $f tri_att_bwd2 --last 72; python tools/overlap_trace.py $f tri_att_proj_fwd --last 72; python tools/overlap_trace.py $f node_att_mfma_bwd --last 72 ) > $O/overlap.txt 2>&1; cat $O/overlap.txt
