# config 4 (N = 48, B = 128): steady-state kernel table on one stream
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06u; mkdir -p $O
export TMPDIR=/tmp
cd $R
rm -rf /tmp/ps
( cd /tmp && TGT_NODE_STREAM=0 TGT_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ps -o bench -- python $R/bench.py --steps 8 --warmup 6 --no-cpu-baseline --nodes 48 --batch 128 ) > /tmp/ps.log 2>&1
grep "^{" /tmp/ps.log | tail -1 > $O/bench_n48_under_rocprof_single_stream.json
f=$(find /tmp/ps -name "*kernel_trace.csv" | head -1)
python tools/trace_summary.py $f --steps 5 > $O/bench_n48_kernel_stats_steady.csv
python tools/trace_layer.py $f > $O/trace_layer_n48.txt 2>&1 || true
head -30 $O/bench_n48_kernel_stats_steady.csv | cut -c1-200
