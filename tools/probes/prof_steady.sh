# steady-state per-kernel table of the training step (single stream, so that kernel durations are clean), r03 form:
#   bash tools/probes/prof_steady.sh <tag>   ->  gpurun_out/<tag>_bench_kernel_stats_steady.csv, <tag>_trace_layer.txt
set -x
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/ps
TGT_NODE_STREAM=0 TGT_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ps -o bench -- python $R/bench.py --steps 8 --warmup 6 --no-cpu-baseline > /tmp/ps.log 2>&1
grep "^{" /tmp/ps.log | tail -1 > $R/gpurun_out/${tag}_bench_under_rocprof_single_stream.json
f=$(find /tmp/ps -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_summary.py $f --steps 5 > $R/gpurun_out/${tag}_bench_kernel_stats_steady.csv
python $R/tools/trace_layer.py $f > $R/gpurun_out/${tag}_trace_layer.txt 2>&1 || true
