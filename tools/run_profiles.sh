set -u
export TMPDIR=/tmp
O=${1:-gpurun_out/r02g}; mkdir -p $O
python -m pytest tests/test_torch_ops.py -q -x 2>&1 | tail -3
tools/pmc_passes.sh $O "tricol node agg" > $O/pmc_passes.log 2>&1; tail -3 $O/pmc_passes.log
python tools/kernel_bench.py > $O/kernel_bench.json 2>/dev/null; cat $O/kernel_bench.json
rm -rf /tmp/prof_run
( cd /tmp && TGT_NODE_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_run -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 4 --no-cpu-baseline ) > $O/prof_bench.log 2>&1
tail -1 $O/prof_bench.log | cut -c1-300
f=$(find /tmp/prof_run -name "*kernel_trace.csv" | head -1)
python tools/trace_summary.py $f --steps 5 --top 25 > $O/bench_kernel_stats_steady.csv 2> $O/bench_kernel_top.txt
head -30 $O/bench_kernel_top.txt | cut -c1-160
