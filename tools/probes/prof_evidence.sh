# A round's evidence set, one gpurun call:  bash tools/probes/prof_evidence.sh <tag> [pmc]
#   the bench lines (default, the driver's command, 50 steps, ragged, config 4, the two-rank rehearsal), kernel micro-benchmarks, the
#   steady-state single-stream kernel table + one layer's launch sequence, rocprofv3 --stats of the command as the driver runs it.
#   `pmc`: re-run the hardware-counter passes first (both shapes + the edge row kernels) -- needed whenever tgt_amd/csrc changed:
#   bench.py quotes `traffic` / `mfma_util` only from a summary whose `_kernel_src_sha` equals the tree's.
set -x
tag=${1:-r07z}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$tag; mkdir -p $O
export TMPDIR=/tmp
cd $R
if [ "$2" = "pmc" ]; then
  tools/pmc_passes.sh $O "tricol node agg proj" > $O/pmc_passes.log 2>&1; tail -3 $O/pmc_passes.log
  PMC_KB_ARGS="--B 128 --N 48" tools/pmc_passes.sh $O/n48 "tricol node" > $O/pmc_passes_n48.log 2>&1; tail -3 $O/pmc_passes_n48.log
  bash tools/pmc_edge.sh $O > $O/pmc_edge_summary.txt 2>&1
  cp $O/pmc_summary.json profiles/${tag}_pmc_summary.json 2>/dev/null || true
  cp $O/n48/pmc_summary.json profiles/${tag}_n48_pmc_summary.json 2>/dev/null || true
fi
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_cmd.json 2>> $O/bench.err
python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_50steps.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --ragged > $O/bench_ragged.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --batch 128 --nodes 48 > $O/bench_n48_b128.json 2>> $O/bench.err
python bench.py --gpus 2 --share-device --batch 64 --steps 5 --warmup 2 --settle-steps 6 --no-cpu-baseline > $O/bench_rehearsal_2ranks_one_gpu.json 2>> $O/bench.err
python tools/kernel_bench.py > $O/kernel_bench.json 2>/dev/null
python tools/edge_gemm_bench.py > $O/edge_gemm_bench.txt 2>&1
python tools/infer_bench.py > $O/infer_bench.jsonl 2>> $O/bench.err
# steady-state single-stream kernel table + one layer's sequence
rm -rf /tmp/ps
( cd /tmp && TGT_NODE_STREAM=0 TGT_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ps -o bench -- python $R/bench.py --steps 8 --warmup 6 --no-cpu-baseline ) > /tmp/ps.log 2>&1
grep "^{" /tmp/ps.log | tail -1 > $O/bench_under_rocprof_single_stream.json
f=$(find /tmp/ps -name "*kernel_trace.csv" | head -1)
python tools/trace_summary.py $f --steps 5 > $O/bench_kernel_stats_steady.csv
python tools/trace_layer.py $f > $O/trace_layer.txt 2>&1 || true
python tools/trace_edges.py $f > $O/trace_edges.txt 2>&1 || true
# the command as the driver runs it, under rocprofv3 --stats (two streams)
rm -rf /tmp/pstats
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pstats -o bench -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline ) > /tmp/pstats.log 2>&1
grep "^{" /tmp/pstats.log | tail -1 > $O/bench_under_rocprof.json
cp $(find /tmp/pstats -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
k=$(find /tmp/pstats -name "*kernel_trace.csv" | head -1)
python tools/timeline2.py $k bwd 14 > $O/timeline_bwd_two_queues.txt 2>&1
python tools/timeline2.py $k fwd 14 > $O/timeline_fwd_two_queues.txt 2>&1
tail -c 900 $O/bench.json
