# Round-4 evidence set, one gpurun call:  bash tools/probes/prof_r05.sh <tag>
#   PMC passes (own runs), kernel micro-benchmarks, steady-state single-stream kernel table, rocprofv3 --stats of the
#   bench command as the driver runs it, and the default bench line (cpu_baseline included).
set -x
tag=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$tag; mkdir -p $O
export TMPDIR=/tmp
cd $R
tools/pmc_passes.sh $O "tricol node agg proj" > $O/pmc_passes.log 2>&1; tail -3 $O/pmc_passes.log
python tools/kernel_bench.py > $O/kernel_bench.json 2>/dev/null
python tools/edge_gemm_bench.py > $O/edge_gemm_bench.txt 2>&1
bash tools/probes/prof_steady.sh $tag
rm -rf /tmp/pstats
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pstats -o bench -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline ) > /tmp/pstats.log 2>&1
grep "^{" /tmp/pstats.log | tail -1 > $O/bench_under_rocprof.json
cp $(find /tmp/pstats -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
cd $R
# (bench.py quotes `traffic` / `mfma_util` from a sha-matched summary under profiles/: hand it the one this run has just measured)
cp $O/pmc_summary.json profiles/${tag}_pmc_summary.json 2>/dev/null || true
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_50steps.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --batch 128 --nodes 48 > $O/bench_n48_b128.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --ragged > $O/bench_ragged.json 2>> $O/bench.err
python tools/infer_bench.py > $O/infer_bench.jsonl 2>> $O/bench.err
bash tools/pmc_edge.sh $O > $O/pmc_edge_summary.txt 2>&1
tail -c 600 $O/bench.json
# BASELINE config 4 shape (N = 48, B = 128): counters of the 16-wide triplet kernels
PMC_KB_ARGS="--B 128 --N 48" tools/pmc_passes.sh $O/n48 "tricol" > $O/pmc_passes_n48.log 2>&1; tail -3 $O/pmc_passes_n48.log
python tools/kernel_bench.py --only "tricol" --B 128 --N 48 > $O/kernel_bench_n48.json 2>/dev/null
