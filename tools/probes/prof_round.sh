# Round profile: rocprofv3 kernel stats of the bench command (default = node FFN on a second stream, and
# single-stream for clean per-kernel durations), PMC traffic of the training-path triplet kernels.
set -x
cd /tmp && export TMPDIR=/tmp
R=/root/repo
mkdir -p $R/gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o bench -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /tmp/b1.log 2>&1
tail -1 /tmp/b1.log > $R/gpurun_out/prof/bench_under_rocprof.json
cp /tmp/p1/bench_kernel_stats.csv $R/gpurun_out/prof/bench_kernel_stats.csv
TGT_NODE_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1s -o bench -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /tmp/b1s.log 2>&1
tail -1 /tmp/b1s.log > $R/gpurun_out/prof/bench_under_rocprof_single_stream.json
cp /tmp/p1s/bench_kernel_stats.csv $R/gpurun_out/prof/bench_kernel_stats_single_stream.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p2 -o f -- python $R/tools/kernel_bench.py --only tricol --iters 3 > /tmp/b2.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p3 -o w -- python $R/tools/kernel_bench.py --only tricol --iters 3 > /tmp/b3.log 2>&1
tail -1 /tmp/b2.log
head -1 /tmp/p2/f_counter_collection.csv > $R/gpurun_out/prof/pmc_tri_att.csv
grep -h "tri_att" /tmp/p2/f_counter_collection.csv | tail -6 >> $R/gpurun_out/prof/pmc_tri_att.csv
grep -h "tri_att" /tmp/p3/w_counter_collection.csv | tail -6 >> $R/gpurun_out/prof/pmc_tri_att.csv
python $R/bench.py > $R/gpurun_out/prof/bench_line.json 2>/tmp/b4.log
TGT_NODE_STREAM=0 python $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof/bench_line_single_stream.json 2>/tmp/b5.log
tail -c 300 $R/gpurun_out/prof/bench_line.json
