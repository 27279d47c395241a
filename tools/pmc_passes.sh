#!/bin/bash
# Hardware-counter passes over the hand-written kernels at the BASELINE shape (B=256, N=32, bf16), each counter group in
# its OWN rocprofv3 run (--pmc with --kernel-trace only: MI355X_MICROARCH.md "rocprofv3 PMC slots"):
#   tools/pmc_passes.sh <outdir> [kernel_bench --only selection, default "tricol node"]     (PMC_KB_ARGS="--B 128 --N 48": another shape)
# pass mfma : SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE
# pass fetch: FETCH_SIZE          pass write: WRITE_SIZE
# then tools/pmc_summary.py <outdir> turns the three csv files into one json (per-kernel averages).
set -u
out=${1:?outdir}; sel=${2:-tricol node}
mkdir -p "$out"; export TMPDIR=/tmp
root=$(cd "$(dirname "$0")/.." && pwd)
run() {  # name, counters...
    name=$1; shift
    rm -rf /tmp/pmc_$name
    ( cd /tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$name -o $name -- \
        python $root/tools/kernel_bench.py --only "$sel" --iters 3 ${PMC_KB_ARGS:-} ) > "$out/pmc_$name.log" 2>&1
    f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then grep -E "Counter_Name|tgt" "$f" > "$out/pmc_$name.csv"; fi
}
run mfma SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
python $root/tools/pmc_summary.py "$out"
