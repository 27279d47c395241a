#!/bin/bash
# Wave-state counters of the tgt_edge_linear kernels at the BASELINE shapes (one rocprofv3 --pmc run with --kernel-trace only, over
# tools/edge_gemm_bench.py):   tools/pmc_edge_sq.sh <outdir>
# SQ_WAIT_ANY = wave parked (s_waitcnt / barrier), SQ_WAIT_INST_ANY = issue stall, SQ_ACTIVE_INST_ANY = issuing; the three add up to
# SQ_WAVE_CYCLES (MI355X_MICROARCH.md "rocprofv3 PMC slots").  Answers "what paces the row kernels" beyond the byte count.
set -u
out=${1:?outdir}; mkdir -p "$out"; export TMPDIR=/tmp
root=$(cd "$(dirname "$0")/.." && pwd)
rm -rf /tmp/pmcsq
( cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmcsq -o e -- python $root/tools/edge_gemm_bench.py ) > "$out/pmc_edge_sq.log" 2>&1
f=$(find /tmp/pmcsq -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && grep -E "Counter_Name|edge_|ln_bwd|Cijk" "$f" > "$out/pmc_edge_sq.csv"
python - "$out" <<'P'
import csv, sys, collections, re
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f'{out}/pmc_edge_sq.csv')):
    name = re.sub(r'\(.*', '', r['Kernel_Name'])[:58]
    acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
print(f'{"kernel":58s} {"n":>4s} {"wait%":>6s} {"stall%":>6s} {"issue%":>6s} {"VALU/wave":>9s} {"MFMA/wave":>9s}')
for name, d in sorted(acc.items()):
    m = {k: sum(v) / max(1, len(v)) for k, v in d.items()}
    wc = m.get('SQ_WAVE_CYCLES', 0) or 1
    print(f'{name:58s} {len(d.get("SQ_WAVE_CYCLES", [])):4d} {100 * m.get("SQ_WAIT_ANY", 0) / wc:6.1f} {100 * m.get("SQ_WAIT_INST_ANY", 0) / wc:6.1f} '
          f'{100 * m.get("SQ_ACTIVE_INST_ANY", 0) / wc:6.1f} {m.get("SQ_INSTS_VALU", 0):9.0f} {m.get("SQ_INSTS_MFMA", 0):9.0f}')
P
