#!/bin/bash
# Wave-state and LDS counters of the fused data + weight gradient row kernel (csrc/edge_wgrad.hip) next to the plain data-gradient launch
# and the library's batched weight-gradient GEMM it replaces, over tools/edge_wgrad_bench.py (BASELINE shape):   tools/pmc_edge_wgrad.sh <outdir>
# Two rocprofv3 --pmc runs with --kernel-trace only (MI355X_MICROARCH.md "rocprofv3 PMC slots"): wave states; LDS activity.
set -u
out=${1:?outdir}; mkdir -p "$out"; export TMPDIR=/tmp
root=$(cd "$(dirname "$0")/.." && pwd)
run() { name=$1; shift; rm -rf /tmp/pmcw_$name
  ( cd /tmp && timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmcw_$name -o e -- python $root/tools/edge_wgrad_bench.py ) > "$out/pmc_wgrad_$name.log" 2>&1
  f=$(find /tmp/pmcw_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && grep -E "Counter_Name|edge_rows|Cijk|sum_planes" "$f" > "$out/pmc_wgrad_$name.csv"; }
run sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE
run lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES
python - "$out" <<'P'
import csv, sys, collections, re
out = sys.argv[1]
for tag, cols in (('sq', None), ('lds', None)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    try:
        rows = list(csv.DictReader(open(f'{out}/pmc_wgrad_{tag}.csv')))
    except FileNotFoundError:
        continue
    for r in rows:
        name = re.sub(r'\(.*', '', r['Kernel_Name'])[:64]
        acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
    if tag == 'sq':
        print(f'{"kernel":64s} {"n":>4s} {"wait%":>6s} {"stall%":>6s} {"issue%":>6s} {"VALU":>10s} {"MFMA":>9s} {"wave-cycles":>12s}')
        for name, d in sorted(acc.items()):
            m = {k: sum(v) / max(1, len(v)) for k, v in d.items()}
            wc = m.get('SQ_WAVE_CYCLES', 0) or 1
            print(f'{name:64s} {len(d.get("SQ_WAVE_CYCLES", [])):4d} {100 * m.get("SQ_WAIT_ANY", 0) / wc:6.1f} {100 * m.get("SQ_WAIT_INST_ANY", 0) / wc:6.1f} '
                  f'{100 * m.get("SQ_ACTIVE_INST_ANY", 0) / wc:6.1f} {m.get("SQ_INSTS_VALU", 0):10.0f} {m.get("SQ_INSTS_MFMA", 0):9.0f} {wc:12.0f}')
    else:
        print(f'{"kernel":64s} {"n":>4s} {"LDS insts":>10s} {"bank conflict cyc":>18s} {"LDS active cyc":>15s} {"conflict %":>10s} {"LDS issue stall % of wave cycles":>33s} {"MFMA busy cyc":>14s}')
        for name, d in sorted(acc.items()):
            m = {k: sum(v) / max(1, len(v)) for k, v in d.items()}
            wc = m.get('SQ_WAVE_CYCLES', 0) or 1
            act = m.get('SQ_LDS_IDX_ACTIVE', 0) or 1
            print(f'{name:64s} {len(d.get("SQ_WAVE_CYCLES", [])):4d} {m.get("SQ_INSTS_LDS", 0):10.0f} {m.get("SQ_LDS_BANK_CONFLICT", 0):18.0f} {act:15.0f} '
                  f'{100 * m.get("SQ_LDS_BANK_CONFLICT", 0) / act:10.1f} {100 * m.get("SQ_WAIT_INST_LDS", 0) / wc:33.1f} {m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0):14.0f}')
P
