#!/bin/bash
# Wave-state and LDS counters of the node attention kernels (tools/kernel_bench.py --only node), two rocprofv3 --pmc runs with
# --kernel-trace only:   tools/pmc_node_sq.sh <outdir>        (PMC_KB_ARGS="--B 128 --N 48": config 4's shape)
# SQ_WAIT_ANY = wave parked (s_waitcnt / barrier), SQ_WAIT_INST_ANY = issue stall, SQ_ACTIVE_INST_ANY = issuing (add up to
# SQ_WAVE_CYCLES); SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = cycles the LDS spent on conflicts / was busy.
set -u
out=${1:?outdir}; mkdir -p "$out"; export TMPDIR=/tmp
root=$(cd "$(dirname "$0")/.." && pwd)
run() { name=$1; shift; rm -rf /tmp/pmcn_$name
  ( cd /tmp && timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmcn_$name -o n -- python $root/tools/kernel_bench.py --only node --iters 3 ${PMC_KB_ARGS:-} ) > "$out/pmc_node_$name.log" 2>&1
  f=$(find /tmp/pmcn_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && grep -E "Counter_Name|node_att" "$f" > "$out/pmc_node_$name.csv"; }
run sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES
python - "$out" <<'P'
import csv, sys, collections, re, os
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for part in ('sq', 'lds'):
    p = f'{out}/pmc_node_{part}.csv'
    if not os.path.exists(p):
        continue
    for r in csv.DictReader(open(p)):
        m = re.search(r'(node_att\w*?_kernel)', r['Kernel_Name'])
        acc[m.group(1) if m else r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
print(f'{"kernel":28s} {"wait%":>6s} {"stall%":>6s} {"issue%":>6s} {"VALU/wave":>9s} {"MFMA/wave":>9s} {"LDS busy cyc":>12s} {"bank conflict %":>15s} {"LDS insts":>10s}')
for name, d in sorted(acc.items()):
    m = {k: sum(v) / max(1, len(v)) for k, v in d.items()}
    wc = m.get('SQ_WAVE_CYCLES', 0) or 1
    la = m.get('SQ_LDS_IDX_ACTIVE', 0)
    print(f'{name:28s} {100 * m.get("SQ_WAIT_ANY", 0) / wc:6.1f} {100 * m.get("SQ_WAIT_INST_ANY", 0) / wc:6.1f} {100 * m.get("SQ_ACTIVE_INST_ANY", 0) / wc:6.1f} '
          f'{m.get("SQ_INSTS_VALU", 0):9.0f} {m.get("SQ_INSTS_MFMA", 0):9.0f} {la:12.0f} {100 * m.get("SQ_LDS_BANK_CONFLICT", 0) / la if la else 0:15.1f} {m.get("SQ_INSTS_LDS", 0):10.0f}')
P
