#!/bin/bash
# Where the matrix-core node attention kernels spend their time (micro-benchmark, B=256 N=32 bf16):
# TGT_NODE_ABLATE bits (1 no tile math, 2 no loads, 4 no stores) and one LDS / VALU counter pass.
#   tools/node_ablate.sh <outdir>
set -u
out=${1:?outdir}; mkdir -p "$out"; export TMPDIR=/tmp
root=$(cd "$(dirname "$0")/.." && pwd)
for ab in 0 1 2 4 3 5 6 7; do
  echo "ablate=$ab $(TGT_NODE_ABLATE=$ab python $root/tools/kernel_bench.py --only node)"
done | tee "$out/node_ablate.txt"
run() {  # name, counters...
    name=$1; shift
    rm -rf /tmp/pmc_$name
    ( cd /tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$name -o $name -- \
        python $root/tools/kernel_bench.py --only node --iters 3 ) > "$out/pmc_$name.log" 2>&1
    f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then grep -E "Counter_Name|tgt" "$f" > "$out/pmc_$name.csv"; fi
}
run lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
run valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU
run vmem SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM TCP_PENDING_STALL_CYCLES_sum
python - "$out" <<'PY'
import csv, sys, os, re, collections, json
out = sys.argv[1]; res = collections.defaultdict(dict)
for p in ('lds', 'valu', 'vmem'):
    f = os.path.join(out, f'pmc_{p}.csv')
    if not os.path.exists(f): continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        m = re.search(r'(node_att\w+?_kernel)', r['Kernel_Name'])
        if m: acc[(m.group(1), r['Counter_Name'])].append(float(r['Counter_Value']))
    for (k, c), v in acc.items(): res[k][c] = round(sum(v) / len(v), 1)
json.dump(res, open(os.path.join(out, 'node_pmc.json'), 'w'), indent=1); print(json.dumps(res, indent=1))
PY
