#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, each in its own rocprofv3 --pmc run with --kernel-trace only) of the tgt_edge_linear kernels
# at the BASELINE shapes, over tools/edge_gemm_bench.py:   tools/pmc_edge.sh <outdir>
# bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 as for the triplet kernels (profiles/README.md); per-kernel averages to stdout.
set -u
out=${1:?outdir}; mkdir -p "$out"; export TMPDIR=/tmp
root=$(cd "$(dirname "$0")/.." && pwd)
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmce_$c
    ( cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmce_$c -o e -- python $root/tools/edge_gemm_bench.py ) > "$out/pmc_edge_$c.log" 2>&1
    f=$(find /tmp/pmce_$c -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && grep -E "Counter_Name|edge_" "$f" > "$out/pmc_edge_$c.csv"
done
python - "$out" <<'P'
import csv, sys, collections, re
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    try:
        rows = list(csv.DictReader(open(f'{out}/pmc_edge_{c}.csv')))
    except OSError:
        continue
    for r in rows:
        name = re.sub(r'\(.*', '', r['Kernel_Name'])[:60]
        acc[name][c].append(float(r['Counter_Value']))
for name, d in sorted(acc.items()):
    f = sum(d['FETCH_SIZE']) / max(1, len(d['FETCH_SIZE']))
    w = sum(d['WRITE_SIZE']) / max(1, len(d['WRITE_SIZE']))
    print(f'{name:62s} launches {len(d["FETCH_SIZE"]):4d}  fetch {2 * f * 1024 / 1e6:8.1f} MB  write {w * 1024 / 1e6:8.1f} MB  total {(2 * f + w) * 1024 / 1e6:8.1f} MB')
P
