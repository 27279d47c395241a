#!/bin/bash
# Kernel durations of a kernel_bench selection from a rocprofv3 kernel trace (the micro-benchmark's own
# HIP-event numbers include Python launch overhead for kernels this short):
#   tools/kprof.sh <tag> <kernel_bench --only selection> [ENV=VAL ...]
set -u
tag=${1:?tag}; sel=${2:?selection}; shift 2
export TMPDIR=/tmp
root=$(cd "$(dirname "$0")/.." && pwd)
rm -rf /tmp/kprof_$tag
( cd /tmp && env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kprof_$tag -o $tag -- \
    python $root/tools/kernel_bench.py --only "$sel" --iters 10 ) > /tmp/kprof_$tag.log 2>&1
f=$(find /tmp/kprof_$tag -name "*kernel_stats.csv" | head -1)
python - "$f" "$tag" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if 'tgt' in n:
        m = re.search(r'tgt\d*(?:3nmf\d+)?(\w+?_kernel)', n)
        print(f"{sys.argv[2]:>12s} {(m.group(1) if m else n[:50]):40s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
PY
