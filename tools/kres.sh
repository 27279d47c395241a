#!/bin/bash
# per-kernel register / spill summary of one HIP source:  tools/kres.sh tgt_amd/csrc/edge_gemm.hip [filter]
src=$1; filt=${2:-.}
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Rpass-analysis=kernel-resource-usage -c "$(realpath -e /root/repo/$src 2>/dev/null || echo $src)" -o /tmp/kres.o 2>&1 | \
  python3 -c "
import re,sys
cur=None
rows={}
for l in sys.stdin:
    if 'error' in l: print(l.rstrip())
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); rows[cur]={}
    for k in ('VGPRs','AGPRs','VGPRs Spill','SGPRs Spill','ScratchSize \[bytes/lane\]','LDS Size \[bytes/block\]','Occupancy \[waves/SIMD\]'):
        m=re.search(r'remark:\s+'+k+r': (\d+)',l)
        if m and cur: rows[cur][k[:6]]=int(m.group(1))
for k,v in rows.items():
    if re.search(r'$filt',k): print(k[:90], v)
"
