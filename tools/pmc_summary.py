#!/usr/bin/env python
"""Per-kernel averages of the counter passes of tools/pmc_passes.sh -> <outdir>/pmc_summary.json.

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE reports half of a wide coalesced
streaming read (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is taken as is (it equals the output size exactly on the
triplet forward, profiles/README.md).  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4
SIMDs): the gfx94x derived-counter formula (ROCm 7.2 ships none for gfx950) with GRBM_GUI_ACTIVE divided by the 8 XCDs it
is summed over here (4.33 M "cycles" for a 248 us kernel = 8 x 2.18 GHz); and, independent of any busy counter,
MFMA FLOP rate = SQ_INSTS_VALU_MFMA_MOPS_BF16 * 512 FLOP / kernel time against the 2.5 PFLOP/s dense bf16 peak."""
import csv
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    """the kernel's own identifier (`tri_att_bwd2_kernel`) from a mangled (`_ZN3tgt4bwd219tri_att_bwd2_kernelI...`) or a demangled
    (`void tgt::bwd2::tri_att_bwd2_kernel<...>`) trace name"""
    m = re.search(r'([A-Za-z_][A-Za-z0-9_]*_kernel)', name)
    if not m:
        return name[:60]
    s = m.group(1)
    # Itanium mangling: <length><identifier>; take the identifier whose length prefix fits
    for i in range(len(s) - 1, 0, -1):
        if s[i - 1].isdigit() and not s[i].isdigit():
            j = i - 1
            while j > 0 and s[j - 1].isdigit():
                j -= 1
            for k in range(j, i):
                if int(s[k:i]) == len(s) - i:
                    return s[i:]
    return s


def kernel_source_sha(root=None):
    """sha256 (16 hex) over every kernel source of tgt_amd/csrc: what a counter summary was measured ON (bench.py only quotes a
    summary whose hash equals the tree's)"""
    import hashlib
    root = root or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, 'tgt_amd', 'csrc')
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.hpp', '.cpp')):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()[:16]


def main():
    out = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for p in ('mfma', 'fetch', 'write'):
        f = os.path.join(out, f'pmc_{p}.csv')
        if not os.path.exists(f):
            continue
        seen = set()
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
            if p == 'mfma' and r['Dispatch_Id'] not in seen:
                seen.add(r['Dispatch_Id'])
                dur[k].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    res = {}
    for k, c in acc.items():
        a = {n: sum(v) / len(v) for n, v in c.items()}
        d = dict(launches=max(len(v) for v in c.values()), counters={n: round(v, 1) for n, v in a.items()})
        if dur[k]:
            d['avg_ns_under_pmc'] = round(sum(dur[k]) / len(dur[k]))
        if 'FETCH_SIZE' in a and 'WRITE_SIZE' in a:
            d['hbm_bytes_per_launch'] = int((2 * a['FETCH_SIZE'] + a['WRITE_SIZE']) * 1024)
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in a and a.get('GRBM_GUI_ACTIVE'):
            d['mfma_util'] = round(a['SQ_VALU_MFMA_BUSY_CYCLES'] / (a['GRBM_GUI_ACTIVE'] / 8 * 256 * 4), 4)
        if 'SQ_INSTS_VALU_MFMA_MOPS_BF16' in a and dur[k]:
            flops = a['SQ_INSTS_VALU_MFMA_MOPS_BF16'] * 512
            d['mfma_flop_per_launch'] = flops
            d['mfma_tflops_under_pmc'] = round(flops / (sum(dur[k]) / len(dur[k])) / 1e3, 1)
            d['mfma_frac_of_2.5PF'] = round(flops / (sum(dur[k]) / len(dur[k])) / 1e3 / 2500, 4)
        if 'SQ_INSTS_VALU' in a and 'SQ_INSTS_MFMA' in a and a['SQ_INSTS_MFMA']:
            d['valu_per_mfma'] = round(a['SQ_INSTS_VALU'] / a['SQ_INSTS_MFMA'], 1)
        res[k] = d
    res['_kernel_src_sha'] = kernel_source_sha()
    json.dump(res, open(os.path.join(out, 'pmc_summary.json'), 'w'), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
