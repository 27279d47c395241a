#!/usr/bin/env python
"""A second build of libtgt_hip.so with ONE translation unit recompiled under extra flags (kernel A/Bs on one box):

    python tools/build_variant.py triplet_attention_bwd2.hip "-DTGT_BWD2_PF=2" tools/probes/lib_pf2.so
    TGT_HIP_LIB=$PWD/tools/probes/lib_pf2.so python tools/kernel_bench.py --only tricol

Every other object is taken from tgt_amd/build/ (run __graft_entry__.build() first).  *.so files are git-ignored and travel
with gpurun; variants are scratch, never the shipped library."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tgt_amd import _lib  # noqa: E402


def main():
    unit, flags, out = sys.argv[1], sys.argv[2].split(), sys.argv[3]
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    build = os.path.join(ROOT, 'tgt_amd', 'build')
    objs, mine = [], []
    for s in _lib.SOURCES:
        name, uflags, suffix = (s, [], '') if isinstance(s, str) else s
        o = os.path.join(build, name + suffix, 'unit.o')
        if name == unit:
            vo = os.path.join('/tmp', f'variant_{os.path.basename(out)}_{name}{suffix}.o')
            cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', *uflags, *flags, '-c',
                   os.path.join(ROOT, 'tgt_amd', 'csrc', name), '-o', vo]
            mine.append(subprocess.Popen(cmd))
            objs.append(vo)
        else:
            objs.append(o)
    assert mine, f'{unit} is not a unit of the library'
    for p in mine:
        assert p.wait() == 0
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs)
    print(out)


if __name__ == '__main__':
    main()
