"""Lint gfx950 assembly for (1) vector registers (AGPRs and VGPRs) that are read but never written and (2) wide vector
stores whose data registers are overwritten before the store has read them (lint_store_hazard).

Why: hipcc (ROCm 7.2) miscompiled one triplet-attention backward instantiation -- a 128-bit
loop-invariant MFMA operand was spilled as 3 dwords to scratch + 1 dword parked in an AGPR
("Reload Reuse"), and the reload restored only the 3 scratch dwords: the MFMA then read an AGPR
no instruction of the kernel ever writes, i.e. whatever the previous wave left there (DESIGN.md section 4 / profiles/HISTORY_rounds_1-4.md
section 4.1, "bf16 two-tile dropout backward").  The same static def/use check would have caught it
at build time, so `__graft_entry__.build()` can run it (TGT_ISA_LINT=1) and
tests/test_isa_lint.py keeps it on the affected translation unit.

An AGPR has no hardware-initialised value, so "read by some instruction, written by none" is always a
bug.  VGPRs are checked the same way, except v0..v2 (work-item ids arrive there) and the half of a
64-bit operand a packed-fp32 instruction does not select (op_sel / op_sel_hi broadcast idiom).

usage: python tools/isa_defuse_lint.py file.s [...]      (exit 1 when a kernel fails)
       python tools/isa_defuse_lint.py --build            (emit asm for every TU of the library first)
"""
import os
import re
import subprocess
import sys

_REG = re.compile(r'^([av])(?:(\d+)|\[(\d+):(\d+)\])$')
_LABEL = re.compile(r'^([A-Za-z_.$][\w.$]*):$')
_SEL = re.compile(r'\b(op_sel|op_sel_hi):\[([01,]+)\]')


def _regs(tok):
    """'a[4:7]' -> ('a', [4,5,6,7]);  's3' -> (None, [])."""
    m = _REG.match(tok)
    if not m:
        return None, []
    if m.group(2) is not None:
        return m.group(1), [int(m.group(2))]
    return m.group(1), list(range(int(m.group(3)), int(m.group(4)) + 1))


def lint_text(text):
    """-> {kernel: ['a183', ...]} for vector registers read but never written inside that kernel."""
    bad = {}
    kernel, reads, writes = None, set(), set()

    def close():
        undefined = {x for x in reads - writes if not (x[0] == 'v' and x[1] <= 2)}
        if kernel is not None and undefined:
            bad[kernel] = [f'{f}{n}' for f, n in sorted(undefined)]

    for line in text.splitlines():
        s = line.split(';')[0].strip()
        if not s:
            continue
        m = _LABEL.match(s)
        if m:
            name = m.group(1)
            if name.startswith('.Lfunc_end'):
                close()
                kernel = None
            elif not name.startswith(('.', 'BB')):        # a function entry (basic blocks are .LBBn_m)
                close()
                kernel, reads, writes = name, set(), set()
            continue
        if s.startswith('.') or kernel is None:
            continue
        parts = s.split(None, 1)
        if len(parts) < 2:
            continue
        op = parts[0]
        ops = [t.strip().split()[0] for t in parts[1].split(',') if t.strip()]
        stores = 'store' in op or op.startswith('ds_write') or (op.startswith('global_atomic') and 'ret' not in op)
        # packed fp32: the low result reads half op_sel[k] of source k, the high result half op_sel_hi[k]
        sel = None
        if op.startswith('v_pk_') and op.endswith(('_f32', '_b32')):
            mods = {k: [int(x) for x in v.split(',')] for k, v in _SEL.findall(parts[1])}
            sel = (mods.get('op_sel', [0, 0, 0]), mods.get('op_sel_hi', [1, 1, 1]))
        for i, t in enumerate(ops):
            file, regs = _regs(t)
            if file is None:
                continue
            if i == 0 and not stores:
                writes.update((file, r) for r in regs)
                continue
            if sel is not None and len(regs) == 2 and 1 <= i <= 3:
                k = i - 1
                lo = sel[0][k] if k < len(sel[0]) else 0
                hi = sel[1][k] if k < len(sel[1]) else 1
                regs = sorted({regs[lo], regs[hi]})
            reads.update((file, r) for r in regs)
    close()
    return bad


_WIDE_STORE = re.compile(r'^(buffer|global|flat|scratch)_store_dwordx[34]$')


def lint_store_hazard(text, states=1):
    """-> {kernel: ['v114 <- v_cvt_f32_f16_e32 @line N', ...]}: a 96/128-bit vector store whose DATA registers a VALU
    instruction overwrites within `states` wait states of the store.

    Why: the store reads its data registers after it has issued.  hipcc pads that hazard only for stores without a scalar
    offset; on MI355X a `buffer_store_dwordx4 v[114:117], v72, s[12:15], s28 offen` directly followed by
    `v_cvt_f32_f16 v114, ...` stored the conversion result in place of the first dword (the fp16 column-sum variant of
    csrc/triplet_attention16.hip: DESIGN.md section 4.1b of profiles/HISTORY_rounds_1-4.md).  `s_nop N` between the two counts as N + 1 states.
    states = 1: the distance at which corruption was observed; the same overwrite one instruction later (129 places in
    this library, all in kernels that pass their parity tests) has never shown it."""
    bad = {}
    kernel, pending = None, []          # pending: [(remaining states, frozenset of data regs, store text)]
    for ln, line in enumerate(text.splitlines(), 1):
        s = line.split(';')[0].strip()
        if not s:
            continue
        m = _LABEL.match(s)
        if m:
            name = m.group(1)
            if name.startswith('.Lfunc_end'):
                kernel, pending = None, []
            elif not name.startswith(('.', 'BB')):
                kernel, pending = name, []
            continue
        if s.startswith('.') or kernel is None:
            continue
        parts = s.split(None, 1)
        op = parts[0]
        ops = [t.strip().split()[0] for t in parts[1].split(',') if t.strip()] if len(parts) > 1 else []
        cost = 1
        if op == 's_nop' and ops:
            try:
                cost = int(ops[0], 0) + 1
            except ValueError:
                cost = 1
        elif op.startswith('v_') and ops and pending:
            file, regs = _regs(ops[0])
            if file == 'v' and not op.startswith(('v_cmp', 'v_readlane', 'v_readfirstlane')):
                for left, data, what in pending:
                    hit = sorted(set(regs) & data)
                    if hit:
                        bad.setdefault(kernel, []).append(f'v{hit[0]} <- {op} @line {ln} ({what})')
        pending = [(left - cost, data, what) for left, data, what in pending if left - cost > 0]
        if _WIDE_STORE.match(op) and ops:
            data_tok = ops[0] if op.startswith('buffer') else (ops[1] if len(ops) > 1 else '')
            file, regs = _regs(data_tok)
            if file == 'v':
                pending.append((states, frozenset(regs), s[:60]))
    return bad


def emit_asm(out_dir):
    """Device-only asm of every translation unit of libtgt_hip.so -> [paths]."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tgt_amd import _lib
    os.makedirs(out_dir, exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    procs, outs = [], []
    for u in _lib.SOURCES:
        src, flags, suffix = (u, [], '') if isinstance(u, str) else u
        out = os.path.join(out_dir, src + suffix + '.s')
        outs.append(out)
        cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', *flags, '--cuda-device-only', '-S',
               os.path.join(_lib.CSRC, src), '-o', out]
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        log, _ = p.communicate()
        if p.returncode:
            raise RuntimeError('hipcc -S failed: ' + ' '.join(cmd) + '\n' + log.decode(errors='replace'))
    return outs


def lint_all(text):
    """both checks: {kernel: [findings]}"""
    bad = {k: list(v) for k, v in lint_text(text).items()}
    for k, v in lint_store_hazard(text).items():
        bad.setdefault(k, []).extend(v)
    return bad


def lint_files(paths):
    bad = {}
    for p in paths:
        with open(p) as f:
            for k, regs in lint_all(f.read()).items():
                bad[f'{os.path.basename(p)}:{k}'] = regs
    return bad


def main(argv):
    paths = [a for a in argv if not a.startswith('--')]
    if '--build' in argv:
        paths += emit_asm(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tgt_amd', 'build', 'isa'))
    bad = lint_files(paths)
    for k, regs in bad.items():
        print(f'ISA LINT  {k}: {regs}')
    print(f'{len(paths)} files, {len(bad)} kernels with findings (undefined vector-register reads / store-data hazards)')
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))
