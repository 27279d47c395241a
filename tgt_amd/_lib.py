"""ctypes binding of libtgt_hip.so (C ABI: include/tgt_hip.h).

There is NO fallback: if the shared library is missing or a call fails, a
RuntimeError is raised.  The library is built in-tree by `build_library()`
(`__graft_entry__.build()` calls it) with hipcc for gfx950.
"""
import ctypes as C
import hashlib
import importlib.util
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libtgt_hip.so')
# same-box A/B of KERNEL changes: a second build of the library (e.g. of another commit, built in a git worktree) loaded instead
LIB_OVERRIDE = os.environ.get('TGT_HIP_LIB')
CSRC = os.path.join(_HERE, 'csrc')
# (source, extra flags, object suffix): the triplet attention kernels compile one dtype per translation unit
SOURCES = ['capi.hip', 'optimizer.hip', 'edge_gemm.hip', 'edge_wgrad.hip', 'params.hip', 'loss.hip', 'predict.hip', 'gaussian.hip', 'triplet_attention_proj.hip',
           ('triplet_attention.hip', ['-DTGT_TRI_INST=9'], '.f32'), ('triplet_attention.hip', ['-DTGT_TRI_INST=2'], '.bf16'),
           ('triplet_attention.hip', ['-DTGT_TRI_INST=4'], '.f16'), 'triplet_attention16.hip', 'triplet_attention_bwd2.hip', 'triplet_aggregate.hip', 'node_attention.hip', 'node_attention_mfma.hip', 'node_attention16.hip', 'node_attention_kb.hip', 'layernorm.hip', 'triangular_update.hip', 'elementwise.hip']
ABI_VERSION = 30

TGT_F32, TGT_BF16, TGT_F16 = 0, 1, 2
TRI_BIASED, TRI_GATED, TRI_MASK_OUT = 1, 2, 4

_i32, _i64, _f32, _vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class TripletAttentionArgs(C.Structure):
    _fields_ = [
        ('B', _i32), ('N', _i32), ('H', _i32), ('D', _i32),
        ('dtype', _i32), ('flags', _i32), ('scale', _f32), ('_pad0', _i32),
        ('qkv', _vp * 2), ('ld_qkv', _i64 * 2), ('q_off', _i32 * 2), ('k_off', _i32 * 2), ('v_off', _i32 * 2),
        ('eg', _vp * 2), ('ld_eg', _i64 * 2), ('e_off', _i32 * 2), ('g_off', _i32 * 2),
        ('mask', _vp),
        ('out', _vp), ('ld_out', _i64), ('o_off', _i32 * 2),
        ('d_out', _vp), ('d_qkv', _vp * 2), ('d_eg', _vp * 2),
        ('d_qkv_colsum', _vp * 2), ('d_eg_colsum', _vp * 2),
        ('dropout_p', _f32), ('_pad1', C.c_uint32), ('dropout_seed', C.c_uint64),
        ('ld_dqkv', _i64 * 2), ('ld_deg', _i64 * 2),
        ('graph_scale', _vp),
    ]


class TripletAggregateArgs(C.Structure):
    _fields_ = [
        ('B', _i32), ('N', _i32), ('H', _i32), ('D', _i32),
        ('dtype', _i32), ('flags', _i32),
        ('v', _vp * 2), ('ld_v', _i64 * 2), ('v_off', _i32 * 2),
        ('eg', _vp * 2), ('ld_eg', _i64 * 2), ('e_off', _i32 * 2), ('g_off', _i32 * 2),
        ('mask', _vp),
        ('out', _vp), ('ld_out', _i64), ('o_off', _i32 * 2),
        ('d_out', _vp), ('d_v', _vp * 2), ('d_eg', _vp * 2),
        ('dropout_p', _f32), ('_pad1', C.c_uint32), ('dropout_seed', C.c_uint64),
    ]


class NodeAttentionArgs(C.Structure):
    _fields_ = [
        ('B', _i32), ('N', _i32), ('H', _i32), ('D', _i32),
        ('dtype', _i32), ('scale_degree', _i32), ('logits_only', _i32), ('_pad0', _i32),
        ('scale', _f32), ('_pad1', _i32),
        ('qkv', _vp), ('ld_qkv', _i64), ('q_off', _i32), ('k_off', _i32), ('v_off', _i32), ('_pad2', _i32),
        ('eg', _vp), ('ld_eg', _i64), ('e_off', _i32), ('g_off', _i32),
        ('mask', _vp),
        ('vatt', _vp), ('hhat', _vp), ('lse', _vp), ('gsum', _vp),
        ('d_vatt', _vp), ('d_hhat', _vp), ('d_qkv', _vp), ('d_eg', _vp), ('_reserved0', _vp), ('hhat_scale', _vp),
    ]


class FuseRowsArgs(C.Structure):
    _fields_ = [
        ('n_rows', _i32), ('n_cols', _i32), ('n_src', _i32), ('src_dtype', _i32), ('fused_dtype', _i32), ('_pad0', _i32),
        ('row_src', _vp), ('row_idx', _vp), ('src', _vp * 8), ('src_bias', _vp * 8), ('fused', _vp), ('fused_bias', _vp),
    ]


class EdgeLinearArgs(C.Structure):
    _fields_ = [
        ('M', _i64), ('K', _i32), ('N', _i32), ('dtype', _i32), ('epilogue', _i32),
        ('a', _vp), ('lda', _i64), ('w', _vp), ('ldw', _i64), ('bias', _vp),
        ('gamma', _vp), ('beta', _vp), ('eps', _f32), ('colsum_rows', _i32),
        ('mean', _vp), ('rstd', _vp), ('y', _vp), ('ldy', _i64),
        ('out', _vp), ('ldo', _i64), ('out2', _vp), ('ldo2', _i64),
        ('res', _vp), ('ldr', _i64), ('ds_in', _vp), ('ld_ds', _i64),
        ('row_scale', _vp), ('out_scale', _vp), ('rows_per_sample', _i64),
        ('dropout_p', _f32), ('flags', C.c_uint32), ('dropout_seed', C.c_uint64),
        ('colsum_partial', _vp), ('dw_partial', _vp),
    ]


class SumItem(C.Structure):
    _fields_ = [('src', _vp), ('dst', _vp), ('planes', _i32), ('_pad', _i32), ('n', _i64)]


SUM_MANY_MAX = 64
EPI_BIAS, EPI_GELU, EPI_RESID, EPI_GELU_BWD, EPI_LN_BWD = range(5)
EDGE_BIAS_SCALED = 1


# symbol -> (restype, argtypes); every symbol include/tgt_hip.h declares
SYMBOLS = {
    'tgt_last_error': (C.c_char_p, []),
    'tgt_abi_version': (C.c_int, []),
    'tgt_triplet_attention_fwd': (C.c_int, [C.POINTER(TripletAttentionArgs), _vp]),
    'tgt_triplet_attention_bwd': (C.c_int, [C.POINTER(TripletAttentionArgs), _vp]),
    'tgt_triplet_attention_proj_supported': (C.c_int, [C.POINTER(TripletAttentionArgs), _i32]),
    'tgt_triplet_attention_proj_fwd': (C.c_int, [C.POINTER(TripletAttentionArgs), _vp, _i32, _vp, _vp, _vp]),
    'tgt_triplet_aggregate_fwd': (C.c_int, [C.POINTER(TripletAggregateArgs), _vp]),
    'tgt_triplet_aggregate_bwd': (C.c_int, [C.POINTER(TripletAggregateArgs), _vp]),
    'tgt_node_attention_fwd': (C.c_int, [C.POINTER(NodeAttentionArgs), _vp]),
    'tgt_node_attention_bwd': (C.c_int, [C.POINTER(NodeAttentionArgs), _vp]),
    'tgt_triangular_update_fwd': (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    'tgt_triangular_update_bwd': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    'tgt_gelu_dropout_fwd': (C.c_int, [_vp, _vp, _i64, _i32, _f32, C.c_uint64, _vp]),
    'tgt_gelu_dropout_bwd': (C.c_int, [_vp, _vp, _vp, _i64, _i32, _f32, C.c_uint64, _vp]),
    'tgt_gelu_dropout_scaled_fwd': (C.c_int, [_vp, _vp, _i64, _i32, _f32, C.c_uint64, _vp, _i64, _vp]),
    'tgt_gelu_dropout_scaled_bwd': (C.c_int, [_vp, _vp, _vp, _i64, _i32, _f32, C.c_uint64, _vp, _i64, _vp]),
    'tgt_gelu_colsum_parts': (C.c_int, []),
    'tgt_gelu_dropout_bwd_colsum': (C.c_int, [_vp, _vp, _vp, _i64, _i32, _f32, C.c_uint64, _vp, _i64, _i32, _vp, _vp, _vp]),
    'tgt_add_layer_norm_fwd': (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i64, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i64, _i32, _f32, _vp]),
    'tgt_add_layer_norm_bwd': (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    'tgt_layer_norm_parts': (C.c_int, []),
    'tgt_colsum': (C.c_int, [_vp, _i32, _i64, _i32, _vp, _vp, _vp]),
    'tgt_sum_rows': (C.c_int, [_vp, _i32, _i32, _vp, _vp]),
    'tgt_sum_planes': (C.c_int, [_vp, _i32, _i64, _vp, _vp]),
    'tgt_transpose_many': (C.c_int, [_vp, _i32, _i32, _vp]),
    'tgt_sum_many': (C.c_int, [C.POINTER(SumItem), _i32, _vp]),
    'tgt_set_seed_counter': (C.c_int, [_vp]),
    'tgt_cross_entropy_fwd': (C.c_int, [_vp, _i32, _vp, _i64, _i32, _vp, _vp, _vp]),
    'tgt_cross_entropy_bwd': (C.c_int, [_vp, _i32, _vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    'tgt_fuse_rows': (C.c_int, [C.POINTER(FuseRowsArgs), _vp]),
    'tgt_unfuse_rows': (C.c_int, [C.POINTER(FuseRowsArgs), _vp]),
    'tgt_permute_cols': (C.c_int, [_vp, _i32, _vp, _vp, _i32, _i32, _i32, _vp]),
    'tgt_layer_norm_fwd': (C.c_int, [_vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _i64, _i32, _f32, _vp]),
    'tgt_layer_norm_bwd': (C.c_int, [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i64, _i32, _vp]),
    'tgt_edge_linear_supported': (C.c_int, [C.POINTER(EdgeLinearArgs)]),
    'tgt_edge_linear_parts': (C.c_int, [_i64, _i32]),
    'tgt_edge_linear': (C.c_int, [C.POINTER(EdgeLinearArgs), _vp]),
    'tgt_edge_linear_set_grid_cap': (None, [_i32]),
    'tgt_adam_step': (C.c_int, [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _i32, _f32, _f32, _vp, _vp, _i32, _vp]),
    'tgt_grad_stats_parts': (C.c_int, []),
    'tgt_grad_scaler_step': (C.c_int, [_vp, _i64, _vp, _vp, _i32, _f32, _f32, _i32, _f32, _f32, _i32, _vp]),
    'tgt_loss_accumulate': (C.c_int, [_vp, _i32, _f32, _vp, _i32, _i32, _vp]),
    'tgt_gaussian_basis_parts': (C.c_int, [_i64]),
    'tgt_gaussian_basis_fwd': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp]),
    'tgt_gaussian_basis_bwd': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp]),
    'tgt_dist_bins_argmax': (C.c_int, [_vp, _i32, _i64, _i32, _i32, _vp, _i32, _i64, _i32, _vp, _vp]),
    'tgt_sample_commit': (C.c_int, [_vp, _i32, _vp]),
    'tgt_softmax_accumulate': (C.c_int, [_vp, _i32, _i64, _i32, _vp, _vp, _i32, _vp]),
    'tgt_probs_finish': (C.c_int, [_vp, _i64, _i32, _i32, _vp, _i32, _f32, _vp, _vp]),
    'tgt_gap_commit': (C.c_int, [_vp, _i32, _i32, _vp, _i32, _vp, _vp]),
    'tgt_pack_triu': (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i64, _vp]),
    'tgt_bins_to_dist': (C.c_int, [_vp, _i32, _i64, _i32, _vp, _i32, _f32, _i32, _i32, _vp, _vp]),
}
BINS_U8, BINS_U16, BINS_I32, BINS_I64, BINS_F32 = range(5)

_lib = None
_isa_lint = None


def isa_lint():
    """tools/isa_defuse_lint.py as a module (build-time only)."""
    global _isa_lint
    if _isa_lint is None:
        spec = importlib.util.spec_from_file_location(
            'isa_defuse_lint', os.path.join(os.path.dirname(_HERE), 'tools', 'isa_defuse_lint.py'))
        _isa_lint = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(_isa_lint)
    return _isa_lint


def build_library(force=False, verbose=False):
    """hipcc -> tgt_amd/libtgt_hip.so (gfx950).  Cross-compiles without a GPU."""
    units = [(s, [], '') if isinstance(s, str) else s for s in SOURCES]
    srcs = [os.path.join(CSRC, u[0]) for u in units]
    deps = srcs + [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith('.hpp')] + \
        [os.path.join(os.path.dirname(_HERE), 'include', 'tgt_hip.h')]
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    # the compile commands (flags included) and the compiler's identity are part of the library's currency, not only mtimes:
    # their hash is kept next to the library (libtgt_hip.stamp travels with the .so; the objects under build/ do not)
    def read(path):
        with open(path) as fh:
            return fh.read()

    try:
        hipcc_id = subprocess.run([hipcc, '--version'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode(errors='replace')
        have_hipcc = True
    except OSError:
        hipcc_id, have_hipcc = 'unknown', False
    lib_stamp = hashlib.sha256(repr([(u[0], u[1], u[2]) for u in units]).encode() + hipcc_id.encode()).hexdigest()
    stamp_path = LIB_PATH[:-3] + '.stamp'
    current = os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(d) for d in deps)
    if not force and current and os.path.exists(stamp_path) and read(stamp_path) == lib_stamp:
        return LIB_PATH
    if not have_hipcc:
        # no compiler on this box (a prebuilt library travelled here): nothing can be rebuilt, so a library that is newer than
        # every source is taken as it is -- lib() still checks its ABI version when it loads
        if current and not force:
            import warnings
            warnings.warn(f'{hipcc} not found: using the prebuilt {LIB_PATH} without its build stamp')
            return LIB_PATH
        raise RuntimeError(f'{hipcc} not found and {LIB_PATH} is missing or older than its sources: cannot build the HIP kernels')
    objs = []
    stamps = []
    procs = []
    build = os.path.join(_HERE, 'build')
    shared = max(os.path.getmtime(d) for d in deps[len(srcs):])        # headers: every unit depends on them
    for s, (_, flags, suffix) in zip(srcs, units):
        # one directory per unit: -save-temps=obj drops the device assembly next to the object, and the
        # three triplet_attention units share a source name
        d = os.path.join(build, os.path.basename(s) + suffix)
        o = os.path.join(d, 'unit.o')
        objs.append(o)
        cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-save-temps=obj', *flags, '-c', s, '-o', o]
        stamp = hashlib.sha256((' '.join(cmd) + '\n' + hipcc_id).encode()).hexdigest()
        stamp_file = os.path.join(d, 'unit.cmd')
        if not force and os.path.exists(o) and os.path.getmtime(o) >= max(shared, os.path.getmtime(s)) and \
                os.path.exists(stamp_file) and read(stamp_file) == stamp:
            continue                                                    # this unit's object is current
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(d)
        stamps.append((stamp_file, stamp))
        procs.append((cmd, d, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    undefined = {}
    bad_units = set()
    failed = None
    for cmd, d, p in procs:
        out, _ = p.communicate()
        if p.returncode:
            failed = failed or ('hipcc failed: ' + ' '.join(cmd) + '\n' + out.decode(errors='replace'))
            shutil.rmtree(d, ignore_errors=True)        # (no stale object, no temporaries left behind)
            continue
        if verbose and out:
            print(out.decode(errors='replace'))
        # the shipped ISA is linted for registers read but never written (a hipcc spill miscompile this
        # library has hit: tools/isa_defuse_lint.py); then the temporaries go, they are large
        for f in os.listdir(d):
            path = os.path.join(d, f)
            if f.endswith('-gfx950.s'):
                with open(path) as fh:
                    for k, regs in isa_lint().lint_all(fh.read()).items():
                        undefined[f'{os.path.basename(d)}:{k}'] = regs
                        bad_units.add(d)
            if f != 'unit.o' and not (f.endswith('-gfx950.s') and os.environ.get('TGT_KEEP_ISA')):
                os.remove(path)
    for stamp_file, stamp in stamps:
        if os.path.isdir(os.path.dirname(stamp_file)) and os.path.dirname(stamp_file) not in bad_units:
            with open(stamp_file, 'w') as fh:
                fh.write(stamp)
    if failed:
        raise RuntimeError(failed)
    for d in bad_units:                 # a unit that fails the lint must not be taken for current by the next build
        shutil.rmtree(d, ignore_errors=True)
    if undefined:
        raise RuntimeError('hipcc produced kernels that read vector registers no instruction writes (miscompiled spill?):\n' +
                           '\n'.join(f'  {k}: {v}' for k, v in undefined.items()))
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB_PATH] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode:
        raise RuntimeError('link failed: ' + ' '.join(cmd) + '\n' + r.stdout.decode(errors='replace'))
    with open(stamp_path, 'w') as fh:
        fh.write(lib_stamp)
    return LIB_PATH


def lib():
    """The loaded library; raises (loudly) when it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} is missing: the TGT HIP kernels are not built. Run '
                f'`python -c "import __graft_entry__ as g; g.build()"` (needs hipcc). '
                f'There is no CPU/eager fallback for the tgt_amd ops.')
        # torch bundles its own libamdhip64.so.7; it must be the process's HIP runtime BEFORE
        # this library is mapped, or two runtimes (two HSA instances) end up in one process.
        import torch  # noqa: F401
        L = C.CDLL(LIB_OVERRIDE or LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)          # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        if L.tgt_abi_version() != ABI_VERSION:
            raise RuntimeError(f'libtgt_hip.so ABI {L.tgt_abi_version()} != binding {ABI_VERSION}; rebuild')
        _lib = L
    return _lib


def check(code, what):
    if code != 0:
        msg = lib().tgt_last_error().decode(errors='replace')
        raise RuntimeError(f'{what} failed (code {code}): {msg}')
