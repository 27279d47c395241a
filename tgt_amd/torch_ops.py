"""torch dispatcher registration of the hot-path ops.

`load()` builds (if needed) and loads `tgt_amd/libtgt_torch_ops.so` with `torch.ops.load_library`; afterwards

    torch.ops.tgt.egt_attention(qkv, eg, mask, num_heads, scale_degree, want_edges) -> (V_att, H_hat)
    torch.ops.tgt.triplet_attention(qkv_in, eg_in, qkv_out, eg_out, mask, num_heads) -> Va
    torch.ops.tgt.triplet_aggregate(v_in, eg_in, v_out, eg_out, mask, num_heads, mask_out) -> Va

are differentiable ops (C++ autograd nodes) and `*_fwd` / `*_bwd` the raw kernels, all on top of the C ABI of
libtgt_hip.so (`csrc/torch_ops.cpp`): outputs come from the caching allocator inside the op, the launch goes to the
current stream of the tensors' device, errors are `RuntimeError`s.  The modules of `tgt_amd.tgt` call the same kernels
through `tgt_amd.ops` (ctypes, with the fused-row layouts and the bias-gradient plumbing of the training step); this
library is the seam for callers that want plain torch custom ops, e.g. the reference's own
`lib/tgt/layers/triplet.py:213-246` / `layers.py:62-77` einsum chains replaced in place.
"""
import hashlib
import os
import subprocess

from . import _lib

_HERE = os.path.dirname(os.path.abspath(__file__))
OPS_LIB_PATH = os.path.join(_HERE, 'libtgt_torch_ops.so')
_SRC = os.path.join(_HERE, 'csrc', 'torch_ops.cpp')
_loaded = False


def build_op_library(force=False):
    """g++ -> tgt_amd/libtgt_torch_ops.so (host code only; links libtgt_hip.so and the torch libraries)."""
    import torch
    from torch.utils import cpp_extension as ce
    _lib.build_library()
    gemm_src = os.path.join(os.path.dirname(_SRC), 'gemm_dispatch.cpp')      # cached-descriptor dispatch of the library GEMMs (tgt_amd/gemm.py)
    deps = [_SRC, gemm_src, os.path.join(os.path.dirname(_HERE), 'include', 'tgt_hip.h'), _lib.LIB_PATH]
    tlib = os.path.join(os.path.dirname(torch.__file__), 'lib')
    cxx = os.environ.get('CXX', 'g++')
    cmd = [cxx, '-O2', '-std=c++17', '-fPIC', '-shared', '-D__HIP_PLATFORM_AMD__=1', '-DUSE_ROCM=1',
           f'-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}']
    cmd += [f'-I{p}' for p in ce.include_paths()] + ['-I/opt/rocm/include', _SRC, gemm_src, '-o', OPS_LIB_PATH,
                                                    f'-L{tlib}', '-lc10', '-lc10_hip', '-ltorch_cpu', '-ltorch_hip', '-ltorch',
                                                    '-l:libhipblaslt.so', '-l:librocblas.so',      # torch's OWN copies (the tuned indices are theirs)
                                                    f'-L{_HERE}', '-l:libtgt_hip.so', '-Wl,-rpath,$ORIGIN', f'-Wl,-rpath,{tlib}']
    # the command line, the compiler and the torch build are part of the library's currency (not only mtimes)
    try:
        cxx_id = subprocess.run([cxx, '--version'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode(errors='replace')
    except OSError:
        cxx_id = 'unknown'
    stamp = hashlib.sha256((' '.join(cmd) + '\n' + cxx_id + torch.__version__).encode()).hexdigest()
    stamp_path = OPS_LIB_PATH[:-3] + '.stamp'
    if not force and os.path.exists(OPS_LIB_PATH) and os.path.getmtime(OPS_LIB_PATH) >= max(os.path.getmtime(d) for d in deps) and \
            os.path.exists(stamp_path) and open(stamp_path).read() == stamp:
        return OPS_LIB_PATH
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode:
        raise RuntimeError('building libtgt_torch_ops.so failed: ' + ' '.join(cmd) + '\n' + r.stdout.decode(errors='replace'))
    with open(stamp_path, 'w') as fh:
        fh.write(stamp)
    return OPS_LIB_PATH


def load():
    """Register the `tgt::` ops with this process's torch dispatcher (idempotent); raises when the library is missing."""
    global _loaded
    if not _loaded:
        import torch
        _lib.lib()                     # libtgt_hip.so first (and torch's HIP runtime before it)
        if not os.path.exists(OPS_LIB_PATH):
            raise RuntimeError(f'{OPS_LIB_PATH} is missing: run `python -c "import __graft_entry__ as g; g.build()"`')
        torch.ops.load_library(OPS_LIB_PATH)
        if torch.ops.tgt.abi_version() != _lib.ABI_VERSION:
            raise RuntimeError('libtgt_torch_ops.so was built against another libtgt_hip.so ABI; rebuild')
        _loaded = True
    import torch
    return torch.ops.tgt
