"""Library-GEMM selection for the plain GEMMs of the step.

The projections / FFN GEMMs are left to hipBLASLt/rocBLAS (through torch), but
their default heuristics pick poor kernels for the tall-skinny TGT shapes
(M = B*N*N = 262144, K,N in 64..1600): e.g. 286 us vs 58 us for 262144x256x256.
PyTorch's TunableOp times the library's candidate solutions per shape; the
winners for the BASELINE shapes on MI355X are shipped in
tgt_amd/tuning/tunableop_gfx950.csv and loaded here, new shapes are tuned online
during warm-up.
"""
import os

import torch

TUNING_FILE = os.environ.get('TGT_TUNING_FILE') or \
    os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tuning', 'tunableop_gfx950.csv')      # (the variable: A/B of a re-tuned table)


def enable_gemm_tuning(online=True, filename=None, max_ms=50, max_iters=20):
    """filename=None: start from the shipped table (a private copy, because TunableOp rewrites
    its file at exit when online tuning found new shapes); a filename: tune and write there."""
    import shutil
    import tempfile
    t = torch.cuda.tunable
    t.enable(True)
    if filename is None:
        filename = os.path.join(tempfile.gettempdir(), f'tgt_tunableop_{os.getpid()}.csv')
        if os.path.exists(TUNING_FILE):
            shutil.copyfile(TUNING_FILE, filename)
    t.set_filename(filename, insert_device_ordinal=False)
    t.tuning_enable(bool(online))
    t.set_max_tuning_duration(int(max_ms))
    t.set_max_tuning_iterations(int(max_iters))
    return t
