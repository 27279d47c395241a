import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no GPU is visible and they were
    not explicitly selected away with -m 'not gpu'."""
    # the RCCL-with-two-ranks tests first: on a box with >= 2 GPUs a broken multi-rank path fails within the first seconds of
    # the run instead of after the 900 single-GPU tests (they skip at once on the 1-GPU boxes)
    first = [it for it in items if 'over_rccl' in it.name]
    if first:
        rest = [it for it in items if 'over_rccl' not in it.name]
        items[:] = first + rest
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session', autouse=True)
def _pinned_gemm_selection():
    """several parity comparisons go through library GEMMs, whose summation order follows the hipBLASLt / TunableOp
    selection.  On the GPU box the session runs on the SHIPPED table (tgt_amd/tuning/tunableop_gfx950.csv, what bench.py loads),
    offline -- no online tuning, so a shape that is not in the table takes the library's default and nothing is timed inside a
    test -- which makes the stated tolerances a statement about one GEMM selection instead of whichever the box would pick."""
    try:
        import torch
        if torch.cuda.is_available() and os.environ.get('TGT_TEST_PIN_GEMMS', '1') != '0':
            from tgt_amd.training.gemm_tuning import enable_gemm_tuning
            enable_gemm_tuning(online=False)
    except Exception as e:                  # (a torch without TunableOp: the tests still run, on the library defaults)
        print(f'conftest: GEMM selection not pinned ({e})')
    yield


@pytest.fixture(autouse=True)
def _parity_log_scope():
    import parity_log
    parity_log.new_test()
    yield


def pytest_sessionfinish(session, exitstatus):
    import parity_log
    parity_log.dump()
