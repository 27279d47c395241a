"""Measured parity errors, on request: with TGT_PARITY_LOG=<file> every rel-L2 a GPU parity test computes is recorded together
with the test id and the dtype whose tolerance the test looked up last; tests/conftest.py writes the summary at session end
(per dtype: the largest error over all cases and the case it came from; per test function: the largest error per dtype).
`profiles/parity_errors.json` is that file for the tree it was measured on; the stated tolerances in tests/test_hip_ops.py are set
from it (the headroom is recorded, not assumed)."""
import json
import os

_LOG = []


class Tol(dict):
    """the tolerance table of a test module; remembers the dtype of the last look-up so that rel() can file its value"""
    last = None

    def __getitem__(self, key):
        Tol.last = str(key).replace('torch.', '')
        return dict.__getitem__(self, key)


def record(value):
    if os.environ.get('TGT_PARITY_LOG'):
        _LOG.append((os.environ.get('PYTEST_CURRENT_TEST', '?').split(' ')[0], Tol.last, float(value)))
    return value


def new_test():
    Tol.last = None


def dump():
    path = os.environ.get('TGT_PARITY_LOG')
    if not path or not _LOG:
        return
    by_dtype, by_func = {}, {}
    for test, dt, v in _LOG:
        if v != v:
            continue
        d = by_dtype.setdefault(dt or 'unknown', dict(max=0.0, case=None, n=0))
        d['n'] += 1
        if v > d['max']:
            d['max'], d['case'] = v, test
        func = test.split('::')[-1].split('[')[0]
        f = by_func.setdefault(func, {})
        f[dt or 'unknown'] = max(f.get(dt or 'unknown', 0.0), v)
    out = dict(note='rel-L2 = |hip - oracle| / |oracle| of every comparison the GPU parity tests made (gradients included: their '
                    'stated tolerance is 2x the forward one); max over all cases per dtype, and per test function',
               comparisons=len(_LOG), max_by_dtype=by_dtype, max_by_test=by_func)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
