"""`python bench.py --gpus N` must start its own ranks (reference execute.py:91-107 spawns one process per
GPU from one command); exercised here on CPU over gloo with the hidden --launcher-selftest mode."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=180):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *args], capture_output=True, text=True,
                          timeout=timeout, env=env)


def test_self_launch_world2_gloo():
    r = _run(['--gpus', '2', '--launcher-selftest'])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout                       # rank 0 only
    out = json.loads(lines[0])
    assert out == {'launcher_selftest': 2, 'ranks_sum': 3.0}


def test_world_size_mismatch_is_a_clear_error():
    r = _run(['--gpus', '2', '--launcher-selftest'], {'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode != 0 and 'WORLD_SIZE=1' in (r.stderr + r.stdout)
    assert 'AssertionError' not in r.stderr
