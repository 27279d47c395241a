"""`python bench.py --gpus 2 --share-device`: the multi-rank branch of bench.py end to end on ONE GPU.
RCCL with more than one rank has never run on the boxes this repo is built on (1 GPU each); the first 8-GPU contact is the
driver's.  This walks everything of that branch that does not need a second device: self-launch of the ranks
(reference lib/training/execute.py:91-107), rendezvous on 127.0.0.1, rank-0 parameter broadcast, bucketed gradient exchange from
the autograd hooks, the settle-step count that must not depend on a rank's own allocator, per-rank NUMA binding, per-rank
TunableOp files, the all_gather of the timings, and the JSON line of rank 0."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *args], capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]               # rank 0 only
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize('launcher', ['self', 'torchrun', 'self_sharded'])
def test_bench_two_ranks_share_one_gpu(launcher):
    common = ['--gpus', '2', '--share-device', '--batch', '16', '--steps', '3', '--warmup', '1', '--settle-steps', '3',
              '--roofline-steps', '1', '--no-cpu-baseline']
    if launcher == 'self_sharded':                          # the sharded-optimizer exchange (StepConfig.shard_optimizer) through the same branch
        out = _bench(common + ['--grad-exchange', 'reduce_scatter_sharded'])
        assert out['grad_exchange']['optimizer_sharded'] is True and out['grad_exchange']['mode'] == 'reduce_scatter'
    elif launcher == 'self':
        out = _bench(common)
    else:
        # the driver's form: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...
        import socket
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
        env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                            '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'bench.py'), *common],
                           capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
        lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
        assert len(lines) == 1, r.stdout[-2000:]
        out = json.loads(lines[0])
    assert out['world'] == 2 and out['n_gpus'] == 2 and out['rccl_ranks'] == 0            # gloo: a rehearsal, and the line says so
    assert 'REHEARSAL' in out['config']['parallelism'] and out['config']['global_batch'] == 32
    assert len(out['ms_per_step_by_rank']) == 2 and all(v > 0 for v in out['ms_per_step_by_rank'])
    assert len(out['comm_exposed_ms_by_rank']) == 2
    assert out['grad_exchange']['buckets'] >= 2 and sorted(out['grad_exchange']['launch_order_last_step']) == \
        list(range(out['grad_exchange']['buckets']))[:16] or out['grad_exchange']['buckets'] > 16
    assert out['allocator_settle_steps'] == 3              # every rank ran the same count (it drives collectives)
    assert out['value'] > 0 and abs(out['value'] - 32 * 3 / (out['ms_per_step'] * 3e-3)) < 1e-3 * out['value'] + 1
    assert out['roofline'] is not None and out['final_loss'] == out['final_loss']        # a number, not NaN
