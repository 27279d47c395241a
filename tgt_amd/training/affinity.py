"""Host-side placement of a rank: the CPUs of the NUMA node its GPU hangs on.

One process per GPU (reference lib/training/execute.py:91-107 spawns them without any placement).  On the two-socket hosts
of the MI355X boxes a rank whose threads run on the other socket pays a cross-socket hop for every doorbell write and
every pinned-memory access; the enqueue path of the training step (67-75 ms of host time per step) is what that
slows.  Best effort: anything missing (sysfs entries, a container without the topology, an affinity mask already
narrowed by the launcher) leaves the process as it was and says so.
"""
import os


def _cpulist(text):
    cpus = set()
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_node(device_index):
    """NUMA node of a visible GPU (None when the platform does not tell)"""
    import torch
    p = torch.cuda.get_device_properties(device_index)
    try:
        bdf = '%04x:%02x:%02x.0' % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        with open(f'/sys/bus/pci/devices/{bdf}/numa_node') as f:
            node = int(f.read().strip())
    except (AttributeError, OSError, ValueError):
        return None
    return node if node >= 0 else None


def _bind_all_threads(cpus):
    """sched_setaffinity on EVERY thread of this process.  On Linux the call binds one thread (pid 0 = the caller); threads that
    already exist -- torch's intra-op / OpenMP pools are created at import -- keep their old mask, threads started later inherit
    the caller's.  Returns (threads bound, threads seen)."""
    try:
        tids = [int(t) for t in os.listdir('/proc/self/task')]
    except OSError:
        tids = []
    done = 0
    for tid in tids:
        try:
            os.sched_setaffinity(tid, cpus)
            done += 1
        except OSError:
            pass                      # (a thread that exited meanwhile)
    os.sched_setaffinity(0, cpus)     # the caller last: what new threads inherit
    return done, len(tids)


def restore_affinity(cpus):
    """give every thread of the process the mask `cpus` back (bench.py: the CPU baseline runs on the whole host)"""
    if hasattr(os, 'sched_setaffinity') and cpus:
        try:
            _bind_all_threads(cpus)
        except OSError:
            pass


def bind_to_gpu_numa(device_index, enabled=True):
    """Restrict EVERY existing thread of this process (and, by inheritance, the ones it starts from now on) to the CPUs of the
    GPU's NUMA node.  Returns a small report for the benchmark line: {'numa_node', 'cpus', 'threads', 'bound'} or
    {'bound': False, 'why': ...}."""
    if not enabled:
        return {'bound': False, 'why': 'disabled'}
    if not hasattr(os, 'sched_setaffinity'):
        return {'bound': False, 'why': 'no sched_setaffinity on this platform'}
    node = gpu_numa_node(device_index)
    if node is None:
        return {'bound': False, 'why': 'the GPU reports no NUMA node'}
    try:
        with open(f'/sys/devices/system/node/node{node}/cpulist') as f:
            cpus = _cpulist(f.read())
    except (OSError, ValueError):
        return {'bound': False, 'why': f'no cpulist for node {node}'}
    allowed = os.sched_getaffinity(0)
    target = cpus & allowed
    if not target:
        return {'bound': False, 'why': f'none of node {node}\'s CPUs is in the current mask'}
    if target == allowed:
        return {'bound': True, 'numa_node': node, 'cpus': len(target), 'note': 'mask already inside the node'}
    try:
        done, seen = _bind_all_threads(target)
    except OSError as e:
        return {'bound': False, 'why': f'sched_setaffinity: {e}'}
    return {'bound': True, 'numa_node': node, 'cpus': len(target), 'threads': f'{done}/{seen}'}
