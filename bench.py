#!/usr/bin/env python
"""Headline benchmark: graphs/sec of one full TGT-At 24L training step
(BASELINE.json `metric`; workload = configs[1]: batch 256 per GPU, N = 32,
bf16 autocast, random-init weights, synthetic PCQM-schema graphs).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = preprocess (edge mask, coordinate noise, distances) + forward + loss +
backward (+ RCCL all-reduce of gradient buckets) + Adam, exactly the sequence of
the reference's train loop (lib/training/training.py:439-470).  Inputs are
resident in HBM before the timed region.  One process per GPU; graphs are
independent, so each rank steps its own 256 graphs (weak scaling) and the only
exchange is the gradient all-reduce.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     : the dominant hand-written kernel (triplet attention backward or
                 forward), algorithmic HBM bytes / HIP-event time inside the timed steps
  cpu_baseline : the oracle (CPU restatement of the reference) timed on this
                 host on a bounded sample of the same workload (N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def node_algorithmic_bytes(kind, B, N, W, Hn, esz=2):
    """node attention (bias/softmax path), DESIGN.md §4 (node attention row): fwd reads Q,K,V (3NW), E,G (2N^2 Hn), writes
    V_att (NW), H_hat (N^2 Hn); bwd reads Q,K,V,V_att,dV_att (5NW), E,G,dH_hat (3N^2 Hn), writes dQ,dK,dV (3NW),
    dE,dG (2N^2 Hn); + mask."""
    n2 = N * N
    per_graph = ((4 * N * W + 3 * n2 * Hn) if kind == 'fwd' else (8 * N * W + 5 * n2 * Hn)) * esz + n2 * 4
    return B * per_graph


def _per_graph_bytes(kind, N, C, Ht, esz):
    """(all bytes, bytes still moved for a DropPath-dropped graph) of one graph in one launch, both directions"""
    n2 = N * N
    if kind == 'fwd':            # Q,K,V in + O out = 4 N^2 C, E,G = 2 N^2 Ht per direction, + mask; dropped: zero O rows
        return 2 * (4 * n2 * C + 2 * n2 * Ht) * esz + n2 * 4, 2 * (n2 * C) * esz
    if kind == 'fwd_proj':       # projection-fused forward (profiles/HISTORY_rounds_1-4.md §4.1a): X in ONCE (the kernel reads it 4 times: Q rows and K/V rows of
        #                          both directions), Q,K,V out (kept for the backward) + O out per direction, E,G in, + mask
        return (n2 * C + 2 * (4 * n2 * C + 2 * n2 * Ht)) * esz + n2 * 4, 2 * (n2 * C) * esz
    # bwd: Q,K,V,dO in + dQ,dK,dV out = 7 N^2 C, E,G in + dE,dG out = 4 N^2 Ht per direction, + mask; dropped: zero gradient rows
    return 2 * (7 * n2 * C + 4 * n2 * Ht) * esz + n2 * 4, 2 * (3 * n2 * C + 2 * n2 * Ht) * esz


def dropped_graph_discount(kind, N, C, Ht, drop_frac, esz=2):
    """factor on algorithmic_bytes for launches in which a fraction `drop_frac` of the graphs is DropPath-dropped and
    skipped by the kernel (tgt_triplet_attention_args.graph_scale): such a graph only has its zeros written"""
    full, moved = _per_graph_bytes(kind, N, C, Ht, esz)
    return 1.0 - drop_frac * (1.0 - moved / full)


def algorithmic_bytes(kind, B, N, C, Ht, esz=2):
    """HBM bytes one launch must move (both directions), SURVEY §8(d): see _per_graph_bytes"""
    return B * _per_graph_bytes(kind, N, C, Ht, esz)[0]


def cpu_baseline_worker(threads, micro=8, nodes=32, budget_s=40.0, full=False):
    """(runs in a subprocess) the oracle's TGT-At 24L training step on the host CPU (fp32), SURVEY 8(d): all physical
    cores (a quick calibration picks between them and 32 threads: oversubscribed hosts are slower with more), micro-batches
    of `micro` graphs of the same synthetic workload with gradient accumulation (1.1 GB of activations per graph
    forbid 256 at once), one Adam step at the end; fwd-only and fwd+loss+bwd graphs/s.  Bounded sample: as many
    micro-batches as fit the budget (>= 5 fwd+bwd); --cpu-baseline-full accumulates all 256 graphs of one GPU step."""
    from oracle import modules as om, core
    from tgt_amd.training.configs import tgt_at_24l
    from tgt_amd.training.synthetic import make_batch, batch_seed
    torch.manual_seed(0)
    model = om.TGT_Multi(**tgt_at_24l()).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)

    def batch_of(step, graphs):
        b = make_batch(graphs, nodes, batch_seed(step))
        nm = b['node_mask']
        b['edge_mask'] = nm.unsqueeze(-1) * nm.unsqueeze(-2)
        coords = core.smoothed_coord_noise(b['dft_coords'], b['edge_mask'], 0.2, 1.0)
        b['dist_input'] = core.pairwise_dist(coords)
        return b

    def loss_of(b):
        gap, logits = model(b)
        return torch.nn.functional.l1_loss(gap, b['target']) + 0.1 * core.binned_distance_xent(
            logits, core.pairwise_dist(b['dft_coords']), b['edge_mask'], 512, 8)

    def forward_only(b):
        with torch.no_grad():
            loss_of(b)

    cands = sorted({max(1, min(threads, 32)), threads})
    rates = {}
    probe = batch_of(0, 4)
    for t in cands:                                   # calibration (also the allocator / thread-pool warm-up)
        torch.set_num_threads(t)
        forward_only(probe)
        t0 = time.perf_counter()
        forward_only(probe)
        rates[t] = 4 / (time.perf_counter() - t0)
    used = max(rates, key=rates.get)
    torch.set_num_threads(used)

    t0, nf = time.perf_counter(), 0
    while nf < 2 or (nf < 4 and time.perf_counter() - t0 < 0.2 * budget_s):
        forward_only(batch_of(100 + nf, micro))
        nf += 1
    fwd_rate = micro * nf / (time.perf_counter() - t0)

    target = 256 // micro
    opt.zero_grad(set_to_none=True)
    t0, nb = time.perf_counter(), 0
    while nb < target and (nb < 5 or full or time.perf_counter() - t0 < 0.7 * budget_s):
        (loss_of(batch_of(200 + nb, micro)) * (micro / 256.0)).backward()          # accumulate towards the 256-graph batch
        nb += 1
    opt.step()
    dt = time.perf_counter() - t0
    print(json.dumps(dict(value=round(micro * nb / dt, 3), unit='graphs/s', cores=used, kind='port',
                          fwd_only_graphs_per_s=round(fwd_rate, 3), micro_batch=micro, micro_batches_timed=nb,
                          micro_batches_per_256_graph_step=target, thread_calibration={str(k): round(v, 2) for k, v in rates.items()},
                          sample=f'oracle TGT-At 24L train step on the host CPU, fp32, dropouts on: {nb} of the {target} '
                                 f'micro-batches ({micro} synthetic N={nodes} graphs each) of one 256-graph step, fwd+loss+bwd '
                                 f'with gradient accumulation + one Adam step, {used} threads '
                                 f'({threads} physical cores); fwd-only over {nf} micro-batches')), flush=True)


def cpu_baseline(timeout_s=280, full=False):
    """Bounded: a subprocess (off the timed region, GPU hidden) with a hard timeout."""
    import subprocess
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or os.cpu_count() or 1
    except Exception:
        phys = os.cpu_count() or 1
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-baseline-worker', str(phys)] + (['--cpu-baseline-full'] if full else [])
    env = dict(os.environ, HIP_VISIBLE_DEVICES='')
    env.pop('OMP_NUM_THREADS', None)
    if full:
        timeout_s = 1800
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
        line = [l for l in out.stdout.splitlines() if l.startswith('{')]
        if line:
            return json.loads(line[-1])
        return dict(value=None, unit='graphs/s', cores=phys, kind='port',
                    sample='cpu baseline worker failed: ' + out.stderr[-200:])
    except subprocess.TimeoutExpired:
        return dict(value=None, unit='graphs/s', cores=phys, kind='port',
                    sample=f'cpu baseline exceeded its {timeout_s}s bound')


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start one process per GPU of this node (the
    reference's execute.py:91-107 spawns its ranks the same way), each re-running this file with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as torch.distributed.run would; returns the first
    non-zero exit code.  Only rank 0 prints, and the children share this process's stdout."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env))
    rc = 0
    try:
        for p in procs:
            p.wait()
            rc = rc or p.returncode
            if p.returncode:                    # a rank died: do not leave the others blocked in a collective
                for q in procs:
                    if q.poll() is None:
                        q.terminate()
    finally:
        for q in procs:
            if q.poll() is None:
                q.kill()
    return rc


def launcher_selftest(rank, world):
    """the launcher path without a GPU: rendezvous over gloo, one all-reduce, rank 0 prints one line"""
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group(backend='gloo', rank=rank, world_size=world)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    dist.barrier()
    if rank == 0:
        print(json.dumps(dict(launcher_selftest=world, ranks_sum=float(t.item()))), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # (defaults = SURVEY 8(d): the mean over >= 50 steps after >= 10 warm-up steps; ~6 s of GPU time)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=256, help='graphs per GPU')
    ap.add_argument('--nodes', type=int, default=32)
    ap.add_argument('--ragged', action='store_true',
                    help='SURVEY 8(d) secondary: num_nodes ~ U{nodes/2..nodes} with the first graph at `nodes` (padded batch)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--settle-steps', type=int, default=64,
                    help='untimed steps BEFORE the warm-up steps that let the caching allocator reach its steady state (it keeps adding '
                         '12-54 MB segments on the node side stream for ~50 steps, then never again: tools/probes/mem_growth_probe.py); 0: none')
    ap.add_argument('--roofline-steps', type=int, default=3,
                    help='untimed single-stream steps after the timed region that measure the roofline kernels alone (0: use the timed region)')
    ap.add_argument('--cpu-baseline-worker', type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument('--cpu-baseline-full', action='store_true',
                    help='CPU baseline over all 256 graphs of one step (16 accumulated micro-batches; minutes)')
    ap.add_argument('--launcher-selftest', action='store_true', help=argparse.SUPPRESS)   # tests/test_bench_launcher.py
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'fp16', 'fp32'])
    ap.add_argument('--sync-every', type=int, default=0,
                    help='read the running mean loss on the host every K steps (K = 1: the reference loop, which pays a '
                         '.item() per step, tgt_training.py:153-157); 0 = never inside the timed steps (default)')
    ap.add_argument('--no-gemm-tuning', action='store_true', help='library default GEMM heuristics')
    ap.add_argument('--no-numa-bind', action='store_true', help='leave the CPU affinity of the rank alone (default: the NUMA node of its GPU)')
    ap.add_argument('--write-gemm-tuning', default='', help='tune online and write the TunableOp file here')
    ap.add_argument('--profile-all', action='store_true',
                    help='A/B: HIP events around EVERY hand-written launch, as rounds 1-4 did (host-bound: profiles/r06f_ab_events.txt)')
    ap.add_argument('--timing-probe', default='', choices=['', 'skip_sums', 'skip_proj_ln', 'skip_wgrad'],
                    help='leave work OUT of the step to bound what it costs (tools/probes/skip_probes.py); the line then carries '
                         '"value": null and "invalid": ... -- the gradients of such a step are wrong by construction')
    ap.add_argument('--grad-exchange', default='all_reduce', choices=['all_reduce', 'reduce_scatter', 'reduce_scatter_sharded'],
                    help="gradient exchange of the multi-rank step: bucketed all-reduce (default, the reference's DDP), reduce-scatter + all-gather per "
                         'bucket, or reduce-scatter + Adam on the owned 1/world slices + all-gather of the parameters (StepConfig.shard_optimizer)')
    ap.add_argument('--share-device', action='store_true',
                    help='multi-rank dress rehearsal on ONE GPU: every rank runs on cuda:0 and the ranks exchange over gloo (RCCL refuses '
                         'two ranks on one device).  Walks the exact multi-rank branch of this file -- rendezvous, rank-0 broadcast, '
                         'bucketed gradient exchange from the hooks, settle-step count, all_gather of the timings, per-rank NUMA binding, '
                         'per-rank TunableOp files -- on a 1-GPU box; the number it prints is NOT a scaling figure (config says so)')
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        cpu_baseline_worker(args.cpu_baseline_worker, full=args.cpu_baseline_full)
        return

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` by itself: one process per GPU, as the reference's launcher does
        # (execute.py:91-107); rank 0 prints the JSON line
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    # (dmabuf IPC is what the host driver of these boxes supports: without it RCCL's hipIpcGetMemHandle fails.  Exported in the image
    # already; set here too so that a launcher with a scrubbed environment still gets it -- it is read when the HIP runtime starts)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}; start it as '
                         f'`python bench.py --gpus N` (self-launching) or with torch.distributed.run --nproc-per-node N')
    if args.launcher_selftest:
        return launcher_selftest(rank, world)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the TGT kernels have no CPU path')
    if args.share_device:
        local_rank = 0                      # (every rank on the one GPU of the box)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    # this rank's threads on the socket its GPU hangs on (the CPU baseline below gets the whole host back)
    from tgt_amd.training.affinity import bind_to_gpu_numa
    host_mask = os.sched_getaffinity(0) if hasattr(os, 'sched_getaffinity') else None
    host_affinity = bind_to_gpu_numa(local_rank, enabled=not args.no_numa_bind)
    if world > 1 and args.share_device:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend='gloo', rank=rank, world_size=world)
    elif world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # RCCL on its own HIGH-PRIORITY stream: a bucket's all-reduce is dispatched ahead of the next workgroups of the backward
        # kernels it overlaps (two priority levels are all HIP offers; the node side stream uses the same one)
        opts = None
        try:
            opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
        except Exception:
            pass
        try:
            dist.init_process_group(backend='nccl', rank=rank, world_size=world, device_id=dev, pg_options=opts)
        except (TypeError, ValueError, RuntimeError):
            if dist.is_initialized():
                raise
            dist.init_process_group(backend='nccl', rank=rank, world_size=world, device_id=dev)     # (this torch does not take the options)

    from tgt_amd import ops
    from tgt_amd.pcqm import TGT_Multi
    from tgt_amd.training.configs import tgt_at_24l
    from tgt_amd.training.step import Trainer, StepConfig, preprocess_batch
    from tgt_amd.training.synthetic import make_batch, batch_seed
    if args.timing_probe:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools', 'probes'))
        import skip_probes
        skip_probes.install(args.timing_probe)

    if not args.no_gemm_tuning:
        from tgt_amd.training.gemm_tuning import enable_gemm_tuning
        if args.write_gemm_tuning:
            enable_gemm_tuning(online=True, filename=args.write_gemm_tuning, max_ms=150, max_iters=50)
        else:
            enable_gemm_tuning(online=True)

    mcfg = tgt_at_24l()
    torch.manual_seed(0)
    model = TGT_Multi(**mcfg).to(dev).train()
    torch.manual_seed(4321 + rank)           # same initial weights on every rank, different dropout streams
    cfg = StepConfig(mixed_precision=None if args.precision == 'fp32' else args.precision,
                     grad_exchange=args.grad_exchange.replace('_sharded', ''),
                     shard_optimizer=args.grad_exchange.endswith('_sharded') and world > 1)
    trainer = Trainer(model, cfg)

    # synthetic batches, resident in HBM before timing (4 distinct ones, cycled)
    pool = [{k: v.to(dev) for k, v in make_batch(args.batch, args.nodes, batch_seed(s, rank), ragged=args.ragged).items()}
            for s in range(4)]
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)

    def step(i):
        batch = preprocess_batch(pool[i % len(pool)], dev, cfg, training=True, generator=gen)
        out = trainer.training_step(batch)
        trainer.update_losses(out[1], batch)         # the reference's loop body (training.py:523-529), without its .item()
        return out

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Allocator settle: same steps as the warm-up, run until the caching allocator stops asking the driver for memory (at most
    # --settle-steps of them).  Not part of the contract's W warm-up steps and not timed; it only moves the allocator's own
    # start-up transient (synchronous hipMalloc calls during the first ~50 steps of any run) out of the timed region.
    settled = 0
    if args.settle_steps > 0:
        quiet, last = 0, torch.cuda.memory_stats(dev).get('num_device_alloc', 0)
        # (every step runs the ranks' gradient exchange: with more than one rank the COUNT must not depend on a rank's own allocator)
        while settled < args.settle_steps and (quiet < 16 or world > 1):
            step(settled)
            settled += 1
            now = torch.cuda.memory_stats(dev).get('num_device_alloc', 0)
            quiet = quiet + 1 if now == last else 0
            last = now
    for i in range(args.warmup):
        step(i)
    # The host queues a step ~20 ms ahead of the GPU.  A generation-2 collection of Python's garbage collector walks every tracked
    # object of the process -- 40-65 ms here (tools/probes/gc_probe.py), once every ~50 steps -- and when it lands on a thin
    # margin the stream runs dry: single steps of 120-130 ms in a 20-step region.  Freezing what exists after the warm-up (model,
    # trainer, batches: all of it lives for the whole run) keeps later collections to the few objects a step creates.  The
    # collector stays ON.
    import gc
    gc.collect()
    gc.freeze()
    # (events around the kernels the roofline leg reports, and only those: timing every launch costs the host ~3 us x 1400 per step)
    ROOFLINE_KERNELS = ('tgt_triplet_attention_bwd', 'tgt_triplet_attention_fwd', 'tgt_triplet_attention_proj_fwd',
                        'tgt_node_attention_fwd', 'tgt_node_attention_bwd')
    if args.profile_all:          # (A/B: events around every launch, as rounds 1-4 did)
        ROOFLINE_KERNELS = None
    prof = ops.profile_kernels(True, only=ROOFLINE_KERNELS, stride=1 if ROOFLINE_KERNELS is None else 5)
    fence()
    ms0 = torch.cuda.memory_stats(dev)
    # one event per step on the step's stream (GPU-side step boundaries: the spread of the steps, e.g. one stalled by a
    # synchronous device allocation, shows in step_ms below; `value` stays the wall time of the whole region)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    ahead = []                              # per step: how many steps the host was ahead of the GPU when it finished queueing (0..6)
    host_ms, lead = [], []                  # per step: host enqueue time; was the GPU still busy with the PREVIOUS step when the host finished queueing this one?
    n_comm_before = len(trainer._comm_events)
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        h0 = time.perf_counter()
        _, loss = step(args.warmup + i)
        marks[i + 1].record()
        host_ms.append((time.perf_counter() - h0) * 1e3)
        # (a non-blocking query: True = the GPU had already finished the previous step when the host finished queueing this one,
        #  i.e. the stream ran dry at some point of this step -- the host, not the GPU, set its duration)
        lead.append(bool(marks[i].query()))
        ahead.append(sum(0 if marks[j].query() else 1 for j in range(max(0, i - 5), i + 1)))     # step-end marks the GPU has not reached yet
        if args.sync_every and (i + 1) % args.sync_every == 0:
            trainer.mean_loss()                      # host read of the control block: synchronises this rank
    fence()
    dt = time.perf_counter() - t0
    seen_region = ops.profile_launch_counts()          # launches per kernel name inside the timed region (one in five carries events)
    step_gpu = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    step_ms = sorted(step_gpu)
    n_comm_timed = len(trainer._comm_events) - n_comm_before
    ms1 = torch.cuda.memory_stats(dev)
    ops.profile_kernels(False)
    # Kernel durations for `roofline`: inside the timed region the node channel runs on a second HIP stream and its kernels share
    # the CUs with the edge kernels, so an event pair around an edge launch also sees the other stream's work (and the wait for
    # CUs it holds).  A few more steps of the same loop, NOT timed, with everything on one stream give the kernel's own duration
    # -- the number `TGT_NODE_STREAM=0 rocprofv3 --kernel-trace --stats` of this command shows (profiles/).  Both are reported.
    prof_iso = None
    if ops.side_stream.enabled and args.roofline_steps > 0:
        ops.side_stream.enabled = False
        forked, ops._WGRAD_STREAM = ops._WGRAD_STREAM, False      # (the forked parameter-gradient stream as well: one stream, kernels alone)
        step(args.warmup + args.steps)                    # (one step for the allocator to settle on the new stream pattern)
        prof_iso = ops.profile_kernels(True, only=ROOFLINE_KERNELS, stride=1 if ROOFLINE_KERNELS is None else 5)
        for i in range(args.roofline_steps):
            step(args.warmup + args.steps + 1 + i)
        fence()
        ops.profile_kernels(False)
        ops.side_stream.enabled = True
        ops._WGRAD_STREAM = forked
    comm_ms = trainer.comm_exposed_ms()             # per step: what of the gradient exchange the backward did not hide
    comm_ms = comm_ms[n_comm_before:n_comm_before + n_comm_timed] if comm_ms else []       # (the timed steps only: not the roofline steps behind them)
    rank_ms = [dt / args.steps * 1e3]
    if world > 1:
        mine = torch.tensor([dt, sum(comm_ms) / max(1, len(comm_ms))], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_ms = [float(t[0]) / args.steps * 1e3 for t in every]
        rank_comm = [float(t[1]) for t in every]
        dt = max(float(t[0]) for t in every)
    else:
        rank_comm = [0.0]
    loss_val = float(loss.detach())

    if rank == 0:
        times_region = ops.kernel_times_ms(prof)
        times = ops.kernel_times_ms(prof_iso) if prof_iso is not None else times_region
        C, Ht = mcfg['edge_width'], mcfg['triplet_heads']
        esz = 4 if args.precision == 'fp32' else 2
        # DropPath ramps linearly 0 .. drop_path over the layers (tgt_amd/tgt/stack.py): the mean fraction of graphs a triplet launch
        # skips when the kernels are given the factors (TGT_TRI_SKIP: 1 = forward only, the default; 2 = backward too; 0 = off)
        skip_mode = os.environ.get('TGT_TRI_SKIP', '1')
        skip_mode = skip_mode if skip_mode in ('1', '2') else ''
        drop_frac = 0.5 * float(mcfg.get('drop_path', 0.0))
        cand = {}
        proj_on = bool(times.get('tgt_triplet_attention_proj_fwd'))     # (then the backward skips dropped graphs too: no Q/K/V rows)
        for name, kind in (('tgt_triplet_attention_bwd', 'bwd'), ('tgt_triplet_attention_fwd', 'fwd'),
                           ('tgt_triplet_attention_proj_fwd', 'fwd_proj')):
            if name in times and times[name]:
                avg_ms = sum(times[name]) / len(times[name])
                nbytes_k = algorithmic_bytes(kind, args.batch, args.nodes, C, Ht, esz)
                if skip_mode and (kind != 'bwd' or skip_mode == '2' or proj_on):
                    # the kernel does not move the bytes of the graphs DropPath drops: count what it moves (expected fraction)
                    nbytes_k = int(nbytes_k * dropped_graph_discount(kind, args.nodes, C, Ht, drop_frac, esz))
                cand[name] = (avg_ms * seen_region.get(name, len(times[name])), avg_ms, nbytes_k)      # (total time in the timed region: mean of the timed launches x all launches)
        # HBM bytes per launch from the PMC passes (collected offline with rocprofv3 --pmc, see
        # profiles/README.md); only valid for the shape they were measured at
        # The counters cannot be collected inside this run (rocprofv3 --pmc wraps the process), so `traffic` / `mfma_util` come
        # from a tracked summary -- but ONLY from one measured on exactly these kernel sources (`_kernel_src_sha`, written by
        # tools/pmc_summary.py) at this shape; anything else is reported as stale, not as a measurement.
        traffic, pmc, pmc_note = {}, {}, None
        n48 = args.batch == 128 and args.nodes == 48 and args.precision == 'bf16'          # BASELINE config 4's shape has its own counter pass (profiles/*_n48_pmc_summary.json)
        if ((args.batch == 256 and args.nodes == 32) or n48) and args.precision == 'bf16':
            sys.path.insert(0, os.path.join(ROOT, 'tools'))
            from pmc_summary import kernel_source_sha
            sha = kernel_source_sha(ROOT)
            summaries = sorted(f for f in os.listdir(os.path.join(ROOT, 'profiles')) if f.endswith('_pmc_summary.json') and (('_n48_' in f) == n48))     # (N = 48 passes: another shape)
            for f in reversed(summaries):
                cand_pmc = json.load(open(os.path.join(ROOT, 'profiles', f)))
                if cand_pmc.get('_kernel_src_sha') == sha:
                    pmc, pmc_note = cand_pmc, dict(file='profiles/' + f, kernel_src_sha=sha, match=True)
                    break
            if not pmc and summaries:
                pmc_note = dict(file='profiles/' + summaries[-1], kernel_src_sha=sha, match=False,
                                note='kernel sources changed since the last counter pass: traffic / mfma_util not quoted')
            # (tools/pmc_summary.py's short names of the mangled kernels; the round-4 backward lives in namespace bwd2)
            for short, name in (('tri_att_fwd_kernel', 'tgt_triplet_attention_fwd'), ('tri_att_bwd_kernel', 'tgt_triplet_attention_bwd'),
                                ('tri_att_bwd2_kernel', 'tgt_triplet_attention_bwd'),
                                ('tri_att_proj_fwd_kernel', 'tgt_triplet_attention_proj_fwd'),
                                # (N in 33..64: the 16-wide kernels of csrc/triplet_attention16.hip)
                                ('tri_att16_fwd_kernel', 'tgt_triplet_attention_fwd'), ('tri_att16_bwd_kernel', 'tgt_triplet_attention_bwd')):
                if short in pmc and 'hbm_bytes_per_launch' in pmc[short]:
                    traffic[name] = dict(traffic_bytes=pmc[short]['hbm_bytes_per_launch'])
        roofline = None
        if cand:
            name = max(cand, key=lambda k: cand[k][0])
            tot, avg_ms, nbytes = cand[name]
            ach = nbytes / (avg_ms * 1e-3) / 1e9
            roofline = dict(bound='hbm', kernel=name, achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit='GB/s',
                            frac=round(ach / HBM_PEAK_GBS, 4), traffic=traffic.get(name, {}).get('traffic_bytes'),
                            avg_launch_ms=round(avg_ms, 4), launches=seen_region.get(name, len(times[name])), launches_timed=len(times[name]),
                            algorithmic_bytes_per_launch=nbytes,
                            share_of_step=round(tot / (dt * 1e3), 4),
                            # matrix-core utilisation of this kernel from the SQ counter pass (offline, same shape):
                            # SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs); an HBM-bound core at 15.5 FLOP/B
                            mfma_util=pmc.get({'tgt_triplet_attention_bwd': next((k for k in ('tri_att_bwd2_kernel', 'tri_att16_bwd_kernel') if k in pmc), 'tri_att_bwd_kernel'),
                                               'tgt_triplet_attention_fwd': 'tri_att16_fwd_kernel' if 'tri_att16_fwd_kernel' in pmc else 'tri_att_fwd_kernel',
                                               'tgt_triplet_attention_proj_fwd': 'tri_att_proj_fwd_kernel'}[name], {}).get('mfma_util'),
                            other_kernels={k: dict(avg_launch_ms=round(v[1], 4),
                                                   achieved=round(v[2] / (v[1] * 1e-3) / 1e9, 1))
                                           for k, v in cand.items() if k != name})
            roofline['offline_pmc'] = pmc_note
            if prof_iso is not None and times_region.get(name):
                # The top-level achieved / frac / avg_launch_ms are the ones the headline throughput is made of: event pairs INSIDE
                # the timed region (the second stream's node kernels share the CUs there).  The kernel alone is under timing.alone.
                reg = sum(times_region[name]) / len(times_region[name])
                roofline['timing'] = dict(
                    top_level='in_timed_region',
                    alone=dict(avg_launch_ms=round(avg_ms, 4), achieved=round(ach, 1), frac=round(ach / HBM_PEAK_GBS, 4),
                               how=f'HIP events on the launch stream over {args.roofline_steps} extra steps after the timed region with the node '
                                   'channel and the parameter-gradient fork on the same stream (= TGT_NODE_STREAM=0 TGT_WGRAD_STREAM=0 '
                                   'rocprofv3 --stats of this command)'),
                    in_timed_region=dict(avg_launch_ms=round(reg, 4), achieved=round(nbytes / (reg * 1e-3) / 1e9, 1),
                                         frac=round(nbytes / (reg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                         note='event pairs inside the timed region: the second stream\'s node kernels share the CUs'))
                roofline.update(avg_launch_ms=round(reg, 4), achieved=round(nbytes / (reg * 1e-3) / 1e9, 1),
                                frac=round(nbytes / (reg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), launches=seen_region.get(name, len(times_region[name])),
                                launches_timed=len(times_region[name]))
                roofline['share_of_step'] = round(reg * seen_region.get(name, len(times_region[name])) / (dt * 1e3), 4)
            if skip_mode and drop_frac > 0:
                roofline['droppath_skip'] = dict(kernels='forward+backward' if (skip_mode == '2' or proj_on) else 'forward',
                                                 expected_dropped_fraction=drop_frac,
                                                 note='algorithmic bytes of the skipping kernels count only what they move for a dropped graph')
            # the bias/softmax path (node attention with edge bias and gate), same accounting
            for kname, kind in (('tgt_node_attention_fwd', 'fwd'), ('tgt_node_attention_bwd', 'bwd')):
                if times.get(kname):
                    avg = sum(times[kname]) / len(times[kname])
                    nb = node_algorithmic_bytes(kind, args.batch, args.nodes, mcfg['node_width'], mcfg['num_heads'], esz)
                    roofline['other_kernels'][kname] = dict(avg_launch_ms=round(avg, 4),
                                                            achieved=round(nb / (avg * 1e-3) / 1e9, 1),
                                                            frac=round(nb / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
                    # (one launch each way: the key-blocked forward of csrc/node_attention_kb.hip when H is a multiple of 32, the matrix-core
                    # kernels of csrc/node_attention_mfma.hip for N <= 32, the 16-wide tiles of csrc/node_attention16.hip for 33..64)
                    if kind == 'fwd':
                        shorts = ('node_att_kb_fwd_kernel',) if mcfg['num_heads'] % 32 == 0 else \
                                 ('node_att_mfma_fwd_kernel',) if args.nodes <= 32 else ('node_att16_fwd_kernel',)
                    else:
                        shorts = ('node_att_mfma_bwd_kernel',) if args.nodes <= 32 else ('node_att16_bwd_kernel',)
                    if all(k in pmc and 'hbm_bytes_per_launch' in pmc[k] for k in shorts):
                        roofline['other_kernels'][kname]['traffic'] = sum(pmc[k]['hbm_bytes_per_launch'] for k in shorts)
                        roofline['other_kernels'][kname]['algorithmic_bytes_per_launch'] = nb
        out = dict(
            metric='graphs/sec training step, TGT-At 24L PCQM batch 256, 1/2/4/8 MI355X',
            value=round(args.batch * world * args.steps / dt, 2), unit='graphs/s',
            n_gpus=world, steps=args.steps, warmup=args.warmup,
            # what actually ran: the process group's size and backend as torch.distributed reports them (1 / None without one)
            world=dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1,
            rccl_ranks=dist.get_world_size() if (dist.is_available() and dist.is_initialized() and dist.get_backend() == 'nccl') else 0,
            ms_per_step=round(dt / args.steps * 1e3, 3), higher_is_better=True, scaling='weak',
            vs_baseline=None, dtype=args.precision, data='synthetic',
            config=dict(workload='TGT-At 24L (TGT_Multi, 103.6M params, 512 dist bins) train step; '
                                 f'{args.batch} synthetic N={args.nodes} graphs per GPU' + (f' (ragged: num_nodes ~ U{{{max(1, args.nodes // 2)}..{args.nodes}}}, padded)' if args.ragged else '') +
                                 '; dropouts of tgt_at_tp.yaml on',
                        global_batch=args.batch * world, nodes=args.nodes,
                        parallelism=f'dp{world}' + (' (REHEARSAL: all ranks share one GPU, gloo exchange; not a scaling figure)' if args.share_device else ''),
                        precision=f'{args.precision} autocast, fp32 params/Adam',
                        **({'host_loss_read_every': args.sync_every} if args.sync_every else {})),
            # the exchange, so that a scaling curve explains itself: per-rank step time, and the milliseconds per step the step's
            # stream sat waiting for the last gradient buckets after the backward had ended (0 with one rank: nothing to exchange)
            ms_per_step_by_rank=[round(v, 3) for v in rank_ms],
            comm_exposed_ms=round(max(rank_comm), 3), comm_exposed_ms_by_rank=[round(v, 3) for v in rank_comm],
            grad_exchange=dict(buckets=(len(trainer.buckets) if trainer.buckets else 0), bucket_mbytes=cfg.bucket_mbytes,
                               launch_order_last_step=trainer.bucket_order[:16], mode=cfg.grad_exchange, optimizer_sharded=bool(trainer.sharded),
                               wire_dtype=cfg.grad_comm_dtype or 'fp32',
                               rccl_stream=('high priority' if (world > 1 and not args.share_device) else None)),
            final_loss=round(loss_val, 5),
            # {} = the default path; bench-side A/B flags of this command line ride along
            knobs_not_default=dict(__import__('tgt_amd.knobs', fromlist=['K']).K.non_default(),
                                   **({'--profile-all': True} if args.profile_all else {})),
            host_affinity=host_affinity,
            step_ms=dict(min=round(step_ms[0], 3), median=round(step_ms[len(step_ms) // 2], 3), max=round(step_ms[-1], 3),
                         note='GPU-side duration of each timed step (events on the step stream)',
                         # every step slower than 1.2x the median, with what the host was doing: its enqueue time for that step and
                         # whether the stream had run dry (the GPU finished the previous step before the host finished queueing this one)
                         stragglers=[dict(step=i, gpu_ms=round(step_gpu[i], 2), host_enqueue_ms=round(host_ms[i], 2), stream_ran_dry=lead[i])
                                     for i in range(args.steps) if step_gpu[i] > 1.2 * step_ms[len(step_ms) // 2]][:8],
                         host_enqueue_ms=dict(median=round(sorted(host_ms)[len(host_ms) // 2], 2), max=round(max(host_ms), 2)),
                         steps_stream_ran_dry=sum(lead),
                         host_lead_steps=dict(min=min(ahead), median=sorted(ahead)[len(ahead) // 2], max=max(ahead),
                                              per_step=ahead[:64],
                                              note='step-end marks (of the last 6) the GPU had not reached when the host finished queueing a step: 0 = the stream ran dry')),
            # conditions of the timed region a plain Trainer loop does not get by itself: stated, not hidden
            timed_region_policy=dict(gc_frozen=True, allocator_settle='up to --settle-steps untimed steps until 16 in a row make no device allocation',
                                     numa_bound=bool(host_affinity.get('bound')), per_kernel_events='roofline kernels only, one launch in five'),
            roofline=roofline,
            # the caching allocator inside the timed region: device allocations / frees there are synchronous driver calls
            allocator_settle_steps=settled,
            memory=dict(reserved_GB=round(ms1.get('reserved_bytes.all.current', 0) / 1e9, 2),
                        peak_allocated_GB=round(ms1.get('allocated_bytes.all.peak', 0) / 1e9, 2),
                        device_allocs_in_timed_region=ms1.get('num_device_alloc', 0) - ms0.get('num_device_alloc', 0),
                        device_frees_in_timed_region=ms1.get('num_device_free', 0) - ms0.get('num_device_free', 0),
                        alloc_retries=ms1.get('num_alloc_retries', 0)),
        )
        if world == 1 and not args.no_cpu_baseline:
            if host_mask is not None:
                from tgt_amd.training.affinity import restore_affinity
                restore_affinity(host_mask)                 # the oracle's step runs on all of the host's cores again (every thread)
            out['cpu_baseline'] = cpu_baseline(full=args.cpu_baseline_full)
        if args.timing_probe:
            # work was left out of the step: the milliseconds bound what that work costs, the throughput is not one
            out.update(value=None, invalid=f'timing probe {args.timing_probe}: work left out of the step, gradients wrong by construction',
                       graphs_per_s_with_work_missing=round(args.batch * world * args.steps / dt, 2))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
