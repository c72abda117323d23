"""Headline benchmark: full LBP sweeps/s on a 10M-edge hybrid MRF (BASELINE.json metric).

Workload (cfg 4 x10): random sparse hybrid pairwise MRF, V = 2.5M variables (80 %
continuous on [-10,10], 20 % binary), F = 5M pairwise factors, E = 10M edges, 10 % evidence; EPBP with n = 64
particles, T = 32 integral points, 'simple' proposal, counter-based device sampler.  One "step" = one full flooding
sweep = v2f + proposal update + resample + f2v (everything EPBP.run does per iteration, EPBPLogVersion.py:245-285).
Inputs are generated on the host, uploaded once, and resident in HBM before the timed region.

N > 1: the same graph is edge-sharded over the ranks, ONE RCCL all-to-all per sweep (lhvi/dist.py); total work is fixed, so
scaling is "strong".  Two splits are built, timed and freed in the same launch -- owner computes (variables partitioned, every
message computed where its target lives; bit-identical to one GPU; `value` is quoted on it) and factor-partitioned (boundary
partial sums) -- and both are printed under `exchanges` with their phase times (`--exchange` / `--also` choose).  Every rank
names its phases on stderr; a phase that outlasts LHVI_BENCH_PHASE_TIMEOUT (600 s) or a collective that outlasts
LHVI_BENCH_COLLECTIVE_TIMEOUT (300 s) ends the run non-zero with its name.

``python3 bench.py --gpus N`` with N > 1 and no WORLD_SIZE in the environment starts its own ranks: the parent (which
never touches the GPU) runs ``python -m torch.distributed.run --nproc-per-node N`` on this same file and exits with
its return code.  Under ``rocprofv3`` call it as ``-- python3 bench.py ...`` (no ``env`` / shell hop after the profiler
has initialised the GPU).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

# multi-process GPU work on this pool needs dmabuf IPC (the launcher's environment normally carries this already)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, 'lifted-hybrid-variational-inference_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s HBM3E
FP64_PEAK_TFLOPS = 78.6        # fp64 vector peak
HBM_TARGET_FRAC = 0.40         # BASELINE.json: ">= 40 % of per-GPU HBM-read roofline"
FLOP_PER_TERM = 16.0           # algorithmic: 7 fma + 1 add + 1 ldexp per (output point, partner particle) term
# executed at the particles since the floor form of the term loop (round 3): 5 fma + 1 add + 1 fract + 1 ldexp
FLOP_PER_PARTICLE_TERM = 13.0
# executed cost of a term at the integral points when the heavy kernel tabulates them along the uniform grid: one
# multiplication, 1.25 additions of the lane reduce-scatter (10 per lane and batch of 8 points) and the lane's two table
# exponentials (2 x 16 flop) spread over its T = 32 points
FLOP_PER_GRID_TERM = 1.0 + 1.25 + 2 * 16.0 / 32.0


def algorithmic_bytes_per_edge(n, T):
    """v2f 16n + f2v 8(2n+T) + proposal 8T+16, + 12 B of indices"""
    return {'v2f': 16 * n, 'f2v': 8 * (2 * n + T), 'proposal': 8 * T + 16, 'index': 12,
            'sweep': 32 * n + 16 * T + 16 + 12}


TRAFFIC_PROFILE = 'profiles/r05_final_traffic.json'     # the PMC summary of THIS round's build (scripts/profile_round.sh r05_final)


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes of this same command -- exactly the file
    named above (written by scripts/summarize_pmc.py with the gfx950 FETCH_SIZE correction), never "whichever file sorts
    last" -- and that file's name; (None, None) if it is absent or does not list the kernel.  PMC passes cannot be collected
    inside a timed run, so this number is read, not measured live."""
    path = os.path.join(ROOT, TRAFFIC_PROFILE)
    try:
        d = json.load(open(path))
    except Exception:
        return None, None
    for k, v in d.items():
        if k.endswith(kernel):
            return v['hbm_bytes'], TRAFFIC_PROFILE
    return None, None


def progress(msg):
    """one flushed line per phase on stderr (with the time since start), so that a run that does not finish shows where it
    stopped; stdout carries only the JSON line"""
    sys.stderr.write('[bench %8.2f s] %s\n' % (time.perf_counter() - _T0, msg))
    sys.stderr.flush()


_T0 = time.perf_counter()
_PHASE = {'name': 'start', 'since': time.perf_counter(), 'limit': None}


def phase(name, limit_s=None):
    """name the phase this rank is in (one flushed line, rank tagged) and arm the watchdog for it: a phase that lasts longer than
    `limit_s` (default: LHVI_BENCH_PHASE_TIMEOUT, 600 s) makes the process say which one and exit 124 -- a stuck collective ends
    the run with a non-zero code and its name instead of hanging the node"""
    _PHASE.update(name=name, since=time.perf_counter(), limit=limit_s)
    progress('rank %s: %s' % (os.environ.get('RANK', '0'), name))


def start_watchdog():
    import threading
    default = float(os.environ.get('LHVI_BENCH_PHASE_TIMEOUT', '600'))

    def watch():
        while True:
            time.sleep(1.0)
            limit = _PHASE['limit'] or default
            waited = time.perf_counter() - _PHASE['since']
            if waited > limit:
                sys.stderr.write('[bench] rank %s STUCK in phase "%s" for %.0f s (limit %.0f s): exiting 124\n'
                                 % (os.environ.get('RANK', '0'), _PHASE['name'], waited, limit))
                sys.stderr.flush()
                os._exit(124)
    threading.Thread(target=watch, daemon=True).start()


def cpu_baseline(n, T, seconds_target=15.0):
    """The CPU oracle (port of the reference's sweep, oracle/c/pbp_oracle.c) on a bounded sample of the same
    workload: same generator, smaller V; all host cores through OpenMP on the f2v half."""
    from lhvi import synth
    from oracle import oracle
    # a one-GPU box owns a 16-core share of the host; more OpenMP threads than that only oversubscribe it
    cores = min(len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1), 16)
    os.environ.setdefault('OMP_NUM_THREADS', str(cores))
    V = 10000
    flat = synth.hybrid_mrf_flat(V=V, deg=4, seed=123, T=T)
    o = oracle.PbpOracle(flat, n, ep=False, epbp=True, var_threshold=3)
    rng = np.random.default_rng(0)
    draw = lambda: np.clip(rng.normal(0.0, np.sqrt(5.0), size=(flat.V, n)), -10, 10)
    o.init()
    o.set_particles(draw())
    sweeps, t0 = 0, time.perf_counter()
    while True:
        o.step_v2f()
        o.step_proposal()
        o.set_particles(draw())
        o.step_f2v()
        sweeps += 1
        dt = time.perf_counter() - t0
        if dt > seconds_target or sweeps >= 50:
            break
    edge_rate = 2.0 * flat.E * sweeps / dt
    return flat.E, sweeps, dt, edge_rate, cores


def python_baseline(n, T, budget_s=10.0):
    """The reference-equivalent pure-Python path (oracle/pyref.py: dict-of-dicts restatement of EPBPLogVersion.py:225-289,
    pinned against the reference's golden vectors in the CPU suite), single thread, on 1e3-, 1e4- and 1e5-edge graphs of the
    same generator; the f -> rv half is bounded to `budget_s` seconds of factors and extrapolated linearly
    (labelled)."""
    from lhvi import graph, potentials, synth
    from oracle import pyref
    out = []
    for V, budget in ((250, budget_s), (2500, budget_s * 0.6), (25000, budget_s * 0.5)):
        flat = synth.hybrid_mrf_flat(V=V, deg=4, seed=123, T=T)
        out.append(pyref.time_sweep(flat, n, graph, potentials, budget_s=budget))
    return out


def self_launch(n_ranks):
    """`--gpus N` without a launcher: start N ranks of this file under torch.distributed.run from a parent that has not
    touched the GPU (and never will); returns the launcher's exit code (non-zero if any rank failed)"""
    import subprocess
    # --standalone: the launcher picks and binds its own rendezvous port (no bind / close / reuse race with other processes)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--standalone', '--local-addr', '127.0.0.1', '--nnodes=1',
           '--nproc-per-node', str(n_ranks), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, LHVI_BENCH_SELF_LAUNCHED='1')
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--edges', type=int, default=10_000_000)
    ap.add_argument('--particles', type=int, default=64)
    ap.add_argument('--grid', type=int, default=32)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--exchange', choices=('pairs', 'owner', 'ownercompute'), default=None,
                    help='multi-GPU runs, the split that `value` is quoted on (default: LHVI_BENCH_EXCHANGE or ownercompute): '
                         'ownercompute: VARIABLES partitioned, every message computed where its target lives, v->f rows of cut '
                         'edges + ghost proposals in one collective (bit-identical to one GPU); pairs: factors partitioned, boundary '
                         'rows between every pair of ranks sharing a variable (one collective); owner: those rows reduced at an owner '
                         'rank and sent back (two smaller collectives)')
    ap.add_argument('--also', default=None,
                    help='multi-GPU runs: comma-separated other splits to build, time and free after the first one in the same '
                         'launch (default: the other of ownercompute / pairs; "none" to skip); all of them are printed under '
                         '"exchanges" with their phases_ms')
    ap.add_argument('--proposal', choices=('simple', 'EP'), default='simple',
                    help="proposal rule of the sweep: 'simple' is what BASELINE.json's metric is quoted on; 'EP' is the reference's "
                         "default (EPBPLogVersion.py:20)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit('--gpus must be at least 1')
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))             # before anything imports torch or touches the GPU
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        raise SystemExit('--gpus %d does not match WORLD_SIZE=%d of the launcher' % (args.gpus, world))
    primary = args.exchange or os.environ.get('LHVI_BENCH_EXCHANGE', 'ownercompute')
    if args.also is None:
        also = [x for x in ('ownercompute', 'pairs') if x != primary][:1]
    else:
        also = [x for x in args.also.split(',') if x and x != 'none' and x != primary]
    for x in [primary] + also:
        if x not in ('pairs', 'owner', 'ownercompute'):
            raise SystemExit('unknown exchange %r' % x)
    start_watchdog()

    import datetime
    import torch
    from lhvi import _abi, synth, dist
    from lhvi.pbp import EPBP

    # LHVI_DIST_BACKEND=gloo rehearses the multi-process path on a box with fewer GPUs than ranks (ranks then share
    # devices and the exchange is staged through the host); the real runs use nccl = RCCL over xGMI, one rank per GPU
    backend = os.environ.get('LHVI_DIST_BACKEND', 'nccl')
    n_dev = torch.cuda.device_count()
    if backend == 'nccl' and local_rank >= n_dev:
        raise SystemExit('rank %d has no GPU of its own (%d visible): one rank per GPU over RCCL; LHVI_DIST_BACKEND=gloo '
                         'rehearses more ranks than GPUs' % (local_rank, n_dev))
    device_index = local_rank % max(n_dev, 1) if backend == 'gloo' else local_rank
    torch.cuda.set_device(device_index)
    if world > 1:
        import torch.distributed as td
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # a collective that does not complete within this time is aborted by the process group's watchdog (RCCL) or raises
        # (gloo): the run then ends non-zero instead of hanging; the phase watchdog above names where it stood
        limit = datetime.timedelta(seconds=float(os.environ.get('LHVI_BENCH_COLLECTIVE_TIMEOUT', '300')))
        phase('init_process_group (%s)' % backend)
        if backend == 'nccl':
            td.init_process_group('nccl', device_id=torch.device('cuda', device_index), timeout=limit)
        else:
            td.init_process_group(backend, timeout=limit)

    phase('rank %d of %d on device %d: building the graph' % (rank, world, device_index))
    n, T = args.particles, args.grid
    deg = 4
    V = args.edges // deg
    flat = synth.hybrid_mrf_flat(V=V, deg=deg, seed=0, T=T)     # identical on every rank (seeded)
    E_total = flat.E

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as td
            td.barrier()
        torch.cuda.synchronize()

    def build(exchange):
        if world == 1:
            bp = EPBP(None, n=n, proposal_approximation=args.proposal, sampler='device', seed=1)
            bp._setup(None, flat=flat)
            return dist.SingleRunner(bp)
        if exchange == 'ownercompute':
            phase('%s: variable partition on rank 0 + broadcast' % exchange)
            owner = dist.broadcast_variable_partition(flat, rank, world)
            phase('%s: this rank\'s plan and work lists' % exchange)
            return dist.OwnerRunner(flat, n=n, seed=1, rank=rank, world=world, proposal_approximation=args.proposal, var_owner=owner)
        # the factor partition (one breadth-first sweep of the whole graph) is computed on rank 0 only and broadcast; every
        # rank then builds just its own slice of the plan
        phase('%s: factor partition on rank 0 + broadcast' % exchange)
        fac_owner = dist.broadcast_partition(flat, rank, world)
        phase('%s: this rank\'s plan and work lists' % exchange)
        return dist.ShardedRunner(flat, n=n, seed=1, rank=rank, world=world, proposal_approximation=args.proposal,
                                  fac_owner=fac_owner, owner_reduce=exchange == 'owner')

    def time_runner(runner, label):
        """W untimed sweeps, then exactly K sweeps bracketed by barrier + synchronize on both sides; MAX over ranks"""
        phase('%s: init' % label)
        runner.init()
        phase('%s: %d warm-up sweeps' % (label, args.warmup))
        for _ in range(args.warmup):
            runner.sweep()
        barrier()
        if hasattr(runner, 'record_phases'):
            runner.record_phases = True
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        phase('%s: %d timed sweeps' % (label, args.steps))
        t0 = time.perf_counter()
        for i in range(args.steps):
            runner.sweep(f2v_events=ev[i])
        barrier()
        elapsed = time.perf_counter() - t0
        phase('%s: %d timed sweeps done in %.3f s on this rank; max over ranks' % (label, args.steps, elapsed))
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda' if backend == 'nccl' else 'cpu')
        if world > 1:
            import torch.distributed as td
            td.all_reduce(t, op=td.ReduceOp.MAX)
        elapsed = float(t.item())
        f2v_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))     # HIP events on the launch stream
        # sharded runs may launch the dominant kernel twice per sweep (interior edges while the exchange is in flight, then the rest)
        f2v_ms += float(sum(a.elapsed_time(b) for a, b in getattr(runner, 'f2v_extra', []))) / args.steps
        return elapsed, f2v_ms

    runner = build(primary if world > 1 else None)
    elapsed, f2v_ms = time_runner(runner, primary if world > 1 else 'single GPU')
    # single GPU: the dominant kernel shares the CUs with the short f -> v kernels on a second stream (lhvi/pbp.py, overlap_f2v); a few
    # sweeps on ONE stream after the timed region give its duration alone, reported beside the timed one (never in `value`)
    f2v_alone_ms = None
    bp0 = getattr(runner, 'bp', None)
    if world == 1 and bp0 is not None and getattr(bp0, 'overlap_f2v', False) and os.environ.get('LHVI_PBP_OVERLAP') != '0' \
            and getattr(bp0, 'n_heavy', 0) >= getattr(bp0, 'overlap_min_heavy', 1 << 62):
        phase('single GPU: five sweeps on one stream (the dominant kernel alone)')
        bp0.overlap_f2v = False
        ev1 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for i in range(5):
            runner.sweep(f2v_events=ev1[i])
        torch.cuda.synchronize()
        bp0.overlap_f2v = True
        f2v_alone_ms = float(np.mean([a.elapsed_time(b) for a, b in ev1[1:]]))
    exchanges = None
    if world > 1:
        def line(r, el):
            return {'value': args.steps / el, 'unit': 'sweeps/s', 'ms_per_step': 1e3 * el / args.steps,
                    'phases_ms': r.phase_ms() if getattr(r, 'phase_ms', None) else None}
        exchanges = {primary: line(runner, elapsed)}
        for other in also:
            # build, time, free: the other split of the same graph in the same launch, so that one node run compares them.
            # (a failure here must not cost the run its line: `value` is already measured)
            try:
                r2 = build(other)
                el2, _ = time_runner(r2, other)
                exchanges[other] = line(r2, el2)
                del r2
            except Exception as exc:                      # noqa: BLE001 -- reported in the line, and on stderr with the phase
                progress('rank %d: %s FAILED: %s: %s' % (rank, other, type(exc).__name__, exc))
                exchanges[other] = {'error': '%s: %s' % (type(exc).__name__, str(exc)[:300])}
            torch.cuda.empty_cache()
    del flat
    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        sweeps_per_s = args.steps / elapsed
        bytes_e = algorithmic_bytes_per_edge(n, T)
        E_local = runner.local_edges()
        # dominant kernel = pbp_f2v_heavy_kernel (continuous target, continuous or observed partner): its edges read the
        # partner's particles + message (16n B), the target's points (8(n+T) B) and write n+T log-messages -- the f2v figure
        # of DESIGN.md section 4 plus the target points, per edge of ITS work list
        heavy_edges, terms = runner.heavy_stats()
        f2v_bytes = (bytes_e['f2v'] + 8 * (n + T)) * heavy_edges
        f2v_gbs = f2v_bytes / (f2v_ms * 1e-3) / 1e9
        hidden_frac = runner.work_fraction()
        # fp64 work.  Algorithmic: 16 flop per (output point, partner particle) term (SURVEY 8(d) / DESIGN 4.7).  Executed:
        # the terms at the particles cost FLOP_PER_PARTICLE_TERM; the terms at the integral points of edges the kernel serves by
        # the grid recurrence cost FLOP_PER_GRID_TERM (the rest of them take the direct form, like the particles)
        grid_terms = runner.heavy_grid_terms()
        # ... as counted by the kernel itself: words 8 / 9 of the ticket buffer hold the edges of the last launch that went through
        # the recurrence / failed its range guard (the guard is data dependent)
        grid_edges = fallback_edges = None
        ticket = getattr(getattr(runner, 'bp', None), 'f2v_ticket', None)
        if ticket is not None and world == 1:
            stat = ticket.cpu().numpy()
            grid_edges, fallback_edges = int(stat[8]), int(stat[9])
            if grid_edges + fallback_edges > 0:
                grid_terms = int(round(grid_terms * grid_edges / float(grid_edges + fallback_edges)))
        f2v_tflops = terms * FLOP_PER_TERM / (f2v_ms * 1e-3) / 1e12
        exec_tflops = ((terms - grid_terms) * FLOP_PER_PARTICLE_TERM + grid_terms * FLOP_PER_GRID_TERM) / (f2v_ms * 1e-3) / 1e12
        sweep_gbs = bytes_e['sweep'] * E_local / (ms * 1e-3) / 1e9
        # what the 0.40 HBM target of BASELINE.json means at THIS particle count.  The sweep's kernels run one after another: the
        # dominant f->v kernel cannot end before max(its flops at the fp64 peak, its bytes at the HBM peak), the rest of the sweep
        # (v->f, proposal, sampler, the other f->v kernels) not before its bytes at the HBM peak; the sweep's algorithmic bytes over
        # that shortest possible time is the highest HBM fraction the arithmetic allows
        sweep_bytes = float(bytes_e['sweep'] * E_local)
        t_f2v = max(terms * FLOP_PER_TERM / (FP64_PEAK_TFLOPS * 1e12), f2v_bytes / (HBM_PEAK_GBS * 1e9))
        t_min = t_f2v + max(0.0, sweep_bytes - f2v_bytes) / (HBM_PEAK_GBS * 1e9)
        intensity = terms * FLOP_PER_TERM / float(f2v_bytes)
        hbm_ceiling = min(1.0, sweep_bytes / t_min / (HBM_PEAK_GBS * 1e9))
        if hbm_ceiling < HBM_TARGET_FRAC:
            hbm_note = ('n = %d: the dominant f->v kernel needs %.1f fp64 flop per algorithmic byte (machine balance %.1f); with it at the '
                        'fp64 vector peak and every other kernel at the HBM peak the sweep takes %.2f ms and its HBM fraction is %.2f, so the '
                        '%.2f target is out of reach by arithmetic intensity, not by wasted traffic'
                        % (n, intensity, FP64_PEAK_TFLOPS * 1e3 / HBM_PEAK_GBS, 1e3 * t_min, hbm_ceiling, HBM_TARGET_FRAC))
        else:
            hbm_note = ('n = %d: %.1f fp64 flop per algorithmic byte in the dominant f->v kernel (machine balance %.1f); with it at its '
                        'roof and every other kernel at the HBM peak the sweep would take %.2f ms, an HBM fraction of %.2f, so the '
                        'arithmetic does NOT excuse missing the %.2f target here: the f->v few-particle kernel is at its vector-issue '
                        'ceiling (94 %% busy) and the fused per-variable kernel (v->f, proposal, sampler in one pass) is bound by its own '
                        'instruction count and its partial-sector writes, not by where the rows lie -- a build that reads a variable\'s rows '
                        'contiguously is no faster (profiles/r05_experiments.md items 12 and 16)'
                        % (n, intensity, FP64_PEAK_TFLOPS * 1e3 / HBM_PEAK_GBS, 1e3 * t_min, hbm_ceiling, HBM_TARGET_FRAC))
        traffic, traffic_src = measured_traffic('pbp_f2v_heavy_kernel') if world == 1 and args.edges == 10_000_000 else (None, None)
        # single GPU: the short f -> v kernels run on a second stream beside the dominant one (lhvi/pbp.py, overlap_f2v), which then
        # shares the CUs with them: its launch duration -- what `achieved` is computed from -- includes that sharing
        bp_ = getattr(runner, 'bp', None)
        concurrent = None
        if bp_ is not None and getattr(bp_, 'overlap_f2v', False) and os.environ.get('LHVI_PBP_OVERLAP') != '0' \
                and getattr(bp_, 'n_heavy', 0) >= getattr(bp_, 'overlap_min_heavy', 1 << 62):
            concurrent = ('pbp_f2v_pair_kernel + pbp_f2v_generic_kernel on a second stream beside this kernel, which leaves a workgroup per CU '
                          'free (LHVI_PBP_SHARE_CUS); alone it takes ~0.6 ms less, the sweep ~0.55 ms more (profiles/r05_experiments.md item 11)')
        out = {
            'metric': 'lbp_sweeps_per_sec_10M_edge_hybrid_mrf' if args.edges == 10_000_000
                      else 'lbp_sweeps_per_sec_%d_edge_hybrid_mrf' % args.edges, 'value': sweeps_per_s, 'unit': 'sweeps/s',
            'edge_messages_per_sec': 2.0 * E_total * sweeps_per_s,
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms,
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'cfg4x10 random hybrid pairwise MRF, EPBP particle sweep', 'edges': E_total,
                       'variables': V, 'particles': n, 'integral_points': T, 'proposal': args.proposal,
                       'sharding': 'single GPU' if world == 1 else
                                   ('variable-partitioned (owner computes), 1 all_to_all/sweep of cut-edge v->f rows + ghost proposals, overlapped with the interior part'
                                    if primary == 'ownercompute' else
                                    'factor-partitioned edge shards, 1 all_to_all/sweep overlapped with the interior part'),
                       'exchange': None if world == 1 else primary},
            # the dominant kernel is compute bound (~40 flop per algorithmic byte): its roof is the fp64 VECTOR peak (the
            # kernel issues VALU FMAs; the term -- rank-2 outer product + exp -- has nothing for MFMA to do, and on gfx950
            # the fp64 matrix path shares the vector fp64 pipe anyway).  The HBM view BASELINE.json asks for is in 'hbm' /
            # 'sweep_hbm' with the 0.40 target beside it.
            'roofline': {'bound': 'fp64_valu', 'kernel': 'pbp_f2v_heavy_kernel', 'achieved': f2v_tflops, 'peak': FP64_PEAK_TFLOPS,
                         'unit': 'TFLOP/s', 'frac': f2v_tflops / FP64_PEAK_TFLOPS,
                         'executed': {'achieved': exec_tflops, 'frac': exec_tflops / FP64_PEAK_TFLOPS,
                                      'grid_terms_per_launch': grid_terms, 'flop_per_grid_term': FLOP_PER_GRID_TERM,
                                      'grid_edges_counted_by_the_kernel': grid_edges, 'guard_fallback_edges': fallback_edges,
                                      'flop_per_particle_term': FLOP_PER_PARTICLE_TERM,
                                      'note': 'flops the kernel really issues: %d per term at the particles (5 fma, add, fract, '
                                              'ldexp), %.2f per term at the integral points tabulated by the grid recurrence'
                                              % (FLOP_PER_PARTICLE_TERM, FLOP_PER_GRID_TERM)},
                         'traffic': traffic, 'traffic_source': ('from_committed_profile: ' + traffic_src) if traffic_src else None,
                         'kernel_ms': f2v_ms, 'edges_per_launch': heavy_edges, 'joint_terms_per_launch': terms,
                         'concurrent': concurrent,
                         'alone': None if f2v_alone_ms is None else {'kernel_ms': f2v_alone_ms, 'frac': terms * FLOP_PER_TERM / (f2v_alone_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                                                                        'note': 'the same kernel in sweeps on one stream, after the timed region'},
                         'note': 'achieved = algorithmic count, 16 flop per (output point, partner particle) term, over the HIP-event '
                                 'time of the launch.  Terms at the particles cost 8 fp64 + 2 int32 issue slots (36 cycles per '
                                 'wave-term) + 2 LDS reads (36.8 LDS-pipe cycles per four wave-terms of a CU) each: both ceilings sit at '
                                 '~0.88 of this peak in algorithmic flops; terms at the integral points are tabulated along the uniform '
                                 'grid by one multiplication each plus a lane reduce-scatter (DESIGN.md sections 4.3, 5)',
                         'hbm': {'achieved': f2v_gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': f2v_gbs / HBM_PEAK_GBS,
                                 'algorithmic_bytes_per_launch': f2v_bytes},
                         'sweep_hbm': {'achieved': sweep_gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': sweep_gbs / HBM_PEAK_GBS},
                         'hbm_target': HBM_TARGET_FRAC,
                         'hbm_target_met': bool(sweep_gbs / HBM_PEAK_GBS >= HBM_TARGET_FRAC),
                         'hbm_target_ceiling': hbm_ceiling, 'hbm_target_note': hbm_note},
            'hidden_edge_fraction': hidden_frac,
        }
        if getattr(runner, 'phase_ms', None):
            out['phases_ms'] = runner.phase_ms()      # sharded runs: HIP-event times of pack / interior / exchange wait / boundary
        if exchanges is not None:
            out['exchanges'] = exchanges              # every split timed in this launch (`value` is the first one's)
        if not args.no_cpu_baseline and world == 1:
            phase('CPU baseline: C oracle on a bounded sample')
            Es, sw, dt, edge_rate, cores = cpu_baseline(n, T)
            phase('CPU baseline: pure-Python restatement at 1e3 / 1e4 / 1e5 edges')
            py = python_baseline(n, T)
            out['cpu_baseline'] = {'value': edge_rate / (2.0 * E_total), 'unit': 'sweeps/s', 'cores': cores, 'kind': 'port',
                                   'edge_messages_per_sec': edge_rate,
                                   'sample': '%d sweeps of the C oracle (OpenMP, %d threads) on a %d-edge graph from the '
                                             'same generator in %.1f s; value = measured edge-message rate / (2 * %d edges)'
                                             % (sw, cores, Es, dt, E_total),
                                   'python': {'value': py[0]['edge_messages_per_sec'] / (2.0 * E_total), 'unit': 'sweeps/s', 'cores': 1,
                                              'kind': 'port (pure-Python dict-of-dicts restatement of EPBPLogVersion.py:225-289, '
                                                      'oracle/pyref.py; the reference itself cannot travel to this box)',
                                              'edge_messages_per_sec': py[0]['edge_messages_per_sec'],
                                              'sample': 'one sweep at n = %d, T = %d on %d-, %d- and %d-edge graphs of the same generator; '
                                                        'the f -> rv half timed on the first %d / %d / %d hidden edges and extrapolated '
                                                        'linearly; value = edge-message rate of the first / (2 * %d edges)'
                                                        % (n, T, py[0]['edges'], py[1]['edges'], py[2]['edges'], py[0]['f2v_edges_done'],
                                                           py[1]['f2v_edges_done'], py[2]['f2v_edges_done'], E_total),
                                              'runs': py}}
        print(json.dumps(out))
        sys.stdout.flush()
    phase('done')
    if world > 1:
        import torch.distributed as td
        phase('destroy_process_group', 60)
        td.destroy_process_group()


if __name__ == '__main__':
    main()
