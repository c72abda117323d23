"""soak of the two multi-GPU splits against the single-GPU sweep on random graphs: `world` simulated ranks on one GPU (loopback
exchange), random hybrid MRFs (size, degree, share of discrete variables, evidence, grid size), particle counts, proposal rules.
Owner-computes: proposals, particles and both message arrays bit for bit; factor-partitioned pairs exchange: to rounding (remote
partial sums are added as a block).  usage: python tests/soak/soak_dist_random.py [first seed] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'lifted-hybrid-variational-inference_amd')]
import numpy as np, torch
from lhvi import _abi, dist, synth
from lhvi.pbp import EPBP


def owner_compute(flat, n, world, approx, sweeps):
    bp = EPBP(None, n=n, proposal_approximation=approx, sampler='device', seed=3)
    bp._setup(None, flat=flat)
    single = dist.SingleRunner(bp)
    single.init()
    owner = dist.partition_variables(flat, world)
    group = dist.LoopbackGroup(world)
    runners = [dist.OwnerRunner(flat, n=n, seed=3, rank=r, world=world, proposal_approximation=approx, group=group, var_owner=owner) for r in range(world)]
    for r in runners:
        r.init()
    dev = bp.q_dev.device
    for it in range(sweeps):
        single.sweep()
        sends = [r.owned_half() for r in runners]
        for r, s in zip(runners, sends):
            group.post(r.rank, s, r.counts)
        for r in runners:
            r.interior()
        for r in runners:
            r.boundary(group.collect(r.rank, None))
        for r in runners:
            plan = r.plan
            gid = torch.from_numpy(plan.var_gid).to(dev)
            own = torch.from_numpy((np.arange(plan.flat.V) < plan.n_owned) & plan.flat.var_hidden).to(dev)
            loc = torch.from_numpy((np.arange(plan.flat.V) < plan.n_owned + plan.n_ghost) & plan.flat.var_hidden).to(dev)
            live = torch.from_numpy(np.arange(n)[None, :] < r.bp.np_host[:, None]).to(dev)
            assert torch.equal(r.bp.q_dev[own], bp.q_dev[gid][own]), 'q of owned variables (sweep %d, rank %d)' % (it, r.rank)
            cont = torch.from_numpy(plan.flat.var_cont).to(dev)
            assert torch.equal(r.bp.q_dev[loc & cont], bp.q_dev[gid][loc & cont]), 'q of ghosts'
            assert torch.equal(torch.where(live, r.bp.particles, 0.0)[loc], torch.where(live, bp.particles[gid], 0.0)[loc]), 'particles'
            eid = torch.from_numpy(plan.edge_ids).to(dev)
            mine = torch.from_numpy(~plan.edge_skip).to(dev)
            le = live[torch.from_numpy(plan.flat.edge_var.astype(np.int64)).to(dev)]
            assert torch.equal(torch.where(le, r.message_rows(), 0.0)[mine], torch.where(le, bp.v2f[eid], 0.0)[mine]), 'v2f'
            assert torch.equal(r.bp.f2v[mine], bp.f2v[eid][mine]), 'f2v'
    assert torch.isfinite(bp.q_dev[torch.from_numpy(flat.var_hidden).to(dev)]).all()


def pairs(flat, n, world, approx, sweeps, two_part):
    bp = EPBP(None, n=n, proposal_approximation=approx, sampler='device', seed=3)
    bp._setup(None, flat=flat)
    single = dist.SingleRunner(bp)
    single.init()
    group = dist.LoopbackGroup(world)
    runners = [dist.ShardedRunner(flat, n=n, seed=3, rank=r, world=world, proposal_approximation=approx, group=group) for r in range(world)]
    for r in runners:
        r.init()
    for it in range(sweeps):
        single.sweep()
        two = two_part and all(r.overlap for r in runners)
        sends = [r.pre(part=1) if two else r.pre() for r in runners]
        for r, s in zip(runners, sends):
            group.post(r.rank, s, r.counts)
        if two:
            for r in runners:
                r.interior()
            for r in runners:
                r.boundary(group.collect(r.rank, r.W))
        else:
            for r in runners:
                r.post(group.collect(r.rank, r.W))
        q, f2v, v2f, P = bp.q_dev.cpu().numpy(), bp.f2v.cpu().numpy(), bp.v2f.cpu().numpy(), bp.particles.cpu().numpy()
        for r in runners:
            plan = r.plan
            hid = plan.flat.var_hidden
            live = np.arange(n)[None, :] < r.bp.np_host[:, None]
            np.testing.assert_allclose(np.where(live, r.bp.particles.cpu().numpy(), 0)[hid], np.where(live, P[plan.var_gid], 0)[hid], rtol=1e-10, atol=1e-11, err_msg='particles')
            np.testing.assert_allclose(r.bp.q_dev.cpu().numpy()[hid], q[plan.var_gid][hid], rtol=1e-10, atol=1e-12, err_msg='q')
            he = hid[plan.flat.edge_var]
            le = live[plan.flat.edge_var] & he[:, None]
            np.testing.assert_allclose(np.where(le, r.bp.v2f.cpu().numpy(), 0), np.where(le, v2f[plan.edge_ids], 0), rtol=1e-8, atol=1e-8, err_msg='v2f')
            np.testing.assert_allclose(r.bp.f2v.cpu().numpy()[he], f2v[plan.edge_ids][he], rtol=1e-8, atol=1e-8, err_msg='f2v')


first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
ok, t0 = 0, time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    V, deg = 2 * int(rng.integers(150, 2000)), int(rng.choice([2, 3, 4, 6, 8]))          # (V * deg even: the generator pairs edge stubs)
    fd, ev, T = float(rng.choice([0.0, 0.2, 0.5])), float(rng.choice([0.0, 0.1, 0.3])), int(rng.choice([8, 32, 48]))
    n, world = int(rng.choice([8, 16, 33, 64])), int(rng.choice([2, 3, 5, 8]))
    approx, sweeps = str(rng.choice(['simple', 'EP'])), int(rng.integers(2, 5))
    mode = ('ownercompute', 'pairs', 'pairs two-part')[seed % 3]
    try:
        flat = synth.hybrid_mrf_flat(V=V, deg=deg, seed=seed, frac_discrete=fd, evidence_ratio=ev, T=T)
        if mode == 'ownercompute':
            owner_compute(flat, n, world, approx, sweeps)
        else:
            pairs(flat, n, world, approx, sweeps, mode.endswith('two-part'))
        ok += 1
    except Exception as e:
        print('FAIL seed %d (%s, V %d deg %d discrete %.1f evidence %.1f T %d n %d world %d %s sweeps %d): %s' % (
            seed, mode, V, deg, fd, ev, T, n, world, approx, sweeps, str(e)[:300].replace('\n', ' ')), flush=True)
print('%d of %d seeds pass (%.0f s)' % (ok, count, time.time() - t0))
sys.exit(0 if ok == count else 1)
