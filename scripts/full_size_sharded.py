"""the owner-computes split at the headline size: `world` simulated ranks on one GPU (loopback exchange) over the 10 M-edge hybrid MRF of
bench.py against the single-GPU sweep -- proposals, particles and both message arrays bit for bit after every sweep.
usage: python scripts/full_size_sharded.py [edges] [world] [sweeps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'lifted-hybrid-variational-inference_amd')]
import numpy as np, torch
from lhvi import dist, synth
from lhvi.pbp import EPBP

E = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 4
sweeps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
n = 64
t0 = time.perf_counter()
flat = synth.hybrid_mrf_flat(V=E // 4, deg=4, seed=0)
bp = EPBP(None, n=n, proposal_approximation='simple', sampler='device', seed=3)
bp._setup(None, flat=flat)
single = dist.SingleRunner(bp)
single.init()
owner = dist.partition_variables(flat, world)
group = dist.LoopbackGroup(world)
runners = [dist.OwnerRunner(flat, n=n, seed=3, rank=r, world=world, group=group, var_owner=owner) for r in range(world)]
for r in runners:
    r.init()
print('graph, plans, states: %.0f s' % (time.perf_counter() - t0), flush=True)
dev = bp.q_dev.device
checked = 0
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
        loc = torch.from_numpy((np.arange(plan.flat.V) < plan.n_owned + plan.n_ghost) & plan.flat.var_hidden).to(dev)
        own = torch.from_numpy((np.arange(plan.flat.V) < plan.n_owned) & plan.flat.var_hidden).to(dev)
        live = torch.from_numpy(np.arange(n)[None, :] < r.bp.np_host[:, None]).to(dev)
        assert torch.equal(r.bp.q_dev[own], bp.q_dev[gid][own]), 'q (sweep %d rank %d)' % (it, r.rank)
        assert torch.equal(torch.where(live, r.bp.particles, 0.0)[loc], torch.where(live, bp.particles[gid], 0.0)[loc]), 'particles'
        eid = torch.from_numpy(plan.edge_ids).to(dev)
        mine = torch.from_numpy(~plan.edge_skip).to(dev)
        le = live[torch.from_numpy(plan.flat.edge_var.astype(np.int64)).to(dev)]
        assert torch.equal(torch.where(le, r.message_rows(), 0.0)[mine], torch.where(le, bp.v2f[eid], 0.0)[mine]), 'v2f'
        assert torch.equal(r.bp.f2v[mine], bp.f2v[eid][mine]), 'f2v'
        checked += int(mine.sum().item())
    print('sweep %d: every rank equals the single-GPU run bit for bit' % it, flush=True)
print(json.dumps(dict(config='owner-computes split at the headline size against one GPU', edges=int(flat.E), world=world, sweeps=sweeps, particles=n,
                      f2v_rows_compared=checked, bit_identical=True, cut_rows_sent=[int(r.lay['send']['row_edge'].size + r.lay['send']['cont_edge'].size) for r in runners])))
