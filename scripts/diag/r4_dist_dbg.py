import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'lifted-hybrid-variational-inference_amd')]
import numpy as np, torch
from lhvi import _abi, dist, synth
from lhvi.pbp import EPBP
seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
V, deg = 2 * int(rng.integers(150, 2000)), int(rng.choice([2, 3, 4, 6, 8]))
fd, ev, T = float(rng.choice([0.0, 0.2, 0.5])), float(rng.choice([0.0, 0.1, 0.3])), int(rng.choice([8, 32, 48]))
n, world = int(rng.choice([8, 16, 33, 64])), int(rng.choice([2, 3, 5, 8]))
approx, sweeps = str(rng.choice(['simple', 'EP'])), int(rng.integers(2, 5))
flat = synth.hybrid_mrf_flat(V=V, deg=deg, seed=seed, frac_discrete=fd, evidence_ratio=ev, T=T)
print('V', V, 'deg', deg, 'fd', fd, 'ev', ev, 'T', T, 'n', n, 'world', world, approx)
bp = EPBP(None, n=n, proposal_approximation=approx, sampler='device', seed=3)
bp._setup(None, flat=flat)
single = dist.SingleRunner(bp); single.init()
owner = dist.partition_variables(flat, world)
group = dist.LoopbackGroup(world)
runners = [dist.OwnerRunner(flat, n=n, seed=3, rank=r, world=world, proposal_approximation=approx, group=group, var_owner=owner) for r in range(world)]
for r in runners: r.init()
def lists_of(b):
    out = {}
    for name in ('heavy_desc', 'small16_desc', 'small32_desc', 'light_desc', 'fast_desc'):
        t = getattr(b, name, None)
        if t is not None and t.numel():
            out[name] = set(t.view(torch.int32).view(-1, 32)[:, 0].cpu().numpy().tolist())
    pd = getattr(b, 'pair_desc', None)
    out['generic'] = set(b.generic_edges.cpu().numpy().tolist())
    out['cq'] = set(b.cq_edges.cpu().numpy().tolist())
    return out
L0 = lists_of(bp)
print('single lists', {k: len(v) for k, v in L0.items()}, 'n_pair', getattr(bp, 'n_pair', 0))
for it in range(sweeps):
    single.sweep()
    sends = [r.owned_half() for r in runners]
    for r, s in zip(runners, sends): group.post(r.rank, s, r.counts)
    for r in runners: r.interior()
    for r in runners: r.boundary(group.collect(r.rank, None))
    F = bp.f2v.cpu().numpy()
    for r in runners:
        plan = r.plan
        mine = ~plan.edge_skip
        f = r.bp.f2v.cpu().numpy()
        g = F[plan.edge_ids]
        bad = np.flatnonzero(mine & (np.abs(np.nan_to_num(f) - np.nan_to_num(g)).max(axis=1) > 0))
        if bad.size:
            Lr = lists_of(r.bp)
            ge = plan.edge_ids[bad]
            where_single = [next((k for k, v in L0.items() if int(e) in v), '?') for e in ge[:2000]]
            where_rank = [next((k for k, v in Lr.items() if int(e) in v), '?') for e in bad[:2000]]
            import collections
            print('sweep', it, 'rank', r.rank, 'bad rows', bad.size, 'of', int(mine.sum()), 'single lists', collections.Counter(where_single), 'rank lists', collections.Counter(where_rank),
                  'max diff', float(np.abs(np.nan_to_num(f[bad]) - np.nan_to_num(g[bad])).max()))
            e = bad[0]
            print('   e.g. local edge', int(e), 'cols differing', np.flatnonzero(f[e] != g[e])[:10], 'np target', int(r.bp.np_host[plan.flat.edge_var[e]]))
            sys.exit(0)
print('no difference')
