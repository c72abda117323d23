"""soak of the particle coarse-to-fine run on random instances: HybridLBP.run(it, c2f) through arrays (lhvi.c2f.run_c2f_flat), through
Python objects per cluster (c2f_on_objects) and without any object (on_flat(...).run_flat) -- same partitions at every draw, same
state arrays bit for bit.  Random RGM instances (template sizes, evidence patterns with tied and distinct values, c2f thresholds,
k-means settings), and -- third argument `hmln` / `mixed` -- instances of the paper-popularity hybrid MLN.
usage: python tests/soak/soak_c2f_random.py [first seed] [count] [rgm | hmln | mixed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'lifted-hybrid-variational-inference_amd')]
import numpy as np, torch
from lhvi import generators
from lhvi.flat import flatten
from lhvi.pbp import HybridLBP

def same(x, y, what):
    """bit-identical (NaN payloads included)"""
    x, y = x.cpu().numpy(), y.cpu().numpy()
    if x.shape != y.shape:
        raise AssertionError('%s: shapes %s / %s' % (what, x.shape, y.shape))
    if x.tobytes() != y.tobytes():
        d = np.abs(np.nan_to_num(x.astype(np.float64), nan=0.0, posinf=1e300, neginf=-1e300) - np.nan_to_num(y.astype(np.float64), nan=0.0, posinf=1e300, neginf=-1e300))
        bad = np.argwhere(d > 0)
        raise AssertionError('%s: %d entries differ (max %g) of %d, nan pattern equal: %s, first %s: %r vs %r' % (
            what, len(bad), d.max(), x.size, bool((np.isnan(x) == np.isnan(y)).all()), bad[:1].tolist(), x[tuple(bad[0])] if len(bad) else None, y[tuple(bad[0])] if len(bad) else None))


first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
MODEL = sys.argv[3] if len(sys.argv) > 3 else 'rgm'          # rgm | hmln | mixed
ok, t0 = 0, time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    hmln = MODEL == 'hmln' or (MODEL == 'mixed' and seed % 3 == 2)
    if hmln:
        # the paper-popularity hybrid MLN (binary SameSession / PaperIn atoms, continuous popularities, ternary formulas)
        C, B = int(rng.integers(3, 9)), int(rng.integers(2, 4))
        rel = generators.paper_popularity(C, B, points=int(rng.choice([8, 12])))
        rel.ground_graph()
        data = {}
        for k in rel.rvs_dict:
            if rng.random() < rng.choice([0.1, 0.3, 0.6]):
                data[k] = int(rng.integers(0, 2)) if k[0] in ('SameSession', 'PaperIn') else float(np.round(rng.choice([rng.uniform(0, 10), 2.5, 7.0]), 2))
    else:
        C, B = int(rng.integers(4, 16)), int(rng.integers(2, 6))
        if os.environ.get('SOAK_HUBS'):          # the shared atoms touch 70 ... 400 factors: hub rows in the lifted graphs of the early sweeps
            C, B = int(rng.integers(70, 400)), int(rng.integers(1, 3))
        rel = generators.rgm(C, B)
        rel.ground_graph()
        keys = [('market', 'c%d' % c) for c in range(C)] + [('loss', 'c%d' % c, 'b%d' % b) for c in range(C) for b in range(B)] + \
               [('revenue', 'b%d' % b) for b in range(B)] + [('recession', 'all')]
        pool = np.round(rng.uniform(-30, 30, int(rng.integers(1, 6))), 2)          # a few tied values ...
        data = {}
        for k in keys:
            if rng.random() < rng.choice([0.05, 0.15, 0.4]):
                data[k] = float(rng.choice(pool)) if rng.random() < 0.6 else float(np.round(rng.uniform(-30, 30), 3))     # ... and distinct ones
    g, table = rel.add_evidence(data)
    rvs = list(g.rvs)
    n, its = int(rng.choice([5, 10, 16])), int(rng.integers(3, 7))
    c2f = float(rng.choice([0.0, 0.5, 5.0]))
    kk, kit = int(rng.choice([2, 3])), int(rng.choice([3, 10]))
    gflat = flatten(g, require_device_potentials=True)
    samples = np.clip(rng.normal(5, 4, (its + 1, gflat.V, n)), -15, 15) if hmln else np.clip(rng.normal(0, 8, (its + 1, gflat.V, n)), -50, 50)
    inject = lambda k, flat, q: samples[k][flat.rep_ground]
    if seed % 4 == 3:
        c2f = -1.0                 # every fourth instance: colour passing to the stable partition, then the sweeps
    try:
        if c2f == -1.0:
            a = HybridLBP(g, n=n, k_mean_k=kk, k_mean_iteration=kit, proposal_approximation='simple', sampler='device', seed=7)
            a.run(its, c2f=-1)
            c = HybridLBP.on_flat(gflat, n=n, k_mean_k=kk, k_mean_iteration=kit, proposal_approximation='simple', sampler='device', seed=7)
            c.run_flat(its, c2f=-1)
            for name in ('f2v', 'v2f', 'eta', 'q_dev', 'particles', 'old_particles', 'uniq'):
                same(getattr(a, name), getattr(c, name), 'stable partition, objects vs on_flat: ' + name)
            assert torch.isfinite(a.v2f).all()
            ok += 1
            continue
        runs = []
        for on_objects in (True, False):
            bp = HybridLBP(g, n=n, k_mean_k=kk, k_mean_iteration=kit, proposal_approximation='simple', sampler=inject)
            bp.c2f_on_objects = on_objects
            bp.run(its, c2f=c2f)
            runs.append(bp)
        a, b = runs
        assert len(a.c2f_history) == len(b.c2f_history)
        for (ra, fa), (rb, fb) in zip(a.c2f_history, b.c2f_history):
            assert (ra == rb).all() and (fa == fb).all(), 'partitions differ'
        for name in ('f2v', 'v2f', 'eta', 'q_dev', 'particles', 'old_particles', 'uniq'):
            same(getattr(a, name), getattr(b, name), name)
        c = HybridLBP.on_flat(gflat, n=n, k_mean_k=kk, k_mean_iteration=kit, proposal_approximation='simple', sampler=inject)
        c.run_flat(its, c2f=c2f)
        for name in ('f2v', 'v2f', 'eta', 'q_dev', 'particles'):
            same(getattr(b, name), getattr(c, name), 'on_flat: ' + name)
        assert torch.isfinite(b.v2f).all()
        ok += 1
    except Exception as e:
        print('FAIL seed %d (C %d B %d evidence %d n %d its %d c2f %s k %d/%d): %r' % (seed, C, B, len(data), n, its, c2f, kk, kit, e), flush=True)
print('%d of %d seeds pass (%.0f s)' % (ok, count, time.time() - t0))
sys.exit(0 if ok == count else 1)
