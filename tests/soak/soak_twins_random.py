"""soak of twin code paths that must not change a result, on random graphs: (a) the listed device draw (hidden continuous variables only)
against the full one, (b) the heavy kernel's ticketed work distribution against static striding, (c) the recorded (hipGraph) Gaussian
run against direct launches, (d) the fused ADAM loop against one call per step -- all bit for bit.
usage: python tests/soak/soak_twins_random.py [first seed] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'lifted-hybrid-variational-inference_amd')]
import numpy as np, torch
from lhvi import _abi, dist, synth
from lhvi.gabp import GaBP
from lhvi.pbp import EPBP
from lhvi.vi import VarInference


def same(x, y, what):
    x, y = x.cpu().numpy() if torch.is_tensor(x) else np.asarray(x), y.cpu().numpy() if torch.is_tensor(y) else np.asarray(y)
    assert x.shape == y.shape and x.tobytes() == y.tobytes(), what


def particle_runs(flat, n, approx, sweeps, **attrs):
    out = []
    for variant in (False, True):
        bp = EPBP(None, n=n, proposal_approximation=approx, sampler='device', seed=11)
        for k, (a, b) in attrs.items():
            setattr(bp, k, b if variant else a)
        bp._setup(None, flat=flat)
        r = dist.SingleRunner(bp)
        r.init()
        for _ in range(sweeps):
            r.sweep()
        out.append(bp)
    return out


first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
ok, t0 = 0, time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    V, deg = 2 * int(rng.integers(150, 3000)), int(rng.choice([2, 3, 4, 6]))
    fd, ev, T = float(rng.choice([0.0, 0.2, 0.5])), float(rng.choice([0.0, 0.1, 0.3])), int(rng.choice([8, 32, 48]))
    n, approx, sweeps = int(rng.choice([8, 16, 33, 64])), str(rng.choice(['simple', 'EP'])), int(rng.integers(2, 5))
    what = seed % 4
    try:
        if what == 0:
            flat = synth.hybrid_mrf_flat(V=V, deg=deg, seed=seed, frac_discrete=fd, evidence_ratio=ev, T=T)
            a, b = particle_runs(flat, n, approx, sweeps, listed_resample=(False, True))
            for name in ('particles', 'q_dev', 'eta', 'v2f', 'f2v', 'uniq'):
                same(getattr(a, name), getattr(b, name), 'listed draw: ' + name)
        elif what == 1:
            flat = synth.hybrid_mrf_flat(V=V, deg=deg, seed=seed, frac_discrete=fd, evidence_ratio=ev, T=T)
            a, b = particle_runs(flat, n, approx, sweeps, dynamic_f2v=(False, True))
            for name in ('particles', 'q_dev', 'eta', 'v2f', 'f2v'):
                same(getattr(a, name), getattr(b, name), 'work distribution: ' + name)
        elif what == 2:
            flat = synth.rgm_flat(C=int(rng.integers(5, 80)), B=int(rng.integers(3, 40)), n_values=int(rng.integers(0, 4)), evidence_ratio=float(rng.choice([0.05, 0.2, 0.4])), seed=seed)[0]
            its = int(rng.integers(1, 20))
            a, b = GaBP(flat), GaBP(flat)
            b.graph_replay_slots = 0
            a.run(its); a.run(its); b.run(its)
            same(a._f2v, b._f2v, 'recorded run: f2v'); same(a._v2f, b._v2f, 'recorded run: v2f'); same(a._mu_var, b._mu_var, 'recorded run: marginals')
        else:
            flat = synth.paper_popularity_flat(int(rng.integers(3, 30)), int(rng.integers(2, 6)), seed=seed, points=int(rng.choice([8, 20])))[0]
            K, Tq = int(rng.choice([1, 2, 3])), int(rng.choice([2, 3]))
            outs = []
            for fused in (True, False):
                vi = VarInference(None, K, Tq)
                vi.fused_loop = fused
                vi._setup_flat(flat)
                np.random.seed(seed)
                vi.init_param()
                vi.is_log, vi.log_fe = True, True
                vi.alpha, vi.b1, vi.b2, vi.eps, vi.t = 0.2, 0.9, 0.999, 1e-8, 0
                vi.time_log, vi.total_time = [], 0
                vi.ADAM_update(7)
                outs.append(([fe for _, fe in vi.time_log], vi._dev['eta_c'].cpu().numpy(), vi._dev['tau_d'].cpu().numpy(), vi._dev['w_tau'].cpu().numpy()))
            for x, y, name in zip(outs[0], outs[1], ('free energies', 'eta_c', 'tau_d', 'w_tau')):
                same(np.asarray(x), np.asarray(y), 'fused ADAM loop: ' + name)
        ok += 1
    except Exception as e:
        print('FAIL seed %d (case %d, V %d deg %d n %d %s T %d): %s' % (seed, what, V, deg, n, approx, T, str(e)[:300].replace('\n', ' ')), flush=True)
print('%d of %d seeds pass (%.0f s)' % (ok, count, time.time() - t0))
sys.exit(0 if ok == count else 1)
