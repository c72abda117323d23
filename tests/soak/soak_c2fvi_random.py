"""soak of the coarse-to-fine variational run on random instances: run_c2fvi_flat (ground arrays in: refinement and re-lifting on the
device, parameters per cluster) against run_c2fvi (Python objects per cluster), same start -- every round's partition and Gaussian
observations identical, free energies and final parameters to rounding.  Random RGM instances and evidence patterns.
usage: python tests/soak/soak_c2fvi_random.py [first seed] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'lifted-hybrid-variational-inference_amd')]
import numpy as np, torch
from lhvi import c2fvi, generators
from lhvi.c2f import DeviceRefiner
from lhvi.flat import flatten

first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
ok, t0 = 0, time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    C, B = int(rng.integers(4, 14)), int(rng.integers(2, 6))
    rel = generators.rgm(C, B)
    rel.ground_graph()
    keys = [('market', 'c%d' % c) for c in range(C)] + [('loss', 'c%d' % c, 'b%d' % b) for c in range(C) for b in range(B)] + \
           [('revenue', 'b%d' % b) for b in range(B)] + [('recession', 'all')]
    pool = np.round(rng.uniform(-30, 30, int(rng.integers(1, 6))), 2)
    data = {}
    for k in keys:
        if rng.random() < rng.choice([0.05, 0.15, 0.4]):
            data[k] = float(rng.choice(pool)) if rng.random() < 0.6 else float(np.round(rng.uniform(-30, 30), 3))
    g, table = rel.add_evidence(data)
    K, T = int(rng.choice([1, 2])), int(rng.choice([2, 3]))
    its, every = int(rng.choice([20, 30])), int(rng.choice([5, 10]))
    gflat = flatten(g, require_device_potentials=True)
    eta_c0 = np.ones((gflat.V, K, 2)); eta_c0[:, :, 0] = rng.uniform(-1.5, 1.5, (1, K))      # (one start per component: a coarse cluster starts from ITS members' common value)
    tau_d0 = np.zeros((gflat.V, K, 1))
    try:
        out = []
        for arrays in (False, True):
            vi = c2fvi.VarInference(g, K, T)
            opts = dict(vi._options(), update_obs_its=every, k_mean_k=int(2 + seed % 2))
            rounds = []
            obs = lambda r, st: rounds.append((np.array(st['rvc']).copy(), np.array(st['fc']).copy(), np.array(st['obs_var']).copy(), st['flat'].var_value.copy()))
            if arrays:
                res = c2fvi.run_c2fvi_flat(gflat, c2fvi._DeviceEngine(vi), K, its, 0.2, opts, init=(eta_c0, tau_d0), observer=obs)
                eta = res['params']['eta_c'][res['rvc']]
            else:
                res = c2fvi.run_c2fvi(g, c2fvi._DeviceEngine(vi), DeviceRefiner(g), K, its, 0.2, opts, init=(eta_c0, tau_d0), observer=obs)
                eta = res['params']['eta_c']
            out.append((rounds, np.array(res['fe_log']), eta, np.array(res['params']['w_tau'])))
        (ra, fa, ea, wa), (rb, fb, eb, wb) = out
        assert len(ra) == len(rb)
        for r, (x, y) in enumerate(zip(ra, rb)):
            assert (x[0] == y[0]).all() and (x[1] == y[1]).all(), 'partition of round %d' % r
            assert x[2].tobytes() == y[2].tobytes(), 'Gaussian observation variances of round %d' % r
            assert x[3].tobytes() == y[3].tobytes(), 'cluster evidence values of round %d' % r
        np.testing.assert_allclose(fa, fb, rtol=1e-10, err_msg='free energies')
        hid = gflat.var_hidden & gflat.var_cont
        np.testing.assert_allclose(ea[hid], eb[hid], rtol=1e-8, atol=1e-10, err_msg='final eta')
        np.testing.assert_allclose(wa, wb, rtol=1e-8, atol=1e-10, err_msg='final w_tau')
        ok += 1
    except Exception as e:
        print('FAIL seed %d (C %d B %d evidence %d K %d T %d its %d/%d): %s' % (seed, C, B, len(data), K, T, its, every, str(e)[:300].replace('\n', ' ')), flush=True)
print('%d of %d seeds pass (%.0f s)' % (ok, count, time.time() - t0))
sys.exit(0 if ok == count else 1)
