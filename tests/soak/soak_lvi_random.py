"""soak of the lifted variational step on random relational instances: LiftedVarInference(g) (colour passing on the device, lifted
objects, flatten) against the array path (initial_colors_flat -> refine_flat -> lift_flat -> VarInference on the lifted FlatGraph)
and against the C oracle on the same lifted graph: gradient and free energy after the reference's init_param draw, then after five
ADAM updates.  Instances: the RGM (Gaussian pairs) and the paper-popularity hybrid MLN (binary atoms, ternary formulas) with random
evidence.  usage: python tests/soak/soak_lvi_random.py [first seed] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'lifted-hybrid-variational-inference_amd')]
import numpy as np, torch
from lhvi import _abi, generators, lifting
from lhvi.flat import flatten
from lhvi.vi import LiftedVarInference, VarInference
from oracle import oracle

first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
ok, t0 = 0, time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    hmln = seed % 2 == 1
    if hmln:
        P, Tn = int(rng.integers(3, 9)), int(rng.integers(2, 4))
        rel = generators.paper_popularity(P, Tn, points=8)
        rel.ground_graph()
        data = {}
        for k in rel.rvs_dict:
            if rng.random() < rng.choice([0.1, 0.3, 0.6]):
                data[k] = int(rng.integers(0, 2)) if k[0] in ('SameSession', 'PaperIn') else float(np.round(rng.choice([rng.uniform(0, 10), 2.5, 7.0]), 2))
    else:
        C, B = int(rng.integers(4, 16)), int(rng.integers(2, 6))
        rel = generators.rgm(C, B)
        rel.ground_graph()
        pool = np.round(rng.uniform(-30, 30, int(rng.integers(1, 4))), 2)
        data = {k: float(rng.choice(pool)) for k in rel.rvs_dict if rng.random() < rng.choice([0.05, 0.2, 0.4])}
    g, table = rel.add_evidence(data)
    K, T = int(rng.choice([1, 2])), int(rng.choice([2, 3]))
    try:
        a = LiftedVarInference(g, K, T)
        np.random.seed(seed)
        a.init_param()
        a._grad()
        gflat = flatten(g, require_device_potentials=True)
        rv0, f0, sym = lifting.initial_colors_flat(gflat, True)
        dg = _abi.DeviceGraph(gflat)
        rvc, fc = lifting.refine_flat(gflat, sym, rv0, f0, dg=dg, device_out=True)
        lflat = lifting.lift_flat(gflat, rvc, fc, dg=dg)
        b = VarInference(None, K, T)
        b._setup_flat(lflat)
        np.random.seed(seed)
        b.init_param()
        b._grad()
        assert a.flat.V == lflat.V and a.flat.F == lflat.F and a.flat.E == lflat.E, 'lifted sizes %s / %s' % ((a.flat.V, a.flat.F, a.flat.E), (lflat.V, lflat.F, lflat.E))
        for name in ('g_w', 'g_c', 'g_d', 'fe'):
            x, y = a._dev[name].cpu().numpy(), b._dev[name].cpu().numpy()
            np.testing.assert_allclose(x, y, rtol=1e-12, atol=1e-12, err_msg='objects vs arrays: ' + name)
        o = oracle.ViOracle(lflat, K, T, quirks=1)
        o.set_params(b._dev['w_tau'].cpu().numpy(), b._dev['eta_c'].cpu().numpy(), b._dev['tau_d'].cpu().numpy())
        want = o.grad()
        cont, disc = lflat.var_hidden & lflat.var_cont, lflat.var_hidden & ~lflat.var_cont
        if np.isfinite(want[3]):
            np.testing.assert_allclose(b._dev['fe'].cpu().numpy()[0], want[3], rtol=1e-9, err_msg='free energy vs oracle')
            np.testing.assert_allclose(b._dev['g_w'].cpu().numpy(), want[0], rtol=1e-8, atol=1e-8, err_msg='g_w vs oracle')
            np.testing.assert_allclose(b._dev['g_c'].cpu().numpy()[cont], want[1][cont], rtol=1e-8, atol=1e-8, err_msg='g_c vs oracle')
            np.testing.assert_allclose(b._dev['g_d'].cpu().numpy()[disc], want[2][disc], rtol=1e-8, atol=1e-8, err_msg='g_d vs oracle')
        for vi in (a, b):
            vi.is_log, vi.log_fe = True, True
            vi.alpha, vi.b1, vi.b2, vi.eps, vi.t = 0.2, 0.9, 0.999, 1e-8, 0
            vi.time_log, vi.total_time = [], 0
            vi.ADAM_update(5)
        np.testing.assert_allclose([fe for _, fe in a.time_log], [fe for _, fe in b.time_log], rtol=1e-11, err_msg='free energies after the updates')
        if os.environ.get('SOAK_VERBOSE'):
            print('seed', seed, 'hmln' if hmln else 'rgm', 'ground', (gflat.V, gflat.F, gflat.E), 'lifted', (lflat.V, lflat.F, lflat.E), 'K', K, 'T', T, 'fe', [round(fe, 6) for _, fe in b.time_log][:3], flush=True)
        ok += 1
    except Exception as e:
        print('FAIL seed %d (%s, evidence %d, K %d T %d): %s' % (seed, 'hmln' if hmln else 'rgm', len(data), K, T, str(e)[:300].replace('\n', ' ')), flush=True)
print('%d of %d seeds pass (%.0f s)' % (ok, count, time.time() - t0))
sys.exit(0 if ok == count else 1)
