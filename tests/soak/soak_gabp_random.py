"""soak of the Gaussian sweep on random relational instances: GaBP(g).run through the device (pull form, recorded launches) against
the C oracle's kernel-pair loops (1e-12), and GaLBP(g).run (colour passing + counted sweep) against GaBP on the ground graph
(MAP of every hidden variable, 1e-9).  usage: python tests/soak/soak_gabp_random.py [first seed] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'lifted-hybrid-variational-inference_amd')]
import numpy as np
from lhvi import generators, lifting
from lhvi.flat import flatten
from lhvi.gabp import GaBP, GaLBP
from oracle import oracle

first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
ok, t0 = 0, time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    C, B = int(rng.integers(3, 40)), int(rng.integers(2, 12))
    if os.environ.get('SOAK_HUBS'):              # rows of 33-512 entries (chunk sums) and of more than 512 (hub kernel) in the same graph
        C, B = int(rng.integers(40, 1500)), int(rng.integers(1, 3))
    rel = generators.rgm(C, B)
    rel.ground_graph()
    keys = [('market', 'c%d' % c) for c in range(C)] + [('loss', 'c%d' % c, 'b%d' % b) for c in range(C) for b in range(B)] + \
           [('revenue', 'b%d' % b) for b in range(B)] + [('recession', 'all')]
    pool = np.round(rng.uniform(-30, 30, int(rng.integers(1, 6))), 2)
    data = {}
    for k in keys:
        if rng.random() < rng.choice([0.02, 0.1, 0.3]):
            data[k] = float(rng.choice(pool)) if rng.random() < 0.6 else float(np.round(rng.uniform(-30, 30), 3))
    g, table = rel.add_evidence(data)
    its = int(rng.integers(1, 25))
    try:
        flat = flatten(g)
        a = GaBP(g)
        a.run(its)
        f2v, v2f, mv = oracle.gabp_run(flat, its)
        hid = np.flatnonzero(flat.var_hidden)
        got = np.array([a.get_belief_params(flat.rvs[v]) for v in hid])
        np.testing.assert_allclose(got, mv[hid], rtol=1e-12, atol=1e-12, err_msg='GaBP vs oracle')
        b = GaLBP(g)
        b.run(its)
        mb = np.array([b.map(flat.rvs[v]) for v in hid])
        np.testing.assert_allclose(mb, mv[hid, 0], rtol=1e-9, atol=1e-9, err_msg='GaLBP vs ground')
        # the array path of the same lifting: colours, refinement and the lifted graph without objects, GaBP on the lifted FlatGraph
        rv0, f0, sym = lifting.initial_colors_flat(flat, True)
        rvc, fc = lifting.refine_flat(flat, sym, rv0, f0)
        orv, of = b.g.colors()
        assert (oracle.canonical_labels(rvc) == oracle.canonical_labels(orv)) and (oracle.canonical_labels(fc) == oracle.canonical_labels(of)), 'partitions of the two liftings'
        c = GaBP(lifting.lift_flat(flat, rvc, fc))
        c.run(its)
        mc = c._mu_var[rvc[hid], 0]
        np.testing.assert_allclose(mc, mv[hid, 0], rtol=1e-9, atol=1e-9, err_msg='lifted on arrays vs ground')
        ok += 1
    except Exception as e:
        print('FAIL seed %d (C %d B %d evidence %d its %d): %s' % (seed, C, B, len(data), its, str(e)[:300].replace('\n', ' ')), flush=True)
print('%d of %d seeds pass (%.0f s)' % (ok, count, time.time() - t0))
sys.exit(0 if ok == count else 1)
