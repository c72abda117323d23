"""soak of the particle sweep on random relational instances: EPBP(g).run on the ground graph and HybridLBP(g).run lifted (c2f = -1),
'simple' and 'EP' proposals, 5 ... 64 particles, against the C oracle replaying the same samples on the same (lifted) graph:
proposals, sites, v -> f and f -> v tables after the run ('EP' runs whose sites become vacuous: update by update up to that point).  Instances: the RGM (Gaussian pairs, 100 integral points) and the
paper-popularity hybrid MLN (binary atoms, ternary formulas).  usage: python tests/soak/soak_pbp_random.py [first seed] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'lifted-hybrid-variational-inference_amd')]
import numpy as np, torch
from lhvi import generators
from lhvi.pbp import EPBP, HybridLBP
from oracle import oracle

RTOL, ATOL = 1e-9, 1e-8
first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
ok, forked, compared_updates, t0 = 0, 0, 0, time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    hmln = seed % 2 == 1
    if hmln:
        P, Tn = int(rng.integers(3, 10)), int(rng.integers(2, 4))
        if os.environ.get('SOAK_HUBS'):
            P, Tn = int(rng.integers(70, 200)), int(rng.integers(1, 3))          # a topic's popularity touches every paper
        rel = generators.paper_popularity(P, Tn, points=int(rng.choice([8, 20, 32])))
        rel.ground_graph()
        data = {}
        for k in rel.rvs_dict:
            if rng.random() < rng.choice([0.1, 0.3, 0.6]):
                data[k] = int(rng.integers(0, 2)) if k[0] in ('SameSession', 'PaperIn') else float(np.round(rng.choice([rng.uniform(0, 10), 2.5, 7.0]), 2))
    else:
        C, B = int(rng.integers(4, 16)), int(rng.integers(2, 6))
        if os.environ.get('SOAK_HUBS'):          # template variables with 70 ... 700 incident factors: the hub rows of the v -> f and proposal kernels
            C, B = int(rng.integers(70, 700)), int(rng.integers(1, 3))
        rel = generators.rgm(C, B)
        rel.ground_graph()
        pool = np.round(rng.uniform(-30, 30, int(rng.integers(1, 4))), 2)
        data = {k: float(rng.choice(pool)) for k in rel.rvs_dict if rng.random() < rng.choice([0.05, 0.2, 0.4])}
    g, table = rel.add_evidence(data)
    n = int(rng.choice([5, 10, 16, 20, 33, 50, 64]))
    its = int(rng.integers(2, 5))
    approx = str(rng.choice(['simple', 'EP']))
    lifted = bool(rng.integers(0, 2))
    samples = []
    srng = np.random.default_rng(seed + 12345)

    def sampler(k, flat, q):
        cont = flat.var_hidden & flat.var_cont
        lo, hi = flat.dom_lo[flat.var_dom], flat.dom_hi[flat.var_dom]
        out = np.zeros((flat.V, n))
        out[cont] = np.clip(srng.standard_normal((int(cont.sum()), n)) * np.sqrt(q[cont, 1:2]) + q[cont, 0:1], lo[cont, None], hi[cont, None])
        samples.append(out)
        return out
    try:
        bp = (HybridLBP if lifted else EPBP)(g, n=n, proposal_approximation=approx, sampler=sampler)
        bp.run(its)
        flat = bp.flat
        o = oracle.PbpOracle(flat, n, ep=approx == 'EP', epbp=not lifted, var_threshold=5 if lifted else 3)
        o.run(its, samples)
        hid_e = flat.var_hidden[flat.edge_var]
        cont = flat.var_hidden & flat.var_cont
        if approx == 'EP':
            # the EP rule divides Gaussians: new site = tilted / cavity with variance sig c / (c - sig).  While the messages are still
            # flat the tilted distribution IS the cavity on the grid, c - sig is rounding noise, and whether the site comes out as
            # (mean, +1e15) -- accepted, a vacuous site -- or as (mean, -1e15) -- rejected, the old site stays -- is decided by the
            # order of a sum; the reference's own outcome changes with it (EPBP:123-154, update_proposal EPBP:93-97).  Such an
            # update is recognised by a site variance no message could have produced.  The two runs are compared update by update
            # UP TO the first such one (everything before it to 1e-8: sites, proposals and both message tables); at that update
            # every site of either run must be a valid outcome of the reference's rule -- the previous site kept, bit for bit, or
            # an accepted one with 0 < var < inf at or above the clamp -- and the sites neither run made vacuous must still agree;
            # after it the runs have forked (a vacuous site against a kept one) and are not compared.
            ce_ = cont[flat.edge_var] & (flat.edge_canon == np.arange(flat.E))
            deg_tot = np.zeros(flat.V)
            np.add.at(deg_tot, flat.edge_var[flat.var_edge], flat.edge_count[flat.var_edge])
            min_sig = (deg_tot * (5 if lifted else 3))[flat.edge_var]

            def state(m):            # both runs after m sweeps (m - 1 proposal updates), replaying the main run's samples
                if m == its:
                    return bp, o
                replay = iter(list(samples[:m]))
                b_ = (HybridLBP if lifted else EPBP)(g, n=n, proposal_approximation=approx, sampler=lambda k, fl, q: next(replay))
                b_.run(m)
                o_ = oracle.PbpOracle(b_.flat, n, ep=True, epbp=not lifted, var_threshold=5 if lifted else 3)
                o_.run(m, list(samples[:m]))
                return b_, o_
            big = lambda st: (np.abs(st[0].eta.cpu().numpy()[ce_, 1]) > 1e9) | (np.abs(st[1].eta[ce_, 1]) > 1e9)
            states = {m: state(m) for m in range(1, its + 1)}
            fork = next((m for m in range(1, its + 1) if big(states[m]).any()), None)
            if fork is not None:
                for m in range(1, fork):                       # the updates before the first vacuous / rejected one: as any other run
                    b_, o_ = states[m]
                    np.testing.assert_allclose(b_.q_dev.cpu().numpy()[cont], o_.q[cont], rtol=1e-8, atol=1e-10, err_msg='proposals before the fork')
                    np.testing.assert_allclose(b_.eta.cpu().numpy()[ce_], o_.eta[ce_], rtol=1e-8, atol=1e-10, err_msg='sites before the fork')
                    npe_ = o_.np[flat.edge_var]
                    live_ = hid_e[:, None] & (np.arange(n)[None, :] < npe_[:, None])
                    np.testing.assert_allclose(b_.v2f.cpu().numpy()[live_], o_.v2f[live_], rtol=RTOL, atol=ATOL, err_msg='v2f before the fork')
                    np.testing.assert_allclose(b_.f2v.cpu().numpy()[:, :n][live_], o_.f2v[:, :n][live_], rtol=RTOL, atol=ATOL, err_msg='f2v before the fork')
                assert fork >= 2, 'the initial sites cannot be vacuous'
                (b0, o0), (b1, o1) = states[fork - 1], states[fork]
                mask = big(states[fork])
                for name, prev, new in (('device', b0.eta.cpu().numpy()[ce_], b1.eta.cpu().numpy()[ce_]), ('oracle', o0.eta[ce_], o1.eta[ce_])):
                    kept = (new == prev).all(axis=1)
                    accepted = (new[:, 1] > 0) & np.isfinite(new[:, 1]) & (new[:, 1] >= min_sig[ce_] * (1 - 1e-12)) & np.isfinite(new[:, 0])
                    assert (kept | accepted).all(), '%s: a site after the forking update is neither the kept one nor an accepted one' % name
                np.testing.assert_allclose(b1.eta.cpu().numpy()[ce_][~mask], o1.eta[ce_][~mask], rtol=1e-8, atol=1e-10,
                                           err_msg='sites the forking update did not make vacuous')
                forked += 1
                compared_updates += fork - 1
                ok += 1
                continue
        np.testing.assert_allclose(bp.q_dev.cpu().numpy()[cont], o.q[cont], rtol=1e-8, atol=1e-10, err_msg='proposals')
        canon = flat.edge_canon == np.arange(flat.E)           # (a repeated cluster's later occurrence in a lifted factor aliases the first: no site of its own)
        np.testing.assert_allclose(bp.eta.cpu().numpy()[cont[flat.edge_var] & canon], o.eta[cont[flat.edge_var] & canon], rtol=1e-8, atol=1e-10, err_msg='sites')
        npe = o.np[flat.edge_var]
        live = hid_e[:, None] & (np.arange(n)[None, :] < npe[:, None])
        np.testing.assert_allclose(bp.v2f.cpu().numpy()[live], o.v2f[live], rtol=RTOL, atol=ATOL, err_msg='v2f')
        got, want = bp.f2v.cpu().numpy(), o.f2v
        np.testing.assert_allclose(got[:, :n][live], want[:, :n][live], rtol=RTOL, atol=ATOL, err_msg='f2v at the particles')
        ce = cont[flat.edge_var]
        np.testing.assert_allclose(got[ce, n:], want[ce, n:], rtol=RTOL, atol=ATOL, err_msg='f2v at the integral points')
        if os.environ.get('SOAK_VERBOSE'):
            print('seed', seed, 'hmln' if hmln else 'rgm', 'lifted' if lifted else 'ground', (flat.V, flat.F, flat.E), 'n', n, approx, 'its', its, flush=True)
        ok += 1
    except Exception as e:
        print('FAIL seed %d (%s %s, evidence %d, n %d, %s, its %d): %s' % (seed, 'hmln' if hmln else 'rgm', 'lifted' if lifted else 'ground', len(data), n, approx, its,
                                                                       str(e)[:300].replace('\n', ' ')), flush=True)
print('%d of %d seeds pass; %d of them are EP runs that fork at a division of equal Gaussians: compared up to the forking update (%d updates in all), '
      'both outcomes of that update valid, not compared after it (%.0f s)' % (ok, count, forked, compared_updates, time.time() - t0))
sys.exit(0 if ok == count else 1)
