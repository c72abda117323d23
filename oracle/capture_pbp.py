"""Golden vectors for the particle-BP path (EPBP / HybridLBP), captured from the reference.

TEST INFRASTRUCTURE, build container only (see capture_golden.py).  The reference's solvers are driven
unmodified; a wrapper around ``generate_sample`` (called once per sweep, right after ``update_proposal``)
snapshots messages / proposals at that point, which yields every per-iteration quantity without
restating the run loop.
"""
import copy
import json
import os

import numpy as np


def _snap_messages(solver, edges, sample, grid_of):
    """per edge (factor-major): v2f at the sample points, f2v at sample points then grid points"""
    v2f, f2v = [], []
    for f, rv in edges:
        if rv.value is not None:
            v2f.append(None)
            f2v.append(None)
            continue
        pts = list(sample[rv])
        mv = solver.message[(rv, f)]
        mf = solver.message[(f, rv)]
        v2f.append([float(mv[x]) for x in pts])
        row = [float(mf[x]) for x in pts]
        if rv.domain.continuous:
            row += [float(mf[x]) for x in grid_of(rv)]
        f2v.append(row)
    return v2f, f2v


def _pad(rows, width):
    out = np.full((len(rows), width), np.nan)
    for i, r in enumerate(rows):
        if r is not None:
            out[i, :len(r)] = r
    return out


def run_with_snapshots(solver, rvs, edges, iterations, run_kwargs=None):
    """returns list of snapshots; snapshot k is taken at the k-th generate_sample call (k=0: initial draw),
    plus a final one after run()"""
    snaps = []
    orig = solver.generate_sample
    grid_of = lambda rv: rv.domain.integral_points

    def wrapped():
        k = len(snaps)
        state = {}
        if k > 0:
            prev_sample = snaps[-1]['new_sample']
            v2f, f2v = _snap_messages(solver, edges, prev_sample, grid_of)
            state['v2f'], state['f2v'] = v2f, f2v
            state['q'] = {i: tuple(map(float, solver.q[rv])) for i, rv in enumerate(rvs) if rv in solver.q}
            state['eta'] = [tuple(map(float, solver.eta_message[(f, rv)])) if (f, rv) in solver.eta_message else None
                            for f, rv in edges]
        new = orig()
        state['new_sample'] = new
        snaps.append(state)
        return new

    solver.generate_sample = wrapped
    solver.run(iterations, **(run_kwargs or {}))
    solver.generate_sample = orig
    last_sample = solver.sample
    v2f, f2v = _snap_messages(solver, edges, last_sample, grid_of)
    snaps.append({'v2f': v2f, 'f2v': f2v,
                  'q': {i: tuple(map(float, solver.q[rv])) for i, rv in enumerate(rvs) if rv in solver.q},
                  'eta': [tuple(map(float, solver.eta_message[(f, rv)])) if (f, rv) in solver.eta_message else None
                          for f, rv in edges]})
    return snaps


def pack(snaps, rvs, edges, n):
    """dense arrays: sample [K, V, n] (NaN padded), v2f [K+1, E, n], f2v [K+1, E, n+T], q [K+1, V, 2], eta [K+1, E, 2]"""
    V, E = len(rvs), len(edges)
    T = max([len(rv.domain.integral_points) for rv in rvs if rv.domain.continuous] + [0])
    K = len(snaps) - 1       # number of generate_sample calls
    sample = np.full((K, V, n), np.nan)
    for k in range(K):
        for i, rv in enumerate(rvs):
            s = snaps[k]['new_sample'].get(rv)
            if s is not None:
                sample[k, i, :len(s)] = np.asarray(s, dtype=float)
    v2f = np.full((K, E, n), np.nan)
    f2v = np.full((K, E, n + T), np.nan)
    q = np.full((K, V, 2), np.nan)
    eta = np.full((K, E, 2), np.nan)
    for k in range(1, K + 1):
        st = snaps[k]
        v2f[k - 1] = _pad(st['v2f'], n)
        for e, row in enumerate(st['f2v']):
            if row is None:
                continue
            rv = edges[e][1]
            npts = len(snaps[k - 1]['new_sample'][rv]) if k - 1 < K else 0
            f2v[k - 1, e, :npts] = row[:npts]
            f2v[k - 1, e, n:n + len(row) - npts] = row[npts:]
        for i, val in st['q'].items():
            q[k - 1, i] = val
        for e, val in enumerate(st['eta']):
            if val is not None:
                eta[k - 1, e] = val
    return dict(sample=sample, v2f=v2f, f2v=f2v, q=q, eta=eta)


def model_hybrid_small(cg, discrete_evidence=True):
    """small hybrid MRF touching every device potential kind of the particle path: continuous + binary +
    3-state variables, quadratic / hybrid-quadratic / table / MLN (unary, pairwise, ternary) factors, evidence"""
    RG, RP, RM, modelio = cg.RG, cg.RP, cg.RM, cg.modelio
    dc = RG.Domain((-10, 10), continuous=True, integral_points=np.linspace(-10, 10, 20))
    db = RG.Domain((0, 1))
    d3 = RG.Domain((0, 1, 2))
    c = [RG.RV(dc) for _ in range(6)]
    b = [RG.RV(db) for _ in range(4)]
    t = [RG.RV(d3) for _ in range(2)]
    c[5].value = 2.5
    if discrete_evidence:   # the lifted solver averages evidence to a float, which cannot index a table
        b[3].value = 1
    quad = RP.QuadraticPotential(np.array([[-0.3, 0.1], [0.1, -0.25]]), np.array([0.2, -0.1]), 0.05)
    lin = RP.LinearGaussianPotential(0.8, 1.5)
    x2 = RP.X2Potential(1.0, 6.0)
    hq = RP.HybridQuadraticPotential(np.array([[[-0.2]], [[-0.5]]]), np.array([[0.5], [-0.4]]), np.array([0.0, 0.3]))
    tab = RP.TablePotential(np.array([[2.0, 0.5], [0.7, 1.5]]))
    tab3 = RP.TablePotential(np.array([[1.0, 0.5, 0.2], [0.3, 1.2, 0.9]]))
    m1 = RM.MLNPotential(modelio.FORMULAS['eq1'], w=0.3)
    m3 = RM.MLNPotential(modelio.FORMULAS['x0_eq12'], w=0.5)
    gauss = RP.GaussianPotential([0.5, -0.5], [[4.0, 1.5], [1.5, 3.0]])
    fs = [
        RG.F(quad, [c[0], c[1]]), RG.F(lin, [c[1], c[2]]), RG.F(gauss, [c[2], c[3]]), RG.F(lin, [c[3], c[4]]),
        RG.F(quad, [c[4], c[5]]), RG.F(lin, [c[0], c[4]]),
        RG.F(hq, [b[0], c[0]]), RG.F(hq, [b[1], c[2]]), RG.F(hq, [b[3], c[3]]),
        RG.F(tab, [b[0], b[1]]), RG.F(tab, [b[1], b[2]]), RG.F(tab, [b[2], b[3]]),
        RG.F(tab3, [b[2], t[0]]), RG.F(tab3, [b[0], t[1]]),
        RG.F(m3, [b[2], c[1], c[3]]), RG.F(m3, [b[3], c[0], c[5]]), RG.F(m3, [b[1], c[4], c[5]]),
        RG.F(m1, [c[1]]), RG.F(m1, [c[3]]),
    ]
    fs += [RG.F(x2, [c[i]]) for i in range(5)]
    g = RG.Graph()
    g.rvs = c + b + t
    g.factors = fs
    g.init_nb()
    return g


def model_rgm_c2f(cg, C=6, B=4):
    """RGM template with continuous evidence drawn from exactly two distinct values: the coarse start
    (init_cluster(False)) merges all of it, the first k-means split (k=2) separates the two values unambiguously, and
    the per-iteration split_rvs / split_factors then refine the structure while messages are inherited"""
    RG, RP = cg.RG, cg.RP
    d = RG.Domain((-30, 30), continuous=True, integral_points=np.linspace(-30, 30, 24))
    p1 = RP.GaussianPotential([0., 0.], [[10., -7.], [-7., 10.]])
    p2 = RP.GaussianPotential([0., 0.], [[10., 5.], [5., 10.]])
    p3 = RP.GaussianPotential([0., 0.], [[10., 7.], [7., 10.]])
    rec = RG.RV(d)
    market = [RG.RV(d) for _ in range(C)]
    loss = [[RG.RV(d) for _ in range(B)] for _ in range(C)]
    revenue = [RG.RV(d) for _ in range(B)]
    market[0].value = 3.0
    market[1].value = -2.0
    market[4].value = 3.0
    for c, b, v in ((2, 1, -2.0), (3, 1, -2.0), (5, 0, 3.0), (2, 3, 3.0), (0, 2, -2.0)):
        loss[c][b].value = v
    revenue[3].value = -2.0
    fs = [RG.F(p1, [rec, m]) for m in market]
    fs += [RG.F(p2, [market[c], loss[c][b]]) for c in range(C) for b in range(B)]
    fs += [RG.F(p3, [loss[c][b], revenue[b]]) for c in range(C) for b in range(B)]
    g = RG.Graph()
    g.rvs = [rec] + market + [x for row in loss for x in row] + revenue
    g.factors = fs
    g.init_nb()
    return g


def model_rgm_small(cg, C=4, B=3, seed=5):
    """the RGM template at toy size, evidence from two distinct values so lifting has something to merge"""
    RG, RP = cg.RG, cg.RP
    rng = np.random.RandomState(seed)
    d = RG.Domain((-30, 30), continuous=True, integral_points=np.linspace(-30, 30, 24))
    p1 = RP.GaussianPotential([0., 0.], [[10., -7.], [-7., 10.]])
    p2 = RP.GaussianPotential([0., 0.], [[10., 5.], [5., 10.]])
    p3 = RP.GaussianPotential([0., 0.], [[10., 7.], [7., 10.]])
    rec = RG.RV(d)
    market = [RG.RV(d) for _ in range(C)]
    loss = [[RG.RV(d) for _ in range(B)] for _ in range(C)]
    revenue = [RG.RV(d) for _ in range(B)]
    market[0].value = 3.0
    market[1].value = 3.0
    loss[2][1].value = -2.0
    loss[3][1].value = -2.0
    fs = [RG.F(p1, [rec, m]) for m in market]
    fs += [RG.F(p2, [market[c], loss[c][b]]) for c in range(C) for b in range(B)]
    fs += [RG.F(p3, [loss[c][b], revenue[b]]) for c in range(C) for b in range(B)]
    g = RG.Graph()
    g.rvs = [rec] + market + [x for row in loss for x in row] + revenue
    g.factors = fs
    g.init_nb()
    return g


def model_hmln_small(cg, P=6, T=3, seed=31, two_values=False, points=32, values=None):
    """cfg 3 at fixture size: the paper-popularity hybrid MLN template of
    Demo/Data/HMLN/GeneratorPaperPopularity.py:7-40 -- same atoms, same three parametric factors with their weights, the
    t1 != t2 constraint -- grounded by the reference's own RelationalGraph for P papers x T topics, with the demo's domain
    Domain((-15, 15), integral_points=linspace(0, 10, 32)).  Evidence follows generate_data (:51-72): 70 % of the
    popularity atoms ~ U(0, 10) (two_values: drawn from {2.5, 7.0}, so a k-means split is seeding-independent), PaperIn
    of 70 % of the papers for a random subset of topics, half of the SameSession atoms."""
    import RelationalGraph as RR
    RG, RM, modelio = cg.RG, cg.RM, cg.modelio
    rng = np.random.RandomState(seed)
    papers = ['p%d' % i for i in range(P)]
    topics = ['t%d' % i for i in range(T)]
    d_bool = RG.Domain((0, 1))
    d_real = RG.Domain((-15, 15), continuous=True, integral_points=np.linspace(0, 10, points))
    lv_p, lv_t = RR.LV(papers), RR.LV(topics)
    atoms = (RR.Atom(d_bool, logical_variables=(lv_t, lv_t), name='SameSession'),
             RR.Atom(d_bool, logical_variables=(lv_p, lv_t), name='PaperIn'),
             RR.Atom(d_real, logical_variables=(lv_t,), name='TopicPopularity'),
             RR.Atom(d_real, logical_variables=(lv_p,), name='PaperPopularity'))
    f0 = RR.ParamF(RM.MLNPotential(modelio.FORMULAS['eq1'], w=0.3), nb=['PaperPopularity(p)'])
    f1 = RR.ParamF(RM.MLNPotential(modelio.FORMULAS['x0_eq12'], w=0.5),
                   nb=['SameSession(t1,t2)', 'TopicPopularity(t1)', 'TopicPopularity(t2)'],
                   constrain=lambda sub: sub['t1'] != sub['t2'])
    f2 = RR.ParamF(RM.MLNPotential(modelio.FORMULAS['x0_eq12'], w=1),
                   nb=['PaperIn(p,t)', 'PaperPopularity(p)', 'TopicPopularity(t)'])
    rel_g = RR.RelationalGraph(atoms, (f0, f1, f2))
    rel_g.ground_graph()
    if values is not None:
        draw = lambda: float(rng.choice(list(values)))
    else:
        draw = (lambda: float(rng.choice([2.5, 7.0]))) if two_values else (lambda: float(rng.uniform(0, 10)))
    data = {}
    for x in rng.choice(P, int(P * 0.7), replace=False):
        data[('PaperPopularity', 'p%d' % x)] = draw()
    for x in rng.choice(T, int(T * 0.7), replace=False):
        data[('TopicPopularity', 't%d' % x)] = draw()
    for x in rng.choice(P, int(P * 0.7), replace=False):
        for y in rng.choice(T, rng.randint(T), replace=False):
            data[('PaperIn', 'p%d' % x, 't%d' % y)] = int(rng.choice([0, 1]))
    for x in range(T):
        for y in range(T):
            if x != y and rng.rand() < 0.5:
                data[('SameSession', 't%d' % x, 't%d' % y)] = int(rng.choice([0, 1]))
    g, rvs_dict = rel_g.add_evidence(data)
    # deterministic iteration order for the fixture: rvs in rvs_dict (first use) order, factors by (template, scope)
    rvs = list(rvs_dict.values())
    idx = {id(rv): i for i, rv in enumerate(rvs)}
    tmpl = {id(f0.potential): 0, id(f1.potential): 1, id(f2.potential): 2}
    g.rvs = rvs
    g.factors = sorted(g.factors, key=lambda f: (tmpl[id(f.potential)], [idx[id(r)] for r in f.nb]))
    g.init_nb()
    return g


def _edges(g):
    return [(f, rv) for f in g.factors for rv in f.nb]


def _safe(fn, *args):
    """a query of the reference, NaN where the reference itself fails: its normalisers exponentiate unnormalised
    log-beliefs (EPBP:342, HLBP:372), which overflows on models with strong evidence (the robot-mapping HMLN)"""
    try:
        return float(fn(*args))
    except OverflowError:
        return float('nan')


def capture_epbp(cg, name, g, n, its, approx, seed):
    import EPBPLogVersion as REP
    np.random.seed(seed)
    bp = REP.EPBP(g, n=n, proposal_approximation=approx)
    edges = _edges(g)
    with cg.quiet():
        snaps = run_with_snapshots(bp, g.rvs, edges, its)
    rec = pack(snaps, g.rvs, edges, n)
    # post-sweep queries (A8): log-belief at a few points, map, for every hidden rv
    xs, lb, mp = [], [], []
    for rv in g.rvs:
        if rv.value is not None:
            xs.append([np.nan] * 5); lb.append([np.nan] * 5); mp.append(float(rv.value)); continue
        if rv.domain.continuous:
            pts = np.linspace(rv.domain.values[0] * 0.6, rv.domain.values[1] * 0.6, 5)
        else:
            pts = (list(rv.domain.values) + [rv.domain.values[0]] * 5)[:5]
        xs.append([float(x) for x in pts])
        lb.append([float(bp.belief_rv(x, rv, bp.sample)) for x in pts])
        mp.append(float(bp.map(rv)))
    rec.update(query_x=np.array(xs), query_logb=np.array(lb), map=np.array(mp))
    # normalised beliefs for a couple of rvs (quad-based normaliser; slow, so only two continuous + all discrete)
    nb = []
    done_c = 0
    for i, rv in enumerate(g.rvs):
        if rv.value is not None:
            continue
        if rv.domain.continuous:
            if done_c >= 2:
                continue
            done_c += 1
            x0 = xs[i][2]
            nb.append([i, x0, _safe(bp.belief, x0, rv)])
        else:
            nb.append([i, float(rv.domain.values[0]), _safe(bp.belief, rv.domain.values[0], rv)])
    rec['belief'] = np.array(nb)
    # interval probabilities (EPBP:356-375: 5-point over 20-point trapezoid) for every hidden continuous rv
    pr = []
    for i, rv in enumerate(g.rvs):
        if rv.value is None and rv.domain.continuous:
            lo, hi = rv.domain.values
            a, b = lo + 0.35 * (hi - lo), lo + 0.6 * (hi - lo)
            pr.append([i, a, b, _safe(bp.probability, a, b, rv)])
    rec['probability'] = np.array(pr).reshape(-1, 4)
    rec['meta'] = json.dumps({'model': cg.modelio.dump_model(g), 'n': n, 'iterations': its, 'approx': approx,
                              'seed': seed, 'solver': 'EPBP'})
    path = os.path.join(cg.OUT, 'pbp_%s.npz' % name)
    np.savez_compressed(path, **rec)
    print('wrote', path, os.path.getsize(path), 'bytes')


def capture_hlbp(cg, name, g, n, its, approx, seed, c2f=-1):
    import HybridLBPLogVersion as RH
    np.random.seed(seed)
    bp = RH.HybridLBP(g, n=n, proposal_approximation=approx)
    drawn, labels, draw_q, draw_msg, draw_eta = [], [], [], [], []
    orig = bp.generate_sample

    def wrapped():
        new = orig()
        # the draw, broadcast to the ground rvs NOW (cluster objects are reused and shrink when they split later)
        sk = np.full((len(g.rvs), n), np.nan)
        member = {}
        for c, smp in new.items():
            for rv in c.rvs:
                member[id(rv)] = smp
        for i, rv in enumerate(g.rvs):
            if id(rv) in member:
                arr = np.asarray(member[id(rv)], dtype=float)
                sk[i, :len(arr)] = arr
        drawn.append(sk)
        # partition at the moment of the draw (coarse-to-fine runs refine it between draws)
        labels.append([cg.partition_labels(g.rvs, bp.g.rvs, 'rvs'), cg.partition_labels(g.factors, bp.g.factors, 'factors')])
        qk = np.full((len(g.rvs), 2), np.nan)          # proposals at the moment of the draw, per ground rv
        member_q = {}
        for c, val in bp.q.items():
            for rv in c.rvs:
                member_q[id(rv)] = val
        for i, rv in enumerate(g.rvs):
            if id(rv) in member_q:
                qk[i] = member_q[id(rv)]
        draw_q.append(qk)
        # per ground rv and per ground factor of it: f->rv log message at the integral points and the site (eta)
        T = max(len(rv.domain.integral_points) for rv in g.rvs if rv.domain.continuous)
        deg = max(len(rv.nb) for rv in g.rvs)
        mg = np.full((len(g.rvs), deg, T), np.nan)
        et = np.full((len(g.rvs), deg, 2), np.nan)
        for i, rv in enumerate(g.rvs):
            if rv.value is not None or not rv.domain.continuous:
                continue
            for s_, f in enumerate(rv.nb):
                key = (f.cluster, rv.cluster)
                if key in bp.message:
                    mg[i, s_] = [bp.message[key][x] for x in rv.domain.integral_points]
                if key in bp.eta_message:
                    et[i, s_] = bp.eta_message[key]
        draw_msg.append(mg)
        draw_eta.append(et)
        return new

    bp.generate_sample = wrapped
    with cg.quiet():
        bp.run(its, c2f=c2f)
    bp.generate_sample = orig
    rv_label = cg.partition_labels(g.rvs, bp.g.rvs, 'rvs')
    f_label = cg.partition_labels(g.factors, bp.g.factors, 'factors')
    xs, lb, mp, bel = [], [], [], []
    for rv in g.rvs:
        if rv.value is not None:
            xs.append([np.nan] * 5); lb.append([np.nan] * 5); mp.append(float(rv.value)); bel.append(np.nan); continue
        if rv.domain.continuous:
            pts = np.linspace(rv.domain.values[0] * 0.6, rv.domain.values[1] * 0.6, 5)
        else:
            pts = (list(rv.domain.values) + [rv.domain.values[0]] * 5)[:5]
        xs.append([float(x) for x in pts])
        lb.append([float(bp.belief_rv_query(x, rv, bp.sample)) for x in pts])
        mp.append(float(bp.map(rv)))
        bel.append(_safe(bp.belief, pts[2], rv))
    pr = []                # HLBP:384-403 interval probabilities
    for i, rv in enumerate(g.rvs):
        if rv.value is None and rv.domain.continuous:
            lo, hi = rv.domain.values
            a, b = lo + 0.35 * (hi - lo), lo + 0.6 * (hi - lo)
            pr.append([i, a, b, _safe(bp.probability, a, b, rv)])
    # final samples / q per ground rv (through its cluster)
    V = len(g.rvs)
    sample = np.full((V, n), np.nan)
    q = np.full((V, 2), np.nan)
    for i, rv in enumerate(g.rvs):
        c = rv.cluster
        if c in bp.sample:
            s = np.asarray(bp.sample[c], dtype=float)
            sample[i, :len(s)] = s
        if c in bp.q:
            q[i] = bp.q[c]
    samples = np.array(drawn)                         # every draw, broadcast to the ground rvs at draw time
    rec = dict(samples=samples, draw_q=np.array(draw_q), draw_f2v_grid=np.array(draw_msg), draw_eta=np.array(draw_eta),
               draw_rv_labels=np.array([l[0] for l in labels]), draw_f_labels=np.array([l[1] for l in labels]),
               rv_label=np.array(rv_label), f_label=np.array(f_label), query_x=np.array(xs), query_logb=np.array(lb),
               map=np.array(mp), belief_mid=np.array(bel), final_sample=sample, final_q=q,
               probability=np.array(pr).reshape(-1, 4),
               meta=json.dumps({'model': cg.modelio.dump_model(g), 'n': n, 'iterations': its, 'approx': approx,
                                'seed': seed, 'solver': 'HybridLBP', 'c2f': c2f}))
    path = os.path.join(cg.OUT, 'pbp_%s.npz' % name)
    np.savez_compressed(path, **rec)
    print('wrote', path, os.path.getsize(path), 'bytes')


def capture_pbp(cg):
    capture_epbp(cg, 'epbp_kalman_simple', cg.model_kalman(3, 4, 2), 16, 4, 'simple', 11)
    capture_epbp(cg, 'epbp_kalman_ep', cg.model_kalman(3, 4, 2), 16, 4, 'EP', 12)
    capture_epbp(cg, 'epbp_kalman_n64', cg.model_kalman(2, 3, 3), 64, 3, 'simple', 13)
    capture_epbp(cg, 'epbp_hybrid_ep', model_hybrid_small(cg), 10, 4, 'EP', 14)
    capture_epbp(cg, 'epbp_hybrid_simple', model_hybrid_small(cg), 12, 3, 'simple', 15)
    capture_hlbp(cg, 'hlbp_rgm_small', model_rgm_small(cg), 10, 4, 'EP', 21)
    capture_hlbp(cg, 'hlbp_hybrid', model_hybrid_small(cg, False), 10, 3, 'simple', 22)
    capture_hlbp(cg, 'hlbp_kalman_full', cg.model_kalman(3, 5, 1, False), 12, 4, 'EP', 23)
    capture_hlbp(cg, 'hlbp_c2f_rgm', model_rgm_c2f(cg), 10, 5, 'EP', 24, c2f=0)
    capture_hlbp(cg, 'hlbp_c2f_rgm_simple', model_rgm_c2f(cg), 12, 4, 'simple', 25, c2f=0)
    # cfg 3: paper-popularity HMLN (6 papers x 3 topics), T = 32 integral points, the demo's n = 10 / 'simple'
    capture_hlbp(cg, 'hlbp_hmln', model_hmln_small(cg), 10, 4, 'simple', 26)
    capture_hlbp(cg, 'hlbp_hmln_ep', model_hmln_small(cg, 5, 3, seed=33), 10, 3, 'EP', 27)
    capture_hlbp(cg, 'hlbp_hmln_lifted', model_hmln_small(cg, two_values=True, seed=37), 10, 4, 'simple', 30)
    capture_hlbp(cg, 'hlbp_c2f_hmln', model_hmln_small(cg, two_values=True), 10, 4, 'simple', 28, c2f=0)
    capture_epbp(cg, 'epbp_hmln', model_hmln_small(cg, 4, 3, seed=35), 10, 3, 'simple', 29)
