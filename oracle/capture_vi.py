"""Golden vectors for the mixture variational path (VarInference / LiftedVarInference), captured from the reference.

TEST INFRASTRUCTURE, build container only (see capture_golden.py).  For each model: seed NumPy, let the reference draw
its initial parameters, record them together with every gradient and the free energy at that point, then run a few ADAM
iterations and record the per-iteration free energy and the final parameters.
"""
import json
import os

import numpy as np


def _params(vi, rvs, cluster_of):
    """per ground rv: eta rows of its (super) rv; continuous -> [K,2], discrete -> [K,d]"""
    K = vi.K
    dmax = max([len(rv.domain.values) for rv in rvs if not rv.domain.continuous] + [1])
    eta_c = np.full((len(rvs), K, 2), np.nan)
    eta_d = np.full((len(rvs), K, dmax), np.nan)
    tau_d = np.full((len(rvs), K, dmax), np.nan)
    for i, rv in enumerate(rvs):
        c = cluster_of(rv)
        if c.value is not None or c not in vi.eta:
            continue
        if rv.domain.continuous:
            eta_c[i] = vi.eta[c]
        else:
            d = len(rv.domain.values)
            eta_d[i, :, :d] = vi.eta[c]
            tau_d[i, :, :d] = vi.eta_tau[c]
    return eta_c, eta_d, tau_d


def _grads(vi, rvs, cluster_of):
    K = vi.K
    dmax = max([len(rv.domain.values) for rv in rvs if not rv.domain.continuous] + [1])
    g_c = np.full((len(rvs), K, 2), np.nan)
    g_d = np.full((len(rvs), K, dmax), np.nan)
    for i, rv in enumerate(rvs):
        c = cluster_of(rv)
        if c.value is not None:
            continue
        if rv.domain.continuous:
            g_c[i] = vi.gradient_mu_var(c)
        else:
            d = len(rv.domain.values)
            g_d[i, :, :d] = vi.gradient_category_tau(c)
    return g_c, g_d


def capture_one(cg, name, g, lifted, K, T, seed, iterations, lr=0.1):
    import VarInference as RVI
    import LiftedVarInference as RLVI
    cls = RLVI.VarInference if lifted else RVI.VarInference
    vi = cls(g, K, T)
    cluster_of = (lambda rv: rv.cluster) if lifted else (lambda rv: rv)
    np.random.seed(seed)
    vi.init_param()
    rec = {}
    rec['w_tau0'] = np.array(vi.w_tau, dtype=float)
    rec['eta_c0'], rec['eta_d0'], rec['tau_d0'] = _params(vi, g.rvs, cluster_of)
    rec['g_w0'] = np.array(vi.gradient_w_tau(), dtype=float)
    rec['g_c0'], rec['g_d0'] = _grads(vi, g.rvs, cluster_of)
    rec['fe0'] = np.array(float(vi.free_energy()))
    np.random.seed(seed)
    with cg.quiet():
        vi.run(iterations, lr=lr)
    rec['fe_log'] = np.array([x[1] for x in vi.time_log], dtype=float)
    rec['w_final'] = np.array(vi.w, dtype=float)
    rec['eta_c_final'], rec['eta_d_final'], _ = _params(vi, g.rvs, cluster_of)
    hid = [rv for rv in g.rvs if rv.value is None]
    # queries take the GROUND rv in both classes (the lifted one goes through rv.cluster itself)
    rec['belief_mid'] = np.array([float(vi.belief(0.5 if rv.domain.continuous else rv.domain.values[0], rv))
                                  if rv.value is None else np.nan for rv in g.rvs])
    rec['map'] = np.array([float(vi.map(rv)) for rv in g.rvs])
    if lifted:
        rec['rv_label'] = np.array(cg.partition_labels(g.rvs, vi.g.rvs, 'rvs'))
        rec['f_label'] = np.array(cg.partition_labels(g.factors, vi.g.factors, 'factors'))
    rec['meta'] = json.dumps({'model': cg.modelio.dump_model(g), 'K': K, 'T': T, 'seed': seed, 'iterations': iterations,
                              'lr': lr, 'lifted': lifted})
    path = os.path.join(cg.OUT, 'vi_%s.npz' % name)
    np.savez_compressed(path, **rec)
    print('wrote', path, os.path.getsize(path), 'bytes')


def capture_vi(cg):
    from capture_pbp import model_hybrid_small, model_rgm_small
    capture_one(cg, 'kalman_k1', cg.model_kalman(3, 4, 2), False, 1, 3, 31, 3)
    capture_one(cg, 'kalman_k3', cg.model_kalman(3, 4, 2), False, 3, 3, 32, 3)
    capture_one(cg, 'hybrid_k2', model_hybrid_small(cg), False, 2, 3, 33, 3)
    capture_one(cg, 'hybrid_k1_t5', model_hybrid_small(cg), False, 1, 5, 34, 2)
    capture_one(cg, 'rgm_small_k2', model_rgm_small(cg), False, 2, 3, 35, 3)
    capture_one(cg, 'lifted_rgm_small_k2', model_rgm_small(cg), True, 2, 3, 36, 3)
    capture_one(cg, 'lifted_kalman_full_k2', cg.model_kalman(3, 5, 1, False), True, 2, 3, 37, 3)
    capture_one(cg, 'lifted_hybrid_k2', model_hybrid_small(cg, False), True, 2, 3, 38, 2)


# ---------------------------------------------------------------------------------------------------------------------
# C2FVarInference (coarse-to-fine lifted VI with Gaussian observation clusters), C2FVarInference.py:33-68,301-352
def model_rgm_c2fvi(cg, C=6, B=4):
    """RGM template, every market observed at -3 or 3 and every revenue at -10 or 10, losses and recession hidden.  The
    coarse start merges all evidence; the first structural refinement separates markets from revenues; with the shrinking
    threshold (max std 10 -> 6.7 -> 3.3 -> 0) the revenue cluster is split into its two values before the first ADAM round
    while the market cluster (std 3) stays ONE Gaussian observation N(0, 9) for two rounds and is split in the third.
    No cluster ever holds more than k = 2 distinct values, so k-means does not depend on its seeding (the reference seeds
    it in set order)."""
    RG, RP = cg.RG, cg.RP
    d = RG.Domain((-30, 30), continuous=True, integral_points=np.linspace(-30, 30, 24))
    p1 = RP.GaussianPotential([0., 0.], [[10., -7.], [-7., 10.]])
    p2 = RP.GaussianPotential([0., 0.], [[10., 5.], [5., 10.]])
    p3 = RP.GaussianPotential([0., 0.], [[10., 7.], [7., 10.]])
    rec = RG.RV(d)
    market = [RG.RV(d, 3.0 if c % 2 == 0 else -3.0) for c in range(C)]
    loss = [[RG.RV(d) for _ in range(B)] for _ in range(C)]
    revenue = [RG.RV(d, 10.0 if b % 2 == 0 else -10.0) for b in range(B)]
    fs = [RG.F(p1, [rec, m]) for m in market]
    fs += [RG.F(p2, [market[c], loss[c][b]]) for c in range(C) for b in range(B)]
    fs += [RG.F(p3, [loss[c][b], revenue[b]]) for c in range(C) for b in range(B)]
    g = RG.Graph()
    g.rvs = [rec] + market + [x for row in loss for x in row] + revenue
    g.factors = fs
    g.init_nb()
    return g


def model_hmln_c2fvi(cg, P=5, T=3, seed=43):
    """the paper-popularity template (see capture_pbp.model_hmln_small) with popularity evidence of at most two values per
    atom type: papers p0..p3 at 2 / 3 (std 0.5: a Gaussian observation until the last round), topics t0, t1 at 1 / 9
    (split before the first round), p4 and t2 hidden; PaperIn(p, t0) = PaperIn(p, t1) = 1 for the observed papers.  Exercises Gaussian observations
    next to discrete hidden variables (gradient_category_tau, C2FVarInference.py:207-239)"""
    import RelationalGraph as RR
    RG, RM, modelio = cg.RG, cg.RM, cg.modelio
    rng = np.random.RandomState(seed)
    papers, topics = ['p%d' % i for i in range(P)], ['t%d' % i for i in range(T)]
    d_bool = RG.Domain((0, 1))
    d_real = RG.Domain((-15, 15), continuous=True, integral_points=np.linspace(0, 10, 32))
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
    data = {}
    for i in range(P - 1):
        data[('PaperPopularity', 'p%d' % i)] = 2.0 if i % 2 == 0 else 3.0
    data[('TopicPopularity', 't0')], data[('TopicPopularity', 't1')] = 1.0, 9.0
    for i in range(P - 1):          # the same boolean evidence around every observed paper / topic: they stay exchangeable
        data[('PaperIn', 'p%d' % i, 't0')] = 1
        data[('PaperIn', 'p%d' % i, 't1')] = 1
    g, rvs_dict = rel_g.add_evidence(data)
    rvs = list(rvs_dict.values())
    idx = {id(rv): i for i, rv in enumerate(rvs)}
    tmpl = {id(f0.potential): 0, id(f1.potential): 1, id(f2.potential): 2}
    g.rvs = rvs
    g.factors = sorted(g.factors, key=lambda f: (tmpl[id(f.potential)], [idx[id(r)] for r in f.nb]))
    g.init_nb()
    return g


def _c2f_state(cg, vi, g):
    """what the run holds at this moment, per GROUND rv / factor: partition, evidence clusters' (value, variance), whether
    the evidence cluster is tracked for further splitting, the parameters and ADAM moments through rv.cluster"""
    rvs, K = list(g.rvs), vi.K
    dmax = max([len(rv.domain.values) for rv in rvs if not rv.domain.continuous] + [1])
    st = dict(rv_label=np.array(cg.partition_labels(rvs, vi.g.rvs, 'rvs')),
              f_label=np.array(cg.partition_labels(list(g.factors), vi.g.factors, 'factors')),
              value=np.full(len(rvs), np.nan), variance=np.full(len(rvs), np.nan), tracked=np.zeros(len(rvs), dtype=np.int8),
              eta_c=np.full((len(rvs), K, 2), np.nan), tau_d=np.full((len(rvs), K, dmax), np.nan),
              m_c=np.full((len(rvs), K, 2), np.nan), s_c=np.full((len(rvs), K, 2), np.nan),
              w_tau=np.array(vi.w_tau, dtype=float), t=np.array(vi.t))
    for i, rv in enumerate(rvs):
        c = rv.cluster
        if c.value is not None:
            st['value'][i], st['variance'][i] = float(c.value), float(c.variance)
            st['tracked'][i] = 1 if c in vi.g.clustered_evidence else 0
        elif rv.domain.continuous:
            st['eta_c'][i] = vi.eta[c]
            st['m_c'][i], st['s_c'][i] = vi.eta_g[0][c], vi.eta_g[1][c]
        else:
            d = len(rv.domain.values)
            st['tau_d'][i, :, :d] = vi.eta_tau[c]
    return st


def capture_c2fvi(cg, name, g, K, T, seed, iterations, lr, update_obs_its=10, log_fe=True):
    import C2FVarInference as RC
    vi = RC.VarInference(g, K, T)
    vi.update_obs_its = update_obs_its
    snaps = []
    orig_adam, orig_init = vi.ADAM_update, vi.init_param

    def init_param():
        orig_init()
        snaps.append(('init', _c2f_state_init(cg, vi, g)))

    def adam(n):
        snaps.append(('round', _c2f_state(cg, vi, g)))
        orig_adam(n)
    vi.init_param, vi.ADAM_update = init_param, adam
    # k-means of an evidence cluster is seeded in the iteration order of a Python set of RV objects (CGWO:83-97), which
    # differs from run to run: record the order every split saw (observation only -- the original method runs unchanged)
    import CompressedGraphWithObs as CGWO
    index = {id(rv): i for i, rv in enumerate(g.rvs)}
    orders = {}
    orig_split = CGWO.SuperRV.split_by_evidence

    def split_by_evidence(self, k=2, iteration=10):
        seen = [index[id(rv)] for rv in self.rvs]
        orders[','.join(map(str, sorted(seen)))] = seen
        return orig_split(self, k, iteration)
    CGWO.SuperRV.split_by_evidence = split_by_evidence
    np.random.seed(seed)
    try:
        with cg.quiet():
            vi.run(iterations, lr=lr, log_fe=log_fe)
    finally:
        CGWO.SuperRV.split_by_evidence = orig_split
    vi.ADAM_update = orig_adam
    rec = {}
    init = [s for k, s in snaps if k == 'init'][0]
    rounds = [s for k, s in snaps if k == 'round']
    rec['eta_c0'], rec['tau_d0'] = init['eta_c'], init['tau_d']
    for key in ('rv_label', 'f_label', 'value', 'variance', 'tracked', 'eta_c', 'tau_d', 'm_c', 's_c', 'w_tau', 't'):
        rec['round_' + key] = np.array([r[key] for r in rounds])
    final = _c2f_state(cg, vi, g)
    for key in ('rv_label', 'f_label', 'eta_c', 'tau_d', 'w_tau', 'value', 'variance'):
        rec['final_' + key] = final[key]
    rec['fe_log'] = np.array([x[1] for x in vi.time_log], dtype=float)
    rec['w_final'] = np.array(vi.w, dtype=float)
    rec['fe_final'] = np.array(float(vi.free_energy()))
    rvs = list(g.rvs)
    rec['belief_mid'] = np.array([float(vi.belief(0.5 if rv.domain.continuous else rv.domain.values[0], rv))
                                  if rv.value is None else np.nan for rv in rvs])
    rec['map'] = np.array([float(vi.map(rv)) for rv in rvs])
    rec['meta'] = json.dumps({'model': cg.modelio.dump_model(g), 'K': K, 'T': T, 'seed': seed, 'iterations': iterations,
                              'lr': lr, 'update_obs_its': update_obs_its, 'solver': 'C2FVarInference', 'log_fe': log_fe,
                              'kmeans_orders': orders})
    path = os.path.join(cg.OUT, 'vi_%s.npz' % name)
    np.savez_compressed(path, **rec)
    print('wrote', path, os.path.getsize(path), 'bytes', 'rounds', len(rounds),
          'rv clusters per round', [len(set(r['rv_label'].tolist())) for r in rounds], len(set(final['rv_label'].tolist())))
    return rec


def _c2f_state_init(cg, vi, g):
    """initial parameters per ground rv (the coarse clusters share one random draw each)"""
    rvs, K = list(g.rvs), vi.K
    dmax = max([len(rv.domain.values) for rv in rvs if not rv.domain.continuous] + [1])
    eta_c = np.full((len(rvs), K, 2), np.nan)
    tau_d = np.full((len(rvs), K, dmax), np.nan)
    for i, rv in enumerate(rvs):
        c = rv.cluster
        if c.value is not None:
            continue
        if rv.domain.continuous:
            eta_c[i] = vi.eta[c]
        else:
            tau_d[i, :, :len(rv.domain.values)] = vi.eta_tau[c]
    return dict(eta_c=eta_c, tau_d=tau_d)


def capture_c2f(cg):
    capture_c2fvi(cg, 'c2f_rgm_k2', model_rgm_c2fvi(cg), 2, 3, 41, 30, 0.1)
    capture_c2fvi(cg, 'c2f_hmln_k2', model_hmln_c2fvi(cg), 2, 3, 42, 30, 0.2)


def capture_c2f_loglik(cg):
    """the run's OTHER log (C2FVarInference.py:393-404, run(log_fe=False)): -log phi of the ground graph at the current MAP after
    every update -- the quantity of the reference's published HMLN logs (Demo/HMLN/HMLNTimeLog.py:57)"""
    capture_c2fvi(cg, 'c2f_rgm_k2_loglik', model_rgm_c2fvi(cg), 2, 3, 44, 20, 0.1, log_fe=False)
    capture_c2fvi(cg, 'c2f_hmln_k2_loglik', model_hmln_c2fvi(cg), 2, 3, 45, 20, 0.2, log_fe=False)
