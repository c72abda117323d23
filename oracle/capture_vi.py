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
