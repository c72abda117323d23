"""CPU suite: the variational oracle reproduces the vectors captured from the reference's VarInference / LiftedVarInference."""
import json
import os

import numpy as np
import pytest

import modelio
from test_oracle_golden import API, _initial
from lhvi import lifting
from lhvi.flat import flatten
from oracle import oracle

VI_CASES = ['kalman_k1', 'kalman_k3', 'hybrid_k2', 'hybrid_k1_t5', 'rgm_small_k2']
LVI_CASES = ['lifted_rgm_small_k2', 'lifted_kalman_full_k2', 'lifted_hybrid_k2', 'lifted_robot_k2']


def load_vi(golden_dir, name):
    z = np.load(os.path.join(golden_dir, 'vi_%s.npz' % name))
    return z, json.loads(str(z['meta']))


def build(golden_dir, name):
    """(z, meta, rvs, flat, gather): flat is ground or lifted; gather[i] = flat variable of ground rv i"""
    z, meta = load_vi(golden_dir, name)
    g, rvs, factors = modelio.load_model(meta['model'], API)
    if not meta['lifted']:
        flat = flatten(g, require_device_potentials=True)
        return z, meta, rvs, flat, np.arange(len(rvs))
    gflat = flatten(g)
    sym, rv0, f0 = _initial(gflat, g)
    rv_color, f_color = oracle.color_passing(gflat, sym, rv0, f0)
    assert oracle.canonical_labels(rv_color) == z['rv_label'].tolist()
    assert oracle.canonical_labels(f_color) == z['f_label'].tolist()
    cg = lifting.CompressedGraph(g)
    cg.set_colors(rv_color, f_color)
    flat = flatten(cg, require_device_potentials=True)
    return z, meta, rvs, flat, np.array([flat.var_index[rv.cluster] for rv in rvs])


def scatter_params(z, flat, gather, key):
    """golden arrays are per ground rv; the lifted flat graph wants them per cluster"""
    src = z[key]
    out = np.full((flat.V,) + src.shape[1:], np.nan)
    out[gather] = src
    return out


@pytest.mark.parametrize('name', VI_CASES + LVI_CASES)
def test_vi_oracle_matches_reference(golden_dir, name):
    z, meta, rvs, flat, gather = build(golden_dir, name)
    o = oracle.ViOracle(flat, meta['K'], meta['T'], quirks=1)
    o.set_params(z['w_tau0'], scatter_params(z, flat, gather, 'eta_c0'), scatter_params(z, flat, gather, 'tau_d0'))
    g_w, g_c, g_d, fe = o.grad()
    # fp64; the oracle follows the reference's loop order, differences are libm-level
    np.testing.assert_allclose(fe, float(z['fe0']), rtol=1e-10)
    np.testing.assert_allclose(g_w, z['g_w0'], rtol=1e-8, atol=1e-9)
    cont = np.array([rv.value is None and rv.domain.continuous for rv in rvs])
    disc = np.array([rv.value is None and not rv.domain.continuous for rv in rvs])
    np.testing.assert_allclose(g_c[gather][cont], z['g_c0'][cont], rtol=1e-8, atol=1e-9)
    if disc.any():
        D = z['g_d0'].shape[2]
        want = np.nan_to_num(z['g_d0'][disc], nan=0.0)
        np.testing.assert_allclose(g_d[gather][disc][:, :, :D], want, rtol=1e-8, atol=1e-9)
    log = o.run(meta['iterations'], meta['lr'])
    np.testing.assert_allclose(log, z['fe_log'], rtol=1e-8)
    np.testing.assert_allclose(o.w, z['w_final'], rtol=1e-8)
    np.testing.assert_allclose(o.eta_c[gather][cont], z['eta_c_final'][cont], rtol=1e-8, atol=1e-10)


# ---- C2FVarInference (coarse-to-fine lifted VI with Gaussian observation clusters) -----------------------------------
C2F_CASES = ['c2f_rgm_k2', 'c2f_hmln_k2', 'c2f_robot_k2', 'c2f_rkf_tree_k1', 'c2f_rkf_cycle_k1']
C2F_OPTS = dict(k_mean_k=2, k_mean_its=10, output_its=0, min_obs_var=0, gaussian_obs=True)


class OracleViEngine:
    """lhvi.c2fvi engine backed by the CPU oracle"""

    def __init__(self, K, T):
        self.K, self.T = K, T

    def stage(self, flat, obs_var):
        eng = self

        class Stage:
            def __init__(self):
                self.o = oracle.ViOracle(flat, eng.K, eng.T, quirks=1, obs_var=obs_var)
                self.mom = {}

            def load(self, P):
                self.o.set_params(P['w_tau'], P['eta_c'], P['tau_d'])
                for name in ('w_tau', 'eta_c', 'tau_d'):
                    ref = getattr(self.o, name)
                    for pre in ('m_', 's_'):
                        a = np.zeros_like(ref)
                        src = np.asarray(P[pre + name], dtype=float)
                        a[..., :min(a.shape[-1], src.shape[-1])] = src[..., :a.shape[-1]]
                        self.mom[pre + name] = a

            loglik = None      # set by the schedule for run(log_fe=False)

            def adam(self, n, t, lr):
                if self.loglik is None:
                    return self.o.run(n, lr, moments=self.mom, t0=t)
                out = []
                for i in range(n):                        # C2FVI:393-404: -log phi of the ground graph at the MAP after every update
                    self.o.run(1, lr, moments=self.mom, t0=t + i)
                    out.append(self.loglik(_OracleRows(self.o, flat).map_rows()))
                return out

            def dump(self):
                return dict(w_tau=self.o.w_tau, eta_c=self.o.eta_c, tau_d=self.o.tau_d, **self.mom)
        return Stage()


class _OracleRows:
    """the product's host-side ``map_rows`` (lhvi/vi.py: scipy.optimize.minimize per hidden row, as VI:355-376) on the CPU
    oracle's parameters"""
    from lhvi.vi import _Variational as _V
    norm_pdf, map_rows, _row_belief = staticmethod(_V.norm_pdf), _V.map_rows, _V._row_belief

    def __init__(self, o, flat):
        self.o, self.flat, self.K = o, flat, o.K

    def _w_host(self):
        return self.o.w

    def _host(self, name):
        return getattr(self.o, name)


def test_c2fvi_logs_the_map_likelihood_like_the_reference(golden_dir):
    """``run(log_fe=False)`` (C2FVI:393-404): after every update the reference logs ``log_likelihood`` of the GROUND graph at the
    current MAP of every ground variable; the schedule computes one MAP per cluster and spreads it (CPU oracle as the engine)"""
    from lhvi import c2fvi
    from test_oracle_pbp import OracleRefiner
    for name in ('c2f_rgm_k2_loglik', 'c2f_hmln_k2_loglik'):
        z, meta = load_vi(golden_dir, name)
        assert meta['log_fe'] is False
        g, rvs, factors = modelio.load_model(meta['model'], API)
        seen = []
        res = c2fvi.run_c2fvi(g, OracleViEngine(meta['K'], meta['T']), OracleRefiner(g), meta['K'], meta['iterations'], meta['lr'],
                              dict(C2F_OPTS, update_obs_its=meta['update_obs_its'], kmeans_member_order=kmeans_order_of(meta),
                                   log_map_likelihood=True),
                              init=(z['eta_c0'], z['tau_d0']), observer=c2fvi_round_checker(z, rvs, seen))
        assert len(res['fe_log']) == meta['iterations'] == len(z['fe_log'])
        # (a MAP is a BFGS answer to ~1e-6; -log phi is quadratic around it)
        np.testing.assert_allclose(res['fe_log'], z['fe_log'], rtol=1e-6, atol=1e-6)
        assert not np.allclose(res['fe_log'], [0.0] * len(z['fe_log']))


def c2fvi_round_checker(z, rvs, seen):
    """observer for lhvi.c2fvi.run_c2fvi: the state right before every round's ADAM updates against what the reference
    held at that point -- partitions (exact), evidence clusters' value / variance and clustered_evidence membership,
    inherited parameters and ADAM moments, update counter"""
    cont = np.array([rv.value is None and rv.domain.continuous for rv in rvs])
    disc = np.array([rv.value is None and not rv.domain.continuous for rv in rvs])
    ev = np.array([rv.value is not None for rv in rvs])

    def observer(r, st):
        rvc, flat, P = st['rvc'], st['flat'], st['params']
        assert oracle.canonical_labels(rvc) == z['round_rv_label'][r].tolist(), 'rv partition of round %d' % r
        assert oracle.canonical_labels(st['fc']) == z['round_f_label'][r].tolist(), 'factor partition of round %d' % r
        np.testing.assert_allclose(flat.var_value[rvc][ev], z['round_value'][r][ev], rtol=1e-15)
        np.testing.assert_allclose(np.array([np.var([rvs[m].value for m in np.flatnonzero(rvc == rvc[i])]) for i in np.flatnonzero(ev)]),
                                   z['round_variance'][r][ev], rtol=1e-13, atol=1e-300)
        np.testing.assert_allclose(st['obs_var'][rvc][ev], z['round_variance'][r][ev], rtol=1e-13, atol=1e-300)
        tracked = np.array([int(rvc[i]) in st['tracked'] for i in range(len(rvs))], dtype=np.int8)
        # (membership matters only for clusters that can still split: more than one member with different values)
        live = ev & (z['round_variance'][r] > 0)
        assert (tracked[live] == z['round_tracked'][r][live]).all()
        np.testing.assert_allclose(P['eta_c'][cont], z['round_eta_c'][r][cont], rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(P['m_eta_c'][cont], z['round_m_c'][r][cont], rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(P['s_eta_c'][cont], z['round_s_c'][r][cont], rtol=1e-8, atol=1e-14)
        if disc.any():
            D = z['round_tau_d'].shape[-1]
            np.testing.assert_allclose(P['tau_d'][disc][:, :, :D], np.nan_to_num(z['round_tau_d'][r][disc]), rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(P['w_tau'], z['round_w_tau'][r], rtol=1e-8, atol=1e-12)
        assert st['t'] == int(z['round_t'][r])
        seen.append(r)
    return observer


def kmeans_order_of(meta):
    """replay of the set order the reference's k-means walked each evidence cluster in (recorded by oracle/capture_vi.py;
    fixtures without the record keep ground order: their clusters never hold more than k distinct values)"""
    rec = meta.get('kmeans_orders')
    if not rec:
        return None

    def order(members):
        seen = rec.get(','.join(map(str, sorted(int(m) for m in members))))
        if seen is None:
            return None
        pos = {int(m): i for i, m in enumerate(members)}
        return [pos[i] for i in seen]
    return order


def check_c2fvi_result(z, rvs, res):
    cont = np.array([rv.value is None and rv.domain.continuous for rv in rvs])
    assert oracle.canonical_labels(res['rvc']) == z['final_rv_label'].tolist()
    assert oracle.canonical_labels(res['fc']) == z['final_f_label'].tolist()
    np.testing.assert_allclose(res['fe_log'], z['fe_log'], rtol=1e-8)
    np.testing.assert_allclose(res['params']['eta_c'][cont], z['final_eta_c'][cont], rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(res['params']['w_tau'], z['final_w_tau'], rtol=1e-7, atol=1e-10)


@pytest.mark.parametrize('name', C2F_CASES)
def test_c2fvi_oracle_matches_reference(golden_dir, name):
    """the coarse-to-fine schedule of lhvi.c2fvi driven by the CPU oracle (Gaussian observation clusters in every
    expectation, oracle/c/vi_oracle.c) against the reference's C2FVarInference: every round's partition and inherited
    state, the free energy after each of the 30 ADAM updates, the final parameters"""
    from lhvi import c2fvi
    from test_oracle_pbp import OracleRefiner
    z, meta = load_vi(golden_dir, name)
    g, rvs, factors = modelio.load_model(meta['model'], API)
    seen = []
    res = c2fvi.run_c2fvi(g, OracleViEngine(meta['K'], meta['T']), OracleRefiner(g), meta['K'], meta['iterations'], meta['lr'],
                          dict(C2F_OPTS, update_obs_its=meta['update_obs_its'], kmeans_member_order=kmeans_order_of(meta)),
                          init=(z['eta_c0'], z['tau_d0']), observer=c2fvi_round_checker(z, rvs, seen))
    assert seen == list(range(meta['iterations'] // meta['update_obs_its']))
    # the fixture does exercise Gaussian observations (the RKF well data is binary: its first split already leaves exact evidence)
    assert (np.nan_to_num(z['round_variance']) > 0).any() or 'rkf' in name
    check_c2fvi_result(z, rvs, res)
