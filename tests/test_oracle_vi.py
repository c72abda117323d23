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
LVI_CASES = ['lifted_rgm_small_k2', 'lifted_kalman_full_k2', 'lifted_hybrid_k2']


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
