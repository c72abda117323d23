"""The reference's third demo family: relational Kalman filters on well data (Demo/RKF/LRKFDemoTree.py, LRKFDemoCycle.py).
Fixtures (oracle/capture_rkf.py): the arrays the demos derive from their .mat inputs and what the reference computes on the graph
its own KalmanFilter builds from them -- GaBP(20) marginals, the GaLBP partition (tree: 117 rv / 225 factor clusters, BASELINE.md
section 2) and marginals; C2FVarInference(g, 1, 3) for three rounds is in vi_c2f_rkf_{tree,cycle}_k1.npz and runs through the
C2F case lists of test_oracle_vi.py / test_gpu_vi.py."""
import json
import os

import numpy as np
import pytest

import modelio
from test_oracle_golden import API, _initial
from lhvi import graph as G, kalman, lifting
from lhvi.flat import flatten
from oracle import oracle

CASES = ['tree', 'cycle']
T_STEPS = 20


def _load(golden_dir, which):
    z = np.load(os.path.join(golden_dir, 'rkf.npz'))
    zc = np.load(os.path.join(golden_dir, 'vi_c2f_rkf_%s_k1.npz' % which))
    model = json.loads(str(zc['meta']))['model']
    return {k[len(which) + 1:]: z[k] for k in z.files if k.startswith(which + '_')}, model


def _filter(which, rec):
    """the demo's KalmanFilter arguments (LRKFDemoTree.py:50-54, LRKFDemoCycle.py:56-60), parameter set 0"""
    data, param = rec['data'], rec['param']
    n = data.shape[0]
    dom = G.Domain((-4, 4), continuous=True, integral_points=np.linspace(-4, 4, 30))
    A = np.eye(n) * param[2, 0] + (0.01 if which == 'cycle' else 0.0)
    return kalman.KalmanFilter(dom, A, param[0, 0], np.eye(n), param[1, 0]), data


@pytest.mark.parametrize('which', CASES)
def test_rkf_builders_give_the_reference_graph(golden_dir, which):
    """``KalmanFilter.grounded_graph`` and ``grounded_flat`` on the demo's inputs: the graph the reference built (same variable
    order and evidence, same factors in the same order with the same potentials)"""
    rec, model = _load(golden_dir, which)
    kf, data = _filter(which, rec)
    g, table = kf.grounded_graph(T_STEPS, data)
    flat, state_id = kf.grounded_flat(T_STEPS, data)
    V, F, E = (int(x) for x in rec['sizes'])
    assert (len(g.rvs), len(g.factors)) == (V, F) == (flat.V, flat.F) and flat.E == E
    if which == 'tree':
        assert (V, F, E) == (2964, 5700, 8588)                 # SURVEY.md section 4 / BASELINE.md section 2
    ref_vals = np.array([np.nan if v is None else v for _, v in model['rvs']])
    np.testing.assert_array_equal(flat.var_value, ref_vals)
    rg, rrvs, rfactors = modelio.load_model(model, API)
    ref_flat = flatten(rg)
    np.testing.assert_array_equal(flat.fac_ptr, ref_flat.fac_ptr)
    np.testing.assert_array_equal(flat.edge_var, ref_flat.edge_var)
    for a, b in ((flat, ref_flat),):
        ka, kb = a.pot_kind[a.fac_pot], b.pot_kind[b.fac_pot]
        np.testing.assert_array_equal(ka, kb)
        pa = np.stack([a.pot_param[a.pot_off[a.fac_pot]], a.pot_param[a.pot_off[a.fac_pot] + 1]], axis=1)
        pb = np.stack([b.pot_param[b.pot_off[b.fac_pot]], b.pot_param[b.pot_off[b.fac_pot] + 1]], axis=1)
        np.testing.assert_array_equal(pa, pb)
    np.testing.assert_array_equal(state_id[T_STEPS - 1], rec['last_step'])
    np.testing.assert_array_equal(np.array([g.rvs.index(rv) for rv in table[T_STEPS - 1]]), rec['last_step'])


@pytest.mark.parametrize('which', CASES)
def test_rkf_oracle_matches_reference(golden_dir, which):
    """CPU oracle on the reference's graph: GaBP(20) marginals of every variable, the GaLBP partition (exact)"""
    rec, model = _load(golden_dir, which)
    g, rvs, factors = modelio.load_model(model, API)
    flat = flatten(g)
    _, _, mv = oracle.gabp_run(flat, 20)
    np.testing.assert_allclose(mv, rec['gabp'], rtol=1e-13, atol=1e-13)
    sym, rv0, f0 = _initial(flat, g)
    rvc, fc = oracle.color_passing(flat, sym, rv0, f0)
    assert oracle.canonical_labels(rvc) == rec['galbp_rv_label'].tolist()
    assert oracle.canonical_labels(fc) == rec['galbp_f_label'].tolist()
    if which == 'tree':
        assert (int(rvc.max()) + 1, int(fc.max()) + 1) == (117, 225)
    # MATLAB's answers for the last time step (another filter: a sanity check only; tree <= 3e-2 per SURVEY.md section 4, cycle 3.5e-2)
    res = rec['res']
    last = rec['last_step']
    assert np.abs(mv[last, 0] - res[:, 0]).max() < (3e-2 if which == 'tree' else 5e-2)


@pytest.mark.gpu
@pytest.mark.parametrize('which', CASES)
def test_rkf_gabp_and_galbp_on_the_device(golden_dir, which):
    """``GaBP`` / ``GaLBP`` through ``KalmanFilter.grounded_flat`` on the demo's inputs against the reference's recorded run:
    marginals of all variables after 20 sweeps, the lifted partition bit for bit, lifted == ground marginals"""
    from lhvi import _abi
    from lhvi.gabp import GaBP
    _abi.require_gpu()
    rec, model = _load(golden_dir, which)
    kf, data = _filter(which, rec)
    flat, state_id = kf.grounded_flat(T_STEPS, data)
    bp = GaBP(flat)
    bp.run(20)
    mv = bp._state['mv'].cpu().numpy()
    hid = flat.var_hidden
    np.testing.assert_allclose(mv[hid], rec['gabp'][hid], rtol=1e-12, atol=1e-12)
    rv0, f0, sym = lifting.initial_colors_flat(flat, True)
    # (build_flat deduplicates potentials by value; the reference's GaLBP groups Linear / X2 / XY potentials by value too)
    rvc, fc = lifting.refine_flat(flat, sym, rv0, f0)
    assert oracle.canonical_labels(rvc) == rec['galbp_rv_label'].tolist()
    assert oracle.canonical_labels(fc) == rec['galbp_f_label'].tolist()
    lflat = lifting.lift_flat(flat, rvc, fc)
    lbp = GaBP(lflat)
    lbp.run(20)
    lmv = lbp._state['mv'].cpu().numpy()
    np.testing.assert_allclose(lmv[rvc][hid, 0], rec['galbp_map'][hid], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(lmv[rvc][hid], mv[hid], rtol=1e-11, atol=1e-12)
