"""GPU parity: variational step (VarInference / LiftedVarInference) through the C ABI vs the golden vectors captured
from the reference (gradients, free energy, ADAM trajectory)."""
import numpy as np
import pytest

import modelio
from test_oracle_golden import API
from test_oracle_vi import VI_CASES, LVI_CASES, load_vi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def api():
    from lhvi import _abi
    _abi.require_gpu()
    return _abi


@pytest.mark.parametrize('name', VI_CASES + LVI_CASES)
def test_vi_matches_reference_golden(api, golden_dir, name):
    from lhvi.vi import VarInference, LiftedVarInference
    from oracle import oracle
    z, meta = load_vi(golden_dir, name)
    g, rvs, factors = modelio.load_model(meta['model'], API)
    vi = (LiftedVarInference if meta['lifted'] else VarInference)(g, meta['K'], meta['T'])
    vi._setup(vi._graph_like())
    flat = vi.flat
    if meta['lifted']:
        rv_color, f_color = vi.g.colors()
        assert oracle.canonical_labels(rv_color) == z['rv_label'].tolist()
        assert oracle.canonical_labels(f_color) == z['f_label'].tolist()
        gather = np.array([flat.var_index[rv.cluster] for rv in rvs])
    else:
        gather = np.arange(len(rvs))

    def scatter(key):
        out = np.full((flat.V,) + z[key].shape[1:], np.nan)
        out[gather] = z[key]
        return out

    vi._upload_params(z['w_tau0'], scatter('eta_c0'), scatter('tau_d0'))
    # the reference-style parameter views: logits of the discrete variables, keyed by rv
    tau = vi.eta_tau
    assert len(tau) == int(np.count_nonzero(vi._disc))
    for rv_, t in tau.items():
        v = flat.var_index[rv_]
        assert t.shape == (vi.K, int(flat.var_nstates[v]))
        np.testing.assert_array_equal(t, np.nan_to_num(scatter('tau_d0')[v], nan=0.0)[:, :t.shape[1]])
    # fp64 tolerance: same formulas, different summation order (per-edge partials gathered per variable)
    assert vi.free_energy() == pytest.approx(float(z['fe0']), rel=1e-10)
    np.testing.assert_allclose(vi.gradient_w_tau(), z['g_w0'], rtol=1e-8, atol=1e-9)
    cont = np.array([rv.value is None and rv.domain.continuous for rv in rvs])
    disc = np.array([rv.value is None and not rv.domain.continuous for rv in rvs])
    g_c = vi._dev['g_c'].cpu().numpy()
    np.testing.assert_allclose(g_c[gather][cont], z['g_c0'][cont], rtol=1e-8, atol=1e-9)
    if disc.any():
        g_d = vi._dev['g_d'].cpu().numpy()
        D = z['g_d0'].shape[2]
        np.testing.assert_allclose(g_d[gather][disc][:, :, :D], np.nan_to_num(z['g_d0'][disc], nan=0.0), rtol=1e-8, atol=1e-9)
    i0 = int(np.flatnonzero(cont)[0])
    np.testing.assert_allclose(vi.gradient_mu_var(rvs[i0].cluster if meta['lifted'] else rvs[i0]), z['g_c0'][i0], rtol=1e-8, atol=1e-9)

    # the ADAM trajectory: run() re-draws the initial parameters from NumPy's stream.  Ground graphs iterate the
    # same ordered rvs as the reference did, so a re-seeded run must land on the reference's numbers; lifted
    # graphs iterate clusters (set order in the reference), so inject the recorded parameters instead.
    if not meta['lifted']:
        np.random.seed(meta['seed'])
        vi.run(meta['iterations'], lr=meta['lr'])
    else:
        vi.run(0, lr=meta['lr'])
        vi._upload_params(z['w_tau0'], scatter('eta_c0'), scatter('tau_d0'))
        vi.ADAM_update(meta['iterations'])
    np.testing.assert_allclose([x[1] for x in vi.time_log], z['fe_log'], rtol=1e-8)
    np.testing.assert_allclose(vi.w, z['w_final'], rtol=1e-8)
    np.testing.assert_allclose(vi._dev['eta_c'].cpu().numpy()[gather][cont], z['eta_c_final'][cont], rtol=1e-8, atol=1e-10)
    # queries
    for i in list(np.flatnonzero(cont))[:4] + list(np.flatnonzero(disc))[:3]:
        rv = rvs[i]
        x = 0.5 if rv.domain.continuous else rv.domain.values[0]
        assert vi.belief(x, rv) == pytest.approx(z['belief_mid'][i], rel=1e-7, abs=1e-12)
        assert vi.map(rv) == pytest.approx(z['map'][i], rel=1e-5, abs=1e-5)


def test_vi_sane_mode_runs_and_decreases_free_energy(api, golden_dir):
    """reference_quirks=False: gradients of the category parameters use proper quadrature for the neighbours"""
    from lhvi.vi import VarInference
    z, meta = load_vi(golden_dir, 'hybrid_k2')
    g, rvs, factors = modelio.load_model(meta['model'], API)
    vi = VarInference(g, 2, 3)
    vi.reference_quirks = False
    np.random.seed(3)
    vi.run(15, lr=0.1)
    fe = [x[1] for x in vi.time_log]
    assert np.isfinite(fe).all() and fe[-1] < fe[0]


def test_vi_fast_paths_match_oracle_on_a_template_graph(api):
    """ground RGM template (pairwise Gaussian factors, template variables with 80 / 61 incident factors, 10 % evidence):
    the pairwise-continuous factor kernel, the wavefront-per-hub gather and the two-stage reduction of g_w / free
    energy against the C oracle's straightforward loops"""
    from lhvi import synth
    from lhvi.vi import VarInference
    from oracle import oracle
    flat, sym, rv0, f0 = synth.rgm_flat(C=80, B=60, n_values=0, evidence_ratio=0.1, seed=4)
    assert np.diff(flat.var_ptr).max() > 64
    K, T = 2, 3
    vi = VarInference(None, K, T)
    vi._setup_flat(flat)
    rng = np.random.default_rng(0)
    eta_c = np.ones((flat.V, K, 2))
    eta_c[:, :, 0] = rng.uniform(-1.5, 1.5, (flat.V, K))
    eta_c[:, :, 1] = rng.uniform(0.5, 3.0, (flat.V, K))
    w_tau = rng.normal(size=K)
    tau_d = np.zeros((flat.V, K, 1))
    vi._upload_params(w_tau, eta_c, tau_d)
    o = oracle.ViOracle(flat, K, T)
    o.set_params(w_tau, eta_c, tau_d)
    g_w, g_c, g_d, fe = o.grad()
    assert vi.free_energy() == pytest.approx(fe, rel=1e-10)
    np.testing.assert_allclose(vi.gradient_w_tau(), g_w, rtol=1e-8, atol=1e-8)
    cont = flat.var_hidden & flat.var_cont
    np.testing.assert_allclose(vi._dev['g_c'].cpu().numpy()[cont], g_c[cont], rtol=1e-8, atol=1e-8)


@pytest.mark.parametrize('name', ['kalman_k1', 'hybrid_k2'])
def test_device_log_likelihood_matches_host(api, golden_dir, name):
    """lhvi_log_likelihood (utils.log_likelihood on flat arrays) against the host loop over factor objects, on a Gaussian
    model and on the hybrid model (MLN formulas, tables, discrete states), including the -inf convention"""
    from lhvi import utils
    from lhvi.flat import flatten
    z, meta = load_vi(golden_dir, name)
    g, rvs, factors = modelio.load_model(meta['model'], API)
    flat = flatten(g, require_device_potentials=True)
    dg = api.DeviceGraph(flat)
    rng = np.random.default_rng(1)
    for trial in range(3):
        assignment = {}
        for rv in flat.rvs:
            if rv.value is not None:
                assignment[rv] = rv.value
            elif rv.domain.continuous:
                assignment[rv] = float(rng.uniform(-2, 2))
            else:
                assignment[rv] = rv.domain.values[int(rng.integers(len(rv.domain.values)))]
        x = np.array([assignment[rv] for rv in flat.rvs], dtype=np.float64)
        # (the reference hands `potential.get` a list, which a table potential would treat as a fancy index: use tuples)
        vals = [float(f.potential.get(tuple(assignment[rv] for rv in f.nb))) for f in flat.factors]
        want = -np.inf if min(vals) == 0 else -float(np.sum(np.log(vals)))
        got = utils.log_likelihood_flat(dg, x)
        if np.isinf(want):
            assert got == want
        else:
            assert got == pytest.approx(want, rel=1e-12, abs=1e-10)


def test_device_evaluators_match_reference_values(api, golden_dir):
    """lhvi_log_likelihood on the device-resident flat graph against utils.log_likelihood values computed by the reference
    (tests/golden/utils.json: chain, Kalman, paper-popularity HMLN, RGM/0, and a vanishing factor -> -inf), and the batched
    utils.kl_tables against the reference's kl_continuous of the same two densities"""
    import json
    import os
    from lhvi import utils
    from lhvi.flat import flatten
    rec = json.load(open(os.path.join(golden_dir, 'utils.json')))
    for case in rec['log_likelihood']:
        g, rvs, factors = modelio.load_model(case['model'], API)
        flat = flatten(g, require_device_potentials=True)
        got = utils.log_likelihood_flat(api.DeviceGraph(flat), np.array(case['x'], dtype=np.float64))
        assert got == case['value'] if np.isinf(case['value']) else got == pytest.approx(case['value'], rel=1e-12)
    m = 20001
    cs = rec['kl']
    x = np.stack([np.linspace(c['lo'], c['hi'], m) for c in cs])
    pdf = lambda x, mu, s: np.exp(-((x - mu) / s) ** 2 * 0.5) / (2.506628274631 * s)
    p = np.stack([pdf(x[i], c['mu1'], c['s1']) for i, c in enumerate(cs)])
    q = np.stack([pdf(x[i], c['mu2'], c['s2']) for i, c in enumerate(cs)])
    got = utils.kl_tables(p, q, np.array([c['lo'] for c in cs], dtype=float), np.array([c['hi'] for c in cs], dtype=float)).cpu().numpy()
    for i, c in enumerate(cs):
        assert got[i] == pytest.approx(c['kl_continuous'], rel=1e-6, abs=1e-9)


@pytest.mark.parametrize('name', ['c2f_rgm_k2', 'c2f_hmln_k2', 'c2f_robot_k2', 'c2f_rkf_tree_k1', 'c2f_rkf_cycle_k1'])
def test_c2f_var_inference_matches_reference(api, golden_dir, name):
    """C2FVarInference on the device (coarse-to-fine lifting with Gaussian observation clusters, csrc/vi.hip through
    lhvi_vi_t.obs_var) against the reference: every round's partition / inherited parameters / ADAM moments, the free
    energy after each of the 30 updates, final parameters, beliefs and MAPs of the ground variables"""
    from lhvi.c2fvi import VarInference as C2FVI
    from test_oracle_vi import c2fvi_round_checker, kmeans_order_of
    from oracle import oracle
    z, meta = load_vi(golden_dir, name)
    g, rvs, factors = modelio.load_model(meta['model'], API)
    vi = C2FVI(g, meta['K'], meta['T'])
    vi.update_obs_its = meta['update_obs_its']
    vi.kmeans_member_order = kmeans_order_of(meta)      # the set order the reference's k-means happened to walk (robot fixture)
    vi.init = (z['eta_c0'], z['tau_d0'])
    seen = []
    vi.observer = c2fvi_round_checker(z, rvs, seen)
    vi.run(meta['iterations'], lr=meta['lr'])
    assert seen == list(range(meta['iterations'] // meta['update_obs_its']))
    res = vi._result
    assert oracle.canonical_labels(res['rvc']) == z['final_rv_label'].tolist()
    np.testing.assert_allclose([fe for _, fe in vi.time_log], z['fe_log'], rtol=1e-8)
    cont = np.array([rv.value is None and rv.domain.continuous for rv in rvs])
    np.testing.assert_allclose(res['params']['eta_c'][cont], z['final_eta_c'][cont], rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(vi.w, z['w_final'], rtol=1e-8)
    assert vi.free_energy() == pytest.approx(float(z['fe_final']), rel=1e-8)
    hidden = [i for i, rv in enumerate(rvs) if rv.value is None]
    for i, rv in enumerate(rvs):
        if rv.value is None:
            if len(hidden) > 60 and i not in hidden[::4]:      # (the robot model has 195 hidden atoms: every fourth)
                continue
            x = 0.5 if rv.domain.continuous else rv.domain.values[0]
            assert vi.belief(x, rv) == pytest.approx(z['belief_mid'][i], rel=1e-7, abs=1e-300)
            assert vi.map(rv) == pytest.approx(z['map'][i], abs=1e-4)
        elif i % 16 == 0 or len(rvs) < 200:
            assert vi.map(rv) == rv.value


@pytest.mark.parametrize('name', ['c2f_rgm_k2_loglik', 'c2f_hmln_k2_loglik'])
def test_c2f_var_inference_logs_the_map_likelihood(api, golden_dir, name):
    """``C2FVarInference.run(log_fe=False)`` (C2FVI:393-404; the quantity of the reference's published HMLN logs): after every
    update ``log_likelihood`` of the ground graph at the MAP of every ground variable, through the objects and on arrays"""
    from lhvi import c2fvi
    from lhvi.flat import flatten
    from test_oracle_vi import kmeans_order_of
    z, meta = load_vi(golden_dir, name)
    g, rvs, factors = modelio.load_model(meta['model'], API)
    vi = c2fvi.VarInference(g, meta['K'], meta['T'])
    vi.update_obs_its = meta['update_obs_its']
    vi.kmeans_member_order = kmeans_order_of(meta)
    vi.init = (z['eta_c0'], z['tau_d0'])
    vi.run(meta['iterations'], lr=meta['lr'], log_fe=False)
    got = [fe for _, fe in vi.time_log]
    assert len(got) == meta['iterations']
    np.testing.assert_allclose(got, z['fe_log'], rtol=1e-6, atol=1e-6)
    vi.is_log, vi.log_fe = True, False
    res = c2fvi.run_c2fvi_flat(flatten(g, require_device_potentials=True), c2fvi._DeviceEngine(vi), meta['K'], meta['iterations'],
                               meta['lr'], vi._options(), init=(z['eta_c0'], z['tau_d0']))
    np.testing.assert_allclose(res['fe_log'], z['fe_log'], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('name', ['hybrid_k2', 'lifted_hybrid_k2', 'lifted_rgm_small_k2', 'kalman_k3', 'lifted_robot_k2'])
def test_fused_adam_loop_equals_the_per_array_calls(api, golden_dir, name):
    """lhvi_vi_adam_run (one gradient pass + one update launch per iteration, enqueued by a single call) against the loop of
    lhvi_vi_grad / masked lhvi_adam_step x 3 / lhvi_softmax_rows: parameters, ADAM moments and logged free energies bit for bit
    (the hybrid fixtures mix binary and three-state variables, for which the per-array path forms its softmax with torch's exp
    and sum instead of the device's: there the two runs agree to rounding)"""
    from lhvi.vi import LiftedVarInference, VarInference
    z, meta = load_vi(golden_dir, name)
    g, rvs, factors = modelio.load_model(meta['model'], API)
    runs = []
    for fused in (True, False):
        vi = (LiftedVarInference if meta['lifted'] else VarInference)(g, meta['K'], meta['T'])
        vi.fused_loop = fused
        np.random.seed(5)
        vi.run(7, lr=0.15)
        runs.append(vi)
    a, b = runs
    exact = bool(a._uniform_states or not a._has_disc)
    for key in ('w_tau', 'w', 'eta_c', 'tau_d', 'eta_d', 'm_w_tau', 's_w_tau', 'm_eta_c', 's_eta_c', 'm_tau_d', 's_tau_d'):
        if exact:
            np.testing.assert_array_equal(a._dev[key].cpu().numpy(), b._dev[key].cpu().numpy(), err_msg=key)
        else:
            np.testing.assert_allclose(a._dev[key].cpu().numpy(), b._dev[key].cpu().numpy(), rtol=1e-11, atol=1e-13, err_msg=key)
    fa, fb = [fe for _, fe in a.time_log], [fe for _, fe in b.time_log]
    assert (fa == fb if exact else np.allclose(fa, fb, rtol=1e-12)) and a.t == b.t == 7


@pytest.mark.parametrize('name', ['c2f_rgm_k2', 'c2f_hmln_k2'])
def test_c2fvi_on_arrays_equals_the_object_path(api, golden_dir, name):
    """``run_c2fvi_flat`` (ground FlatGraph in: refinement to the fixed point and re-lifting on the device, clustered_evidence in
    closed form, parameters per cluster) against ``run_c2fvi`` on the objects and through it against the reference's run:
    every round's partition and Gaussian observations, the free energy after every update, the final parameters"""
    from lhvi import c2fvi
    from lhvi.flat import flatten
    from oracle import oracle
    z, meta = load_vi(golden_dir, name)
    g, rvs, factors = modelio.load_model(meta['model'], API)
    vi = c2fvi.VarInference(g, meta['K'], meta['T'])
    opts = dict(vi._options(), update_obs_its=meta['update_obs_its'])
    rounds = []
    res = c2fvi.run_c2fvi_flat(flatten(g, require_device_potentials=True), c2fvi._DeviceEngine(vi), meta['K'], meta['iterations'],
                               meta['lr'], opts, init=(z['eta_c0'], z['tau_d0']), observer=lambda r, st: rounds.append(st))
    assert len(rounds) == meta['iterations'] // meta['update_obs_its']
    ev = np.array([rv.value is not None for rv in rvs])
    for r, st in enumerate(rounds):
        assert oracle.canonical_labels(st['rvc']) == z['round_rv_label'][r].tolist(), 'rv partition of round %d' % r
        assert oracle.canonical_labels(st['fc']) == z['round_f_label'][r].tolist(), 'factor partition of round %d' % r
        np.testing.assert_allclose(st['obs_var'][st['rvc']][ev], z['round_variance'][r][ev], rtol=1e-12, atol=1e-300)
        np.testing.assert_allclose(st['flat'].var_value[st['rvc']][ev], z['round_value'][r][ev], rtol=1e-14)
        live = ev & (z['round_variance'][r] > 0)
        tracked = np.array([int(st['rvc'][i]) in st['tracked'] for i in range(len(rvs))], dtype=np.int8)
        assert (tracked[live] == z['round_tracked'][r][live]).all()
    assert oracle.canonical_labels(res['rvc']) == z['final_rv_label'].tolist()
    np.testing.assert_allclose(res['fe_log'], z['fe_log'], rtol=1e-8)
    cont = np.array([rv.value is None and rv.domain.continuous for rv in rvs])
    np.testing.assert_allclose(res['params']['eta_c'][res['rvc']][cont], z['final_eta_c'][cont], rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(res['params']['w_tau'], z['final_w_tau'], rtol=1e-7, atol=1e-10)


@pytest.mark.parametrize('name', ['hybrid_k2', 'lifted_robot_k2', 'kalman_k3', 'lifted_rgm_small_k2'])
def test_table_kernels_equal_the_thread_per_factor_kernels(api, golden_dir, name):
    """round 4: the factors are split on the host (``vi.factor_lists``) between the pairwise fast path with per-axis pdf tables in
    registers, the group-of-8-lanes kernel with per-axis tables in LDS, and the thread-per-factor kernels; without the lists
    (``factor_lists = False``) every factor goes through the thread-per-factor kernels of rounds 1-3.  Same gradients and free
    energy (the table kernels take log(phi + 1e-100) = log phi directly and hoist reciprocals: rounding-level differences)."""
    from lhvi.vi import LiftedVarInference, VarInference
    z, meta = load_vi(golden_dir, name)
    g, rvs, factors = modelio.load_model(meta['model'], API)
    outs = []
    for lists in (True, False):
        vi = (LiftedVarInference if name.startswith('lifted') else VarInference)(g, meta['K'], meta['T'])
        vi.factor_lists = lists
        np.random.seed(3)
        vi.init_param()
        vi._grad()
        d = vi._dev
        outs.append([d[k].cpu().numpy().copy() for k in ('g_w', 'g_c', 'g_d', 'fe')] + [vi._fac_counts])
    a, b = outs
    assert b[4] is None and a[4] is not None and sum(a[4]) == len(vi.flat.factors)
    if name == 'lifted_robot_k2':
        assert a[4][3] > 0 and a[4][2] > 0      # arity-5 and short formulas on the two group kernels
    for x, y, what in zip(a[:4], b[:4], ('g_w', 'g_c', 'g_d', 'fe')):
        np.testing.assert_allclose(x, y, rtol=1e-9, atol=1e-9, err_msg=what)


@pytest.mark.parametrize('name', ['hybrid_k2', 'lifted_robot_k2', 'paper_popularity', 'paper_popularity_k1_noquirks'])
def test_tiny_grid_kernel_equals_the_group_kernel(api, golden_dir, name):
    """factors with at most 32 grid nodes go to the thread-per-(factor, k) kernel (``vi_factor_tiny_kernel``) when there are
    enough of them to fill the device (forced here); with ``tiny_kernel = False`` the same factors take the 8-lane group kernel.  Same products in the same order, the grid summed in a
    different order: agreement to rounding."""
    from lhvi import synth
    from lhvi.vi import LiftedVarInference, VarInference
    outs = []
    for tiny in ('always', False):
        if name.startswith('paper_popularity'):
            K = 1 if 'k1' in name else 2
            vi = VarInference(None, K, 3)
            vi.tiny_kernel = tiny
            vi.reference_quirks = 'noquirks' not in name
            vi._setup_flat(synth.paper_popularity_flat(40, 5, seed=1, points=20)[0])
        else:
            z, meta = load_vi(golden_dir, name)
            g, rvs, factors = modelio.load_model(meta['model'], API)
            vi = (LiftedVarInference if name.startswith('lifted') else VarInference)(g, meta['K'], meta['T'])
            vi.tiny_kernel = tiny
        np.random.seed(3)
        vi.init_param()
        vi._grad()
        d = vi._dev
        outs.append([d[k].cpu().numpy().copy() for k in ('g_w', 'g_c', 'g_d', 'fe')] + [vi._fac_counts])
    a, b = outs
    assert a[4][1] > 0 and b[4][1] == 0 and b[4][2] + b[4][3] >= a[4][1]
    for x, y, what in zip(a[:4], b[:4], ('g_w', 'g_c', 'g_d', 'fe')):
        np.testing.assert_allclose(x, y, rtol=1e-11, atol=1e-11, err_msg=what)


def _random_hybrid_graph(rng, n_rv=40, n_f=90, one_discrete_domain=False):
    """a random hybrid factor graph through the object API: binary / ternary / four-state / continuous variables (30 % observed),
    factors of arity 1-3 -- dict-keyed tables over discrete scopes, MLN formulas over mixed
    scopes, Gaussian / linear-Gaussian / XY pairs over continuous ones"""
    from lhvi import mln as M, potentials as P
    from lhvi.graph import Domain, F, Graph, RV
    d2, d3, d4 = Domain((0, 1)), Domain((0, 1, 2)), Domain((1, 2, 3, 5))
    dc = Domain((-10, 10), continuous=True, integral_points=np.linspace(-10, 10, 9))
    rvs = []
    for i in range(n_rv):
        # (under the reference's quirk 10 a discrete neighbour is evaluated at the TARGET's state values: only graphs whose discrete
        # variables share a domain are valid inputs then -- anything else indexes eta out of range, in the reference as here)
        dom = (d4, d4, d4, dc, dc)[rng.integers(5)] if one_discrete_domain else (d2, d3, d4, dc, dc)[rng.integers(5)]
        val = None
        if rng.random() < 0.3:
            val = float(np.round(rng.uniform(-3, 3), 2)) if dom.continuous else dom.values[rng.integers(len(dom.values))]
        rvs.append(RV(dom, val))
    formulas = {1: [lambda x: M.eq_op(x[0], 1), lambda x: x[0] * 0.5 - 1],
                2: [lambda x: M.eq_op(x[0], x[1]), lambda x: M.or_op(M.neg_op(x[0] > 0), x[1] > 1), lambda x: -abs(x[0] - 2 * x[1])],
                3: [lambda x: x[0] * M.eq_op(x[1], x[2]), lambda x: M.imp_op(x[0] >= 1, M.eq_op(x[1], x[2]) + 1) / 2,
                    lambda x: (x[0] + x[1] - x[2]) ** 2 * -0.1]}
    pots = {(a, i): M.MLNPotential(f, w=float(np.round(rng.uniform(-1, 2), 2))) for a in formulas for i, f in enumerate(formulas[a])}
    pair_pots = [P.GaussianPotential([0.0, 0.0], [[2.0, 0.6], [0.6, 1.5]]), P.LinearGaussianPotential(0.8, 1.3), P.XYPotential(0.7, 2.0)]
    tables = {}
    factors = []
    for i in range(n_f):
        a = int(rng.integers(1, 4))
        scope = [rvs[j] for j in rng.choice(n_rv, a, replace=False)]
        # (a table over one domain only: under the reference's quirk 10 a neighbour is evaluated at the TARGET's state values)
        if all(not r.domain.continuous and r.domain is scope[0].domain for r in scope) and rng.random() < 0.7:
            key = tuple(id(r.domain) for r in scope)
            if key not in tables:
                import itertools
                tables[key] = P.TablePotential({vals: float(rng.uniform(0.2, 2.0)) for vals in itertools.product(*[r.domain.values for r in scope])})
            pot = tables[key]
        elif a == 2 and all(r.domain.continuous for r in scope) and rng.random() < 0.7:
            pot = pair_pots[rng.integers(len(pair_pots))]
        else:
            pot = pots[(a, int(rng.integers(len(formulas[a]))))]
        factors.append(F(pot, nb=scope))
    g = Graph()
    g.rvs, g.factors = [r for r in rvs if any(r in f.nb for f in factors)], factors
    g.init_nb()
    return g


@pytest.mark.parametrize('seed,K,T,quirks', [(0, 2, 3, True), (1, 1, 3, True), (2, 2, 2, False), (3, 2, 3, False), (4, 2, 5, True)])
def test_random_hybrid_graphs_through_every_factor_kernel(api, seed, K, T, quirks):
    assert _random_hybrid_check(seed, K, T, quirks, require_every_kernel=True) is None


def _random_hybrid_check(seed, K, T, quirks, require_every_kernel):
    """random hybrid graphs (observed discrete states, three- and four-state variables, formulas
    with comparisons / abs / divisions / squares, dict-keyed tables, Gaussian observations on some observed continuous variables):
    gradient and free energy through (a) the tiny-grid kernel, (b) the group kernel, (c) the thread-per-factor kernels of rounds
    1-3, against the C oracle and against each other"""
    from lhvi import c2fvi
    from lhvi.flat import flatten
    from oracle import oracle
    rng = np.random.default_rng(100 + seed)
    g = _random_hybrid_graph(rng, one_discrete_domain=quirks)
    flat = flatten(g, require_device_potentials=True)
    obs_c = np.flatnonzero(~flat.var_hidden & flat.var_cont)
    obs_var = np.zeros(flat.V)
    obs_var[obs_c[::2]] = rng.uniform(0.3, 1.5, obs_c[::2].size)       # every other observed continuous variable: a Gaussian observation
    owner = c2fvi.VarInference.__new__(c2fvi.VarInference)
    owner._init_common(K, T)
    owner.reference_quirks = quirks
    w_tau = rng.normal(size=K)
    eta_c = np.ones((flat.V, K, 2))
    eta_c[:, :, 0] = rng.uniform(-1.5, 1.5, (flat.V, K))
    eta_c[:, :, 1] = rng.uniform(0.5, 3.0, (flat.V, K))
    o = oracle.ViOracle(flat, K, T, quirks=1 if quirks else 0, obs_var=obs_var)
    tau_d = rng.uniform(0, 2, (flat.V, K, o.Dmax))
    o.set_params(w_tau, eta_c, tau_d)
    want = o.grad()
    if not np.isfinite(want[3]):
        # exp(w * formula) overflowed somewhere on a grid: the reference raises OverflowError there (MLNPotential.py:37, e ** x), the
        # oracle returns inf, the device kernels take log(exp(v) + 1e-100) = v without forming exp(v) -- not a valid input
        return 'overflow'
    outs = {}
    for label, lists, tiny in (('tiny', True, 'always'), ('group', True, False), ('thread per factor', False, False)):
        Stage = type('Stage', (c2fvi._DeviceStage,), dict(factor_lists=lists, tiny_kernel=tiny))
        st = Stage(owner, flat, obs_var)
        st._upload_params(w_tau, eta_c, tau_d)
        st._grad()
        d = st._dev
        outs[label] = [d[k].cpu().numpy().copy() for k in ('g_w', 'g_c', 'g_d', 'fe')]
        if label == 'tiny' and require_every_kernel:
            assert st._fac_counts[1] > 20 and st._fac_counts[0] > 0, st._fac_counts
        if label == 'group' and require_every_kernel:
            assert st._fac_counts[1] == 0 and st._fac_counts[2] > 20, st._fac_counts
    cont, disc = flat.var_hidden & flat.var_cont, flat.var_hidden & ~flat.var_cont
    for label, (g_w, g_c, g_d, fe) in outs.items():
        np.testing.assert_allclose(fe[0], want[3], rtol=1e-9, err_msg=label)
        np.testing.assert_allclose(g_w, want[0], rtol=1e-8, atol=1e-8, err_msg=label)
        np.testing.assert_allclose(g_c[cont], want[1][cont], rtol=1e-8, atol=1e-8, err_msg=label)
        np.testing.assert_allclose(g_d[disc], want[2][disc], rtol=1e-8, atol=1e-8, err_msg=label)
    for a, b in (('tiny', 'group'), ('group', 'thread per factor')):
        for x, y, what in zip(outs[a], outs[b], ('g_w', 'g_c', 'g_d', 'fe')):
            np.testing.assert_allclose(x, y, rtol=1e-9, atol=1e-9, err_msg='%s vs %s: %s' % (a, b, what))
