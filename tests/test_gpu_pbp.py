"""GPU parity: particle BP (EPBP / HybridLBP) through the C ABI vs the golden vectors captured from the
reference, and vs the C oracle on a larger random hybrid MRF."""
import numpy as np
import pytest

import modelio
from test_oracle_golden import API
from test_oracle_pbp import (C2F_CASES, EPBP_CASES, HLBP_CASES, c2f_table_observer, check_draw_tables, lifted_edge_of,
                             load_npz)

pytestmark = pytest.mark.gpu

# Tolerance of the log-message tables (fp64).  The kernels fold phi and the incoming log message into one exp and
# use shuffle-tree reductions, so they differ from CPython's left-to-right sums by rounding only.
RTOL, ATOL = 1e-9, 1e-8


@pytest.fixture(scope='module')
def api():
    from lhvi import _abi
    _abi.require_gpu()
    return _abi


def _injector(samples):
    return lambda k, flat, q: samples[k]


def _init(api, bp):
    api.check(api.lib().lhvi_pbp_init(bp.dg.g, bp._struct(), api.ptr(bp.eta), api.ptr(bp.q_dev), api.ptr(bp.f2v),
                                      api.ptr(bp.v2f), api.stream_ptr()))
    bp._generate_sample()


def _check_per_variable_map(bp, rv, want, log_belief_at):
    """``bp.map(rv)`` three ways: answered from the one batched pass made at the first call (default: every variable runs the
    reference's fminbound iteration in one launch, ``lhvi_pbp_map_brent`` -- the reference's value to 1e-4, the tolerance of a run
    whose objective agrees to ~1e-9 and stops at xtol = 1e-5); with ``exact_queries`` by scipy's fminbound on the device function,
    one call per evaluation; and with ``map_mode = 'global'`` from the scan + bracket refinement (the reference's value, or a point
    with a belief at least as large when the belief is multi-modal)"""
    got = bp.map(rv)
    assert got == pytest.approx(want, abs=1e-4)
    bp.exact_queries = True
    try:
        exact = bp.map(rv)
        assert exact == pytest.approx(want, abs=1e-4)
        assert got == pytest.approx(exact, abs=2e-6)          # the same iterates (the two sums differ in rounding at most)
    finally:
        bp.exact_queries = False
    bp.map_mode = 'global'
    try:
        glob = bp.map(rv)
        assert glob == pytest.approx(want, abs=2e-4) or log_belief_at(glob) >= log_belief_at(want) - 1e-9
    finally:
        bp.map_mode = 'fminbound'


@pytest.mark.parametrize('name', EPBP_CASES)
def test_epbp_matches_reference_golden(api, golden_dir, name):
    from lhvi.pbp import EPBP
    z, meta = load_npz(golden_dir, name)
    g, rvs, factors = modelio.load_model(meta['model'], API)
    n, its = meta['n'], meta['iterations']
    K = z['sample'].shape[0]
    bp = EPBP(g, n=n, proposal_approximation=meta['approx'], sampler=_injector([z['sample'][k] for k in range(K)]))
    # drive the sweeps one by one to compare every iteration's tables
    bp._setup(g)
    _init(api, bp)
    flat = bp.flat
    hid_e = np.flatnonzero(flat.var_hidden[flat.edge_var])
    npe = bp.np_host[flat.edge_var]
    cont = flat.var_hidden & flat.var_cont
    for i in range(its):
        if i > 0:    # f2v of iteration i-1, tabulated on sample i and the grid
            got, want = bp.f2v.cpu().numpy(), z['f2v'][i]
            for e in hid_e:
                np.testing.assert_allclose(got[e, :npe[e]], want[e, :npe[e]], rtol=RTOL, atol=ATOL)
                v = flat.edge_var[e]
                if flat.var_cont[v]:
                    T = flat.var_nstates[v]
                    np.testing.assert_allclose(got[e, n:n + T], want[e, n:n + T], rtol=RTOL, atol=ATOL)
        bp.sweep(last=(i == its - 1))
        got, want = bp.v2f.cpu().numpy(), z['v2f'][i]
        for e in hid_e:
            np.testing.assert_allclose(got[e, :npe[e]], want[e, :npe[e]], rtol=RTOL, atol=ATOL, err_msg='v2f it %d' % i)
        if i < its - 1:
            np.testing.assert_allclose(bp.q_dev.cpu().numpy()[cont], z['q'][i][cont], rtol=1e-9, atol=1e-12)
            ce = cont[flat.edge_var]
            np.testing.assert_allclose(bp.eta.cpu().numpy()[ce], z['eta'][i][ce], rtol=1e-9, atol=1e-12)
    # queries: belief_rv at recorded points, MAP, normalised belief
    hid = [i for i, rv in enumerate(rvs) if rv.value is None]
    got = bp.belief_rv_batch([rvs[i] for i in hid], z['query_x'][hid])
    np.testing.assert_allclose(got, z['query_logb'][hid], rtol=1e-9, atol=1e-7)
    for i in hid[:6]:
        # fminbound stops at xtol=1e-5 and follows the same iterates while the objective agrees to ~1e-9
        _check_per_variable_map(bp, rvs[i], z['map'][i], lambda x, i=i: float(bp._belief_rv_points(i, [x])[0]))
    assert 'map' in bp._batched                      # ... and the loop over the variables cost ONE batched pass
    for i, x0, want in z['belief']:
        # "marginals within 1e-5 of the CPU reference" (BASELINE.json north_star)
        if np.isnan(want):       # the reference's own normaliser overflowed here (e ** log-belief, EPBP:342): so does the restated one
            with pytest.raises(OverflowError):
                bp.belief(x0, rvs[int(i)])
            continue
        assert bp.belief(x0, rvs[int(i)]) == pytest.approx(want, rel=1e-5, abs=1e-7)
    assert 'quad' in bp._batched                     # the normalisers of ALL variables came from one launch (lhvi_pbp_quad) ...
    from scipy.integrate import quad
    zs, status = bp.quad_all()
    for i in [i for i in hid if flat.var_cont[i]][:8]:     # ... and are scipy.integrate.quad's of the same device function to 1e-6
        if status[i] == 3:
            continue
        lo_, hi_ = rvs[i].domain.values[0] - 20, rvs[i].domain.values[1] + 20
        want_z = quad(lambda val: np.e ** float(bp._belief_rv_points(i, [val])[0]), lo_, hi_)[0]
        assert status[i] == 0 and zs[i] == pytest.approx(want_z, rel=1e-6)
    bp.exact_queries, bp.cache = True, dict()        # one scipy quad per variable on the device function, as before
    try:
        for i, x0, want in z['belief'][:4]:
            if not np.isnan(want):
                assert bp.belief(x0, rvs[int(i)]) == pytest.approx(want, rel=1e-5, abs=1e-7)
    finally:
        bp.exact_queries, bp.cache = False, dict()
    # batched queries: every variable in one f2v launch -- the recorded log-beliefs again, and the reference's MAPs
    k = z['query_x'].shape[1]
    xq = bp.particles.cpu().numpy().copy()                 # discrete rows keep their states
    for i in hid:
        if flat.var_cont[i]:
            xq[i] = np.resize(z['query_x'][i], n)
    allb = bp.belief_rv_all(xq).cpu().numpy()
    chid = [i for i in hid if flat.var_cont[i]]
    np.testing.assert_allclose(allb[chid][:, :min(k, n)], z['query_logb'][chid][:, :min(k, n)], rtol=1e-9, atol=1e-7)
    mp, mval = bp.map_all(steps=6 if n >= 32 else 9)
    for i in hid:
        want = z['map'][i]
        if flat.var_cont[i]:
            # same mode as fminbound unless the belief is multi-modal: accept a better optimum, never a worse one
            ref_val = float(bp._belief_rv_points(i, [want])[0])
            assert mp[i] == pytest.approx(want, abs=2e-4) or mval[i] >= ref_val - 1e-9
        else:
            assert mp[i] == want
    # ... and the reference's own answer for EVERY variable from one launch of the batched fminbound
    fm, fval, nfev = bp.map_fminbound_all()
    for i in hid:
        assert fm[i] == (pytest.approx(z['map'][i], abs=1e-4) if flat.var_cont[i] else z['map'][i])
    assert 0 < nfev[hid].max() < 100
    # interval probabilities (5-point over 20-point trapezoid, EPBP:356-375) against the reference's recorded values:
    # the per-variable query and the batched one
    pa, pb = np.zeros(flat.V), np.ones(flat.V)
    for i, a, b, want in z['probability']:
        assert bp.probability(a, b, rvs[int(i)]) == pytest.approx(want, rel=1e-7, abs=1e-300)
        pa[int(i)], pb[int(i)] = a, b
    pall = bp.probability_all(pa, pb).cpu().numpy()
    for i, a, b, want in z['probability']:
        assert pall[int(i)] == pytest.approx(want, rel=1e-7, abs=1e-300)
    lo = flat.dom_lo[flat.var_dom]
    pall = bp.probability_all(lo + 0.5, lo + 2.0).cpu().numpy()
    for i in chid[:8]:
        assert pall[i] == pytest.approx(bp.probability(lo[i] + 0.5, lo[i] + 2.0, rvs[i]), rel=1e-10)
    assert np.isnan(pall[[i for i in range(flat.V) if i not in chid]]).all()
    # discrete rows of belief_all: normalised over the states, like EPBP.belief
    ball = bp.belief_all(np.zeros((flat.V, 2))).cpu().numpy()
    overflowed = {int(i) for i, x0, want in z['belief'] if np.isnan(want)}
    for i in hid:
        if not flat.var_cont[i] and i not in overflowed:
            vals = list(rvs[i].domain.values)
            for k in range(min(2, len(vals))):
                assert ball[i, k] == pytest.approx(bp.belief(vals[k], rvs[i]), rel=1e-10)


@pytest.mark.parametrize('name', HLBP_CASES)
def test_hlbp_matches_reference_golden(api, golden_dir, name):
    from lhvi.pbp import HybridLBP
    from oracle import oracle
    z, meta = load_npz(golden_dir, name)
    g, rvs, factors = modelio.load_model(meta['model'], API)
    samples = z['samples']

    draws = []

    def inject(k, flat, q):
        if k > 0:     # what the reference's message / eta_message hold at this generate_sample call (HLBP:100-118,182-215)
            check_draw_tables(z, k, rvs, meta['n'], lifted_edge_of(flat), bp.f2v.cpu().numpy(), bp.eta.cpu().numpy())
        draws.append(k)
        rep = np.array([rvs.index(min(c.rvs)) for c in flat.rvs])
        return samples[k][rep]

    bp = HybridLBP(g, n=meta['n'], proposal_approximation=meta['approx'], sampler=inject)
    bp.run(meta['iterations'])
    assert draws == list(range(meta['iterations']))
    rv_color, f_color = bp.g.colors()
    assert oracle.canonical_labels(rv_color) == z['rv_label'].tolist()     # integer partition: exact
    assert oracle.canonical_labels(f_color) == z['f_label'].tolist()
    hid = [i for i, rv in enumerate(rvs) if rv.value is None]
    got = bp.belief_rv_batch([rvs[i] for i in hid], z['query_x'][hid])
    np.testing.assert_allclose(got, z['query_logb'][hid], rtol=1e-9, atol=1e-7)
    Q = bp.q
    for i in hid:
        if rvs[i].domain.continuous:
            np.testing.assert_allclose(Q[rvs[i].cluster], z['final_q'][i], rtol=1e-9)
    overflowed = {i for i in hid if np.isnan(z['belief_mid'][i])}     # the reference's normaliser overflowed (e ** log-belief, HLBP:372)
    for i in hid[:6]:
        _check_per_variable_map(bp, rvs[i], z['map'][i], lambda x, i=i: bp.belief_rv_query(float(x), rvs[i]))
        if i not in overflowed:
            assert bp.belief(z['query_x'][i][2], rvs[i]) == pytest.approx(z['belief_mid'][i], rel=1e-5, abs=1e-7)
    flat = bp.flat
    pa, pb = np.zeros(flat.V), np.ones(flat.V)
    for i, a, b, want in z['probability']:      # HLBP:384-403 against the reference's recorded values
        assert bp.probability(a, b, rvs[int(i)]) == pytest.approx(want, rel=1e-7, abs=1e-300)
        c = flat.var_index[rvs[int(i)].cluster]
        pa[c], pb[c] = a, b
    pall = bp.probability_all(pa, pb).cpu().numpy()
    for i, a, b, want in z['probability']:
        assert pall[flat.var_index[rvs[int(i)].cluster]] == pytest.approx(want, rel=1e-7, abs=1e-300)
    # batched queries on the lifted graph (stable partition): one row per cluster
    mp, mval = bp.map_all(steps=9 if meta['n'] < 32 else 6)
    for i in hid:
        c = flat.var_index[rvs[i].cluster]
        if rvs[i].domain.continuous:
            ref_val = bp.belief_rv_query(float(z["map"][i]), rvs[i])
            assert mp[c] == pytest.approx(z['map'][i], abs=2e-4) or mval[c] >= ref_val - 1e-9
        else:
            assert mp[c] == z['map'][i]
    fm, _, _ = bp.map_fminbound_all()                  # the reference's fminbound iterates, every cluster in one launch
    for i in hid:
        c = flat.var_index[rvs[i].cluster]
        assert fm[c] == (pytest.approx(z['map'][i], abs=1e-4) if rvs[i].domain.continuous else z['map'][i])
    # batched normalised beliefs / interval probabilities: the per-variable queries for every cluster at once
    first = {}
    for i in hid:
        first.setdefault(flat.var_index[rvs[i].cluster], i)
    xq = np.zeros((flat.V, 1))
    for c, i in first.items():
        xq[c, 0] = z['query_x'][i][2]
    ball = bp.belief_all(xq).cpu().numpy()
    lo = np.array([rv.domain.values[0] for rv in flat.rvs], dtype=float)
    pall = bp.probability_all(lo + 0.25, lo + 1.5).cpu().numpy()
    for c, i in first.items():
        if i in overflowed:
            continue
        if rvs[i].domain.continuous:
            assert ball[c, 0] == pytest.approx(z['belief_mid'][i], rel=1e-5, abs=1e-7)
            assert ball[c, 0] == pytest.approx(bp.belief(z['query_x'][i][2], rvs[i]), rel=1e-10)
            assert pall[c] == pytest.approx(bp.probability(lo[c] + 0.25, lo[c] + 1.5, rvs[i]), rel=1e-10)
        else:
            assert ball[c, 0] == pytest.approx(bp.belief(rvs[i].domain.values[0], rvs[i]), rel=1e-10)


def test_epbp_host_sampler_reproduces_reference_stream(api, golden_dir):
    """same np.random.seed, ordered g.rvs -> the host sampler draws the reference's particles bit for bit"""
    from lhvi.pbp import EPBP
    z, meta = load_npz(golden_dir, 'epbp_kalman_simple')
    g, rvs, factors = modelio.load_model(meta['model'], API)
    np.random.seed(meta['seed'])
    bp = EPBP(g, n=meta['n'], proposal_approximation=meta['approx'])
    bp.run(meta['iterations'])
    P = bp.particles.cpu().numpy()
    hid = bp.flat.var_hidden
    np.testing.assert_allclose(P[hid], z['sample'][-1][hid], rtol=1e-9, atol=1e-9)
    got = bp.belief_rv_batch([rv for rv in rvs if rv.value is None], z['query_x'][hid])
    np.testing.assert_allclose(got, z['query_logb'][hid], rtol=1e-8, atol=1e-6)


def test_pbp_sweep_matches_oracle_on_random_hybrid_mrf(api):
    """cfg-4 style random hybrid pairwise MRF (n=64, T=32) at a size the C oracle finishes in seconds;
    device-generated particles are downloaded and fed to the oracle so both see identical inputs"""
    from lhvi import synth
    from lhvi.pbp import EPBP
    from oracle import oracle
    flat = synth.hybrid_mrf_flat(V=3000, deg=4, seed=7)
    bp = EPBP(None, n=64, proposal_approximation='simple', sampler='device', seed=5)
    bp._setup(None, flat=flat)
    _init(api, bp)
    o = oracle.PbpOracle(flat, 64, ep=False, epbp=True, var_threshold=3)
    o.init()
    o.set_particles(bp.particles.cpu().numpy())
    assert (o.uniq == bp.uniq.cpu().numpy()).all()
    hid_e = flat.var_hidden[flat.edge_var]
    for i in range(3):
        bp.sweep(last=False)
        o.step_v2f()
        o.step_proposal()
        o.set_particles(bp.particles.cpu().numpy())
        o.step_f2v()
        np.testing.assert_allclose(bp.v2f.cpu().numpy()[hid_e], o.v2f[hid_e], rtol=1e-9, atol=1e-8)
        np.testing.assert_allclose(bp.q_dev.cpu().numpy()[flat.var_hidden], o.q[flat.var_hidden], rtol=1e-9, atol=1e-12)
        got, want = bp.f2v.cpu().numpy()[hid_e], o.f2v[hid_e]
        np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-8)


def _with_domain(flat, dom):
    """`flat` with its first (continuous) domain replaced"""
    from lhvi.flat import build_flat
    specs = [(int(k), flat.pot_param[int(o):int(o2)].tolist()) for k, o, o2 in zip(flat.pot_kind, flat.pot_off[:-1], flat.pot_off[1:])]
    lo, hi = dom.values
    return build_flat(flat.fac_ptr, flat.edge_var, flat.fac_pot, specs, np.clip(flat.var_value, lo, hi), flat.var_dom,
                      [dom, flat.domains[1]])


@pytest.mark.parametrize('grid', ['uniform', 'uneven', 'T48', 'wide', 'T100', 'T128'])
def test_integral_points_by_grid_recurrence_match_the_direct_form(api, grid):
    """the heavy f2v kernel tabulates exp(a_j + b_j x_t) along a uniform integral-point grid by multiplication and
    reduce-scatters the sums over the lanes; LHVI_PBP_NO_GRID forces one exponential per term.  Same messages to 1e-12
    (uniform grids of 32 and 48 points), identical ones where the recurrence may not run: a grid that is not uniform,
    or exponents too close to the double range (domain [-40, 40]: the guard sends those edges through the direct form)."""
    import torch
    from lhvi import synth, _abi
    from lhvi.graph import Domain
    from lhvi.pbp import EPBP
    lo, hi = (-40.0, 40.0) if grid == 'wide' else (-10.0, 10.0)
    pts = np.linspace(lo, hi, {'T48': 48, 'T100': 100, 'T128': 128}.get(grid, 32))      # (100: the grid of the reference's RGM domain)
    if grid == 'uneven':
        pts = np.sign(pts) * np.abs(pts) ** 1.3 / 10 ** 0.3
    flat = _with_domain(synth.hybrid_mrf_flat(V=3000, deg=4, seed=11, frac_discrete=0.1),
                        Domain((lo, hi), continuous=True, integral_points=pts))
    bp = EPBP(None, n=64, proposal_approximation='simple', sampler='device', seed=2)
    bp.long_grid_min_edges = 0          # (100 / 128 points: more than two rounds of output points -- heavy kernel at any list length)
    bp._setup(None, flat=flat)
    _init(api, bp)
    for _ in range(2):
        bp.sweep(last=False)
    l, st = api.lib(), api.stream_ptr()
    s = bp._struct()
    api.check(l.lhvi_pbp_f2v(bp.dg.g, bp.dg.p, s, api.ptr(bp.v2f), api.ptr(bp.f2v), st))
    with_grid = bp.f2v.clone()
    s.flags |= _abi.PBP_NO_GRID
    api.check(l.lhvi_pbp_f2v(bp.dg.g, bp.dg.p, s, api.ptr(bp.v2f), api.ptr(bp.f2v), st))
    words = bp.heavy_desc.view(torch.int32).view(-1, 32).cpu().numpy()
    heavy_e = words[:, 0]
    a, b = with_grid.cpu().numpy()[heavy_e], bp.f2v.cpu().numpy()[heavy_e]
    assert np.isfinite(a).all() and heavy_e.size > 1000
    n = bp.n
    assert (a[:, :n] == b[:, :n]).all()                         # the particle part is the same code either way
    if grid == 'uneven':
        assert (words[:, 15] == 0).all() and (a == b).all()
        return
    assert (words[:, 15] == 1).all()
    np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-12)
    same = (a[:, n:] == b[:, n:]).all(axis=1)                   # edges whose integral points took the direct form
    live = words[:, 7] >= 24                                    # enough partner particles for the recurrence to pay
    assert not same[live].all()                                 # the recurrence did run ...
    if grid == 'wide':
        assert same[live].any()                                 # ... and the guard did reject edges with exponents ~ +-800


@pytest.mark.parametrize('case', ['n10', 'n16', 'n20', 'n32', 'n12 uneven', 'n10 wide', 'n24 wide', 'n16 T100', 'n20 T48'])
def test_few_particle_kernel_grid_recurrence_matches_the_direct_form(api, case):
    """``pbp_f2v_small_kernel<16 / 32>`` (four / two edges per wavefront): the integral points of an edge with a uniform grid by the
    recurrence inside its lane group against the direct form (LHVI_PBP_NO_GRID) -- the particle part identical, the grid part to
    1e-12; identical throughout on a grid that is not uniform; on the domain [-40, 40] the range guard sends an edge's points
    through the direct rounds (decided per edge: the same bits whatever shares its wavefront)"""
    import torch
    from lhvi import synth, _abi
    from lhvi.graph import Domain
    from lhvi.pbp import EPBP
    n = int(case.split()[0][1:])
    kind = case.split()[1] if ' ' in case else 'uniform'
    lo, hi = (-40.0, 40.0) if kind == 'wide' else (-10.0, 10.0)
    pts = np.linspace(lo, hi, {'T48': 48, 'T100': 100}.get(kind, 32))
    if kind == 'uneven':
        pts = np.sign(pts) * np.abs(pts) ** 1.3 / 10 ** 0.3
    flat = _with_domain(synth.hybrid_mrf_flat(V=3000, deg=4, seed=12, frac_discrete=0.1),
                        Domain((lo, hi), continuous=True, integral_points=pts))
    bp = EPBP(None, n=n, proposal_approximation='simple', sampler='device', seed=5)
    bp._setup(None, flat=flat)
    _init(api, bp)
    for _ in range(2):
        bp.sweep(last=False)
    l, st = api.lib(), api.stream_ptr()
    s = bp._struct()
    api.check(l.lhvi_pbp_f2v(bp.dg.g, bp.dg.p, s, api.ptr(bp.v2f), api.ptr(bp.f2v), st))
    with_grid = bp.f2v.clone()
    s.flags |= _abi.PBP_NO_GRID
    api.check(l.lhvi_pbp_f2v(bp.dg.g, bp.dg.p, s, api.ptr(bp.v2f), api.ptr(bp.f2v), st))
    desc = bp.small16_desc if n <= 16 else bp.small32_desc
    assert (bp.n_small16 if n <= 16 else bp.n_small32) > 1000 and bp.n_heavy == 0
    words = desc.view(torch.int32).view(-1, 32).cpu().numpy()
    e = words[:, 0]
    a, b = with_grid.cpu().numpy()[e], bp.f2v.cpu().numpy()[e]
    assert np.isfinite(a).all()
    assert (a[:, :n] == b[:, :n]).all()
    if kind == 'uneven':
        assert (words[:, 15] == 0).all() and (a == b).all()
        return
    assert (words[:, 15] == 1).all()
    np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-12)
    same = (a[:, n:] == b[:, n:]).all(axis=1)
    assert not same.all()                                       # the recurrence did run ...
    if kind == 'wide':
        assert same.any()                                       # ... and the guard did reject edges with exponents ~ +-800



@pytest.mark.parametrize('case', ['n10', 'n9', 'n12', 'n11 T48', 'n20', 'n18 T100', 'n10 wide', 'n20 uneven', 'n24', 'n22 T48', 'n32', 'n28 T100',
                                  'n17 wide', 'n32 uneven', 'n16', 'n13 T48', 'n16 wide', 'n14 uneven', 'n15 T100'])
def test_few_particle_kernel_narrow_lane_groups_match_the_16_and_32_lane_groups(api, case):
    """``pbp_f2v_small_kernel<10>`` (six edges per wavefront when no variable holds more than 10 particles; partial
    sums of the grid recurrence through wave-private LDS) and ``<6 / 8 / 10 / 12 / 16, 2>`` (two particles per lane: ten / eight / six / five /
    four edges per wavefront for up to 12 / 16 / 20 / 24 / 32 particles) against the 16- / 32-lane groups (LHVI_PBP_POW2_GROUPS): the particle part bit for bit
    -- the same term loop over the same records -- the grid part to 1e-12 (another order of the sum over the partner's particles),
    ragged list ends and rejected edges (domain [-40, 40]) included"""
    import torch
    from lhvi import synth, _abi
    from lhvi.graph import Domain
    from lhvi.pbp import EPBP
    n = int(case.split()[0][1:])
    kind = case.split()[1] if ' ' in case else 'uniform'
    lo, hi = (-40.0, 40.0) if kind == 'wide' else (-10.0, 10.0)
    pts = np.linspace(lo, hi, {'T48': 48, 'T100': 100}.get(kind, 32))
    if kind == 'uneven':
        pts = np.sign(pts) * np.abs(pts) ** 1.3 / 10 ** 0.3
    flat = _with_domain(synth.hybrid_mrf_flat(V=3001, deg=4, seed=14, frac_discrete=0.1),
                        Domain((lo, hi), continuous=True, integral_points=pts))
    bp = EPBP(None, n=n, proposal_approximation='simple', sampler='device', seed=6)
    bp._setup(None, flat=flat)
    _init(api, bp)
    for _ in range(2):
        bp.sweep(last=False)
    l, st = api.lib(), api.stream_ptr()
    s = bp._struct()
    bp.f2v.zero_()
    api.check(l.lhvi_pbp_f2v(bp.dg.g, bp.dg.p, s, api.ptr(bp.v2f), api.ptr(bp.f2v), st))
    narrow = bp.f2v.clone()
    bp.f2v.zero_()
    s.flags |= _abi.PBP_POW2_GROUPS
    api.check(l.lhvi_pbp_f2v(bp.dg.g, bp.dg.p, s, api.ptr(bp.v2f), api.ptr(bp.f2v), st))
    desc = bp.small16_desc if n <= 16 else bp.small32_desc
    nitems = bp.n_small16 if n <= 16 else bp.n_small32
    assert nitems > 1000 and bp.n_heavy == 0
    words = desc.view(torch.int32).view(-1, 32).cpu().numpy()
    e = words[:, 0]
    a, b = narrow.cpu().numpy()[e], bp.f2v.cpu().numpy()[e]
    assert np.isfinite(a).all()
    assert (a[:, :n] == b[:, :n]).all()
    # (log-messages below -690 come out of sums the term loop holds as denormals -- it carries them scaled by 2^-24 -- with a
    # handful of significant bits: there the two builds of the same loop differ by their last-bit roundings, 1e-3 at most)
    deep = b < -690.0
    np.testing.assert_allclose(a[~deep], b[~deep], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(a[deep], b[deep], rtol=0, atol=2e-3)
    assert torch.equal(narrow.cpu()[np.setdiff1d(np.arange(flat.E), e)], bp.f2v.cpu()[np.setdiff1d(np.arange(flat.E), e)])
    if kind == 'uneven':
        assert (a == b).all()                                   # no recurrence on a grid that is not uniform
    else:
        assert not (a[:, n:] == b[:, n:]).all()                 # the two reductions do differ in the last bits somewhere


def test_device_sampler_statistics(api):
    """Philox/Box-Muller particles: mean/variance of the clipped normal draws, determinism per (seed, iteration)"""
    from lhvi import synth
    from lhvi.pbp import EPBP
    flat = synth.hybrid_mrf_flat(V=4000, deg=4, seed=1)
    runs = []
    for _ in range(2):
        bp = EPBP(None, n=64, proposal_approximation='simple', sampler='device', seed=9)
        bp._setup(None, flat=flat)
        _init(api, bp)
        runs.append(bp.particles.cpu().numpy())
    cont = flat.var_hidden & flat.var_cont
    x = runs[0][cont].ravel()
    assert abs(x.mean()) < 0.02 and abs(x.var() - 5.0) < 0.1      # q = (0, 5), bounds +-10 (4.5 sigma)
    assert (runs[0] == runs[1]).all()


@pytest.mark.parametrize('n', [64, 40, 10])
def test_device_sampler_draws_do_not_depend_on_the_pairing(api, n):
    """the sampler draws for two variables per wavefront (one Philox block gives the cosine and the sine normal: particles j and
    j + 32).  Which two variables share a wavefront -- the caller's list of hidden continuous variables two by two, or neighbours
    of a variable range, or a range that starts one variable later -- must not change a single bit; rows of discrete and observed
    variables are written once and then left alone by the listed form"""
    import torch
    from lhvi import synth
    from lhvi.pbp import EPBP
    flat = synth.hybrid_mrf_flat(V=3001, deg=4, seed=1)
    bp = EPBP(None, n=n, proposal_approximation='simple', sampler='device', seed=9)
    bp._setup(None, flat=flat)
    _init(api, bp)                      # first draw: fills every row of both buffers
    assert bp._static_rows and bp.resample_vars.shape[0] == int((flat.var_hidden & flat.var_cont).sum())
    bp._generate_sample()               # second draw: the listed form
    k = bp._draws - 1
    l, st = api.lib(), api.stream_ptr()

    def by_range(lo, hi):
        s = bp._struct()
        s.var_lo, s.var_hi = lo, hi
        out, uq = torch.full_like(bp.particles, -77.0), torch.full_like(bp.uniq, 9)
        api.check(l.lhvi_pbp_resample_uniq(bp.dg.g, s, None, int(bp.seed), int(k), api.ptr(out), api.ptr(uq), st))
        return out, uq
    whole, uw = by_range(0, 0)
    live = torch.from_numpy(np.arange(n)[None, :] < bp.np_host[:, None]).to(whole.device)
    assert torch.equal(torch.where(live, whole, 0.0), torch.where(live, bp.particles, 0.0)) and torch.equal(uw, bp.uniq)
    # the list cut after an odd and after an even number of records: the last wavefront of the first works on one variable
    for k in (bp.resample_vars.shape[0] - 1, bp.resample_vars.shape[0] - 2, 1):
        s = bp._struct()
        s.resample_vars, s.n_resample_vars = api.ptr(bp.resample_vars), int(k)
        out, uq = torch.full_like(bp.particles, -77.0), torch.full_like(bp.uniq, 9)
        api.check(l.lhvi_pbp_resample_uniq(bp.dg.g, s, None, int(bp.seed), int(bp._draws - 1), api.ptr(out), api.ptr(uq), st))
        rows = bp.resample_vars[:k, 0].long()
        rest = torch.ones(flat.V, dtype=torch.bool, device=out.device)
        rest[rows] = False
        assert torch.equal(torch.where(live, out, 0.0)[rows], torch.where(live, bp.particles, 0.0)[rows]) and torch.equal(uq[rows], bp.uniq[rows])
        assert (out[rest] == -77.0).all() and (uq[rest] == 9).all()
    shifted, us = by_range(1, flat.V)   # other neighbours share a wavefront now
    assert torch.equal(torch.where(live, shifted, 0.0)[1:], torch.where(live, whole, 0.0)[1:]) and torch.equal(us[1:], uw[1:])
    assert (shifted[0] == -77.0).all() and (us[0] == 9).all()           # outside the range: untouched
    if n == 64:
        cont = flat.var_hidden & flat.var_cont
        x = bp.particles.cpu().numpy()[cont]
        assert abs(x.mean()) < 0.02 and abs(x.var() - 5.0) < 0.1
        a, b = x[:, :32].ravel(), x[:, 32:].ravel()                      # the two normals of a block are independent
        assert abs(np.corrcoef(a, b)[0, 1]) < 0.01 and abs(np.corrcoef(a * a, b * b)[0, 1]) < 0.01
        assert abs(a.var() - 5.0) < 0.1 and abs(b.var() - 5.0) < 0.1
        z = x.ravel() / np.sqrt(5.0)
        assert abs((z ** 4).mean() - 3.0) < 0.05                         # (clipping at 4.5 sigma removes ~1e-4 of it)


@pytest.mark.parametrize('n', [64, 48])
def test_device_sampler_first_occurrence_mask(api, n):
    """fused draw + mask against the stand-alone exact mask kernel: wide proposals on a narrow domain clip most draws to
    the bounds (many exact duplicates), tight ones none; n = 48 takes the partial-wave path"""
    import torch
    from lhvi import synth
    from lhvi.pbp import EPBP
    flat = synth.hybrid_mrf_flat(V=3000, deg=4, seed=2)
    bp = EPBP(None, n=n, proposal_approximation='simple', sampler='device', seed=5)
    bp._setup(None, flat=flat)
    _init(api, bp)
    q = bp.q_dev.cpu().numpy()
    rng = np.random.default_rng(0)
    q[:, 0] = rng.uniform(-12, 12, flat.V)
    q[:, 1] = np.where(rng.random(flat.V) < 0.5, 400.0, 0.01)
    bp.q_dev.copy_(api.to_dev(q))
    bp._generate_sample()
    fused = bp.uniq.cpu().numpy().copy()
    P = bp.particles.cpu().numpy()
    exact = torch.zeros_like(bp.uniq)
    api.check(api.lib().lhvi_pbp_uniq(bp.dg.g, n, api.ptr(bp.particles), api.ptr(bp.np_dev), api.ptr(exact), api.stream_ptr()))
    exact = exact.cpu().numpy()
    np.testing.assert_array_equal(fused, exact)
    cont = np.flatnonzero(flat.var_hidden & flat.var_cont)
    first = np.array([[P[v, j] not in P[v, :j] for j in range(n)] for v in cont[:300]])
    np.testing.assert_array_equal(exact.reshape(flat.V, -1)[cont[:300], :n].astype(bool), first)
    assert (~first).sum() > 1000            # the duplicates are really there


def test_device_exp_accuracy(api):
    """the f2v kernel's table-driven exp stays within 2 ulp of libm over the whole log-message range"""
    import torch
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-745, 709, 200000), rng.uniform(-5, 5, 200000), rng.uniform(-1e-3, 1e-3, 10000),
                        np.array([0.0, -745.0, -800.0, -1e5, 709.7, 710.0, 1e4])])
    xd = api.to_dev(x)
    yd = torch.empty_like(xd)
    api.check(api.lib().lhvi_debug_exp(api.ptr(xd), api.ptr(yd), x.size, api.stream_ptr()))
    y = yd.cpu().numpy()
    want = np.exp(x)
    fin = np.isfinite(want) & (want > 1e-300)
    ulp = np.abs(y[fin] - want[fin]) / np.spacing(want[fin])
    assert ulp.max() <= 2.0, ulp.max()
    assert (y[x < -760] == 0).all() and np.isinf(y[x > 709.9]).all()


def test_device_log_accuracy(api):
    """the f2v epilogue's table-driven log: <= 2 ulp away from 1 and < 2.5e-16 absolute next to it, from the denormals to
    the top of the range; the series log (which = 1): <= 2 ulp everywhere"""
    import torch
    rng = np.random.default_rng(1)
    x = np.concatenate([np.exp(rng.uniform(-744, 709, 300000)), rng.uniform(0.5, 2.0, 200000), 1.0 + rng.uniform(-1e-6, 1e-6, 10000),
                        np.array([1.0, 0.5, 2.0, np.sqrt(0.5), np.sqrt(2.0), 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308, np.inf])])
    xd = api.to_dev(x)
    want = np.log(x)
    fin = np.isfinite(want)
    for which in (0, 1):
        yd = torch.empty_like(xd)
        api.check(api.lib().lhvi_debug_log(api.ptr(xd), api.ptr(yd), x.size, which, api.stream_ptr()))
        y = yd.cpu().numpy()
        scale = np.abs(want[fin]) if which == 1 else np.maximum(np.abs(want[fin]), 1.0)
        scale = np.where(scale == 0, 1.0, scale)
        ulp = np.abs(y[fin] - want[fin]) / np.spacing(scale)
        assert ulp.max() <= (2.0 if which == 1 else 2.25), (which, ulp.max(), x[fin][ulp.argmax()])
        assert abs(y[x == 1.0][0]) <= (0.0 if which == 1 else 2.5e-16) and np.isinf(y[-1])


def test_device_exp_accumulate_accuracy(api):
    """the term loop's exp(t + C): C folded into the rounding constant (exactly), one-constant range reduction of t.
    Relative error <= 6e-16 + 3.8e-17 |t| -- less than the rounding that t = a + b x already carries (1.1e-16 |t|)"""
    import torch
    rng = np.random.default_rng(2)
    t = np.concatenate([rng.uniform(-700, 50, 200000), rng.uniform(-20, 20, 200000), np.array([-800.0, -1e4, 0.0])])
    c = np.concatenate([rng.uniform(-600, 30, 200000), rng.uniform(-20, 20, 200000), np.array([0.0, 0.0, 0.0])])
    keep = (t + c < 700)
    t, c = t[keep], c[keep]
    td, cd = api.to_dev(t), api.to_dev(c)
    yd = torch.empty_like(td)
    api.check(api.lib().lhvi_debug_exp_acc(api.ptr(td), api.ptr(cd), api.ptr(yd), t.size, api.stream_ptr()))
    y = yd.cpu().numpy()
    want = np.exp(np.longdouble(t) + np.longdouble(c)).astype(np.float64)
    fin = want > 1e-300
    rel = np.abs(y[fin] - want[fin]) / want[fin]
    bound = 6e-16 + 3.8e-17 * np.abs(t[fin])          # 2 ulp of the table/polynomial + |t| * (step - RN(step)) / step
    assert (rel <= bound).all(), (rel / bound).max()
    assert (y[t + c < -760] == 0).all()


def test_device_exp_accumulate_floor_form_accuracy(api):
    """the heavy / cq kernels' term: s = t / step, floor(s) from an addition under round-down, v_fract as the polynomial
    argument.  Relative error <= 1.2e-15 + 2.7e-16 |t| (the scaling by 1 / step rounds t once more than the round-to-nearest form
    does), exact zero far below the double range, and no value off by a table step -- which is what an inconsistent floor / fract
    pair would produce (3.4e-4 relative)"""
    import torch
    rng = np.random.default_rng(3)
    step = np.log(2.0) / 2048
    near = (rng.integers(-200000, 20000, 100000) + rng.choice([0.0, 1e-12, -1e-12, 0.5, 0.25, -0.5], 100000)) * step      # s next to integers / ties
    t = np.concatenate([rng.uniform(-700, 50, 200000), rng.uniform(-20, 20, 200000), near, np.array([-800.0, -1e4, 0.0])])
    c = np.concatenate([rng.uniform(-600, 30, 200000), rng.uniform(-20, 20, 200000), rng.uniform(-5, 5, 100000), np.array([0.0, 0.0, 0.0])])
    keep = (t + c < 700)
    t, c = t[keep], c[keep]
    td, cd = api.to_dev(t), api.to_dev(c)
    yd = torch.empty_like(td)
    api.check(api.lib().lhvi_debug_exp_acc_floor(api.ptr(td), api.ptr(cd), api.ptr(yd), t.size, api.stream_ptr()))
    y = yd.cpu().numpy()
    want = np.exp(np.longdouble(t) + np.longdouble(c)).astype(np.float64)
    fin = want > 1e-290
    rel = np.abs(y[fin] - want[fin]) / want[fin]
    bound = 1.2e-15 + 2.7e-16 * np.abs(t[fin])
    assert (rel <= bound).all(), (rel / bound).max()
    assert (y[t + c < -780] == 0).all()
    # and the rounding mode is back to nearest afterwards: the round-to-nearest form still meets its own bound
    api.check(api.lib().lhvi_debug_exp_acc(api.ptr(td), api.ptr(cd), api.ptr(yd), t.size, api.stream_ptr()))
    rel = np.abs(yd.cpu().numpy()[fin] - want[fin]) / want[fin]
    assert (rel <= 6e-16 + 3.8e-17 * np.abs(t[fin])).all()


@pytest.mark.parametrize('name', C2F_CASES)
def test_hlbp_coarse_to_fine_matches_reference(api, golden_dir, name):
    """c2f=0: coarse start, per-sweep refinement with message inheritance; partitions and proposals at every draw and the
    final log-beliefs / MAPs / normalised beliefs against the reference (its particles injected per cluster)"""
    from lhvi.pbp import HybridLBP
    from oracle import oracle
    z, meta = load_npz(golden_dir, name)
    g, rvs, factors = modelio.load_model(meta['model'], API)
    samples = z['samples']
    q_at_draw = []

    def inject(k, flat, q):
        q_at_draw.append(q.copy())
        return samples[k][flat.rep_ground]

    bp = HybridLBP(g, n=meta['n'], proposal_approximation=meta['approx'], sampler=inject)
    # the variable-side message / site tables at every draw against the reference's (HLBP:268-308 inheritance included)
    bp.c2f_observer = c2f_table_observer(z, rvs, factors, meta['n'], lambda t: t.cpu().numpy())
    bp.run(meta['iterations'], c2f=meta['c2f'])
    hist = bp.c2f_history
    assert len(hist) == z['draw_rv_labels'].shape[0]
    for k, (r, f) in enumerate(hist):
        assert oracle.canonical_labels(r) == z['draw_rv_labels'][k].tolist(), 'rv partition at draw %d' % k     # exact
        assert oracle.canonical_labels(f) == z['draw_f_labels'][k].tolist(), 'factor partition at draw %d' % k
        want = z['draw_q'][k]
        m = ~np.isnan(want[:, 0])
        np.testing.assert_allclose(q_at_draw[k][r][m], want[m], rtol=1e-8, atol=1e-10, err_msg='q at draw %d' % k)
    rv_color, f_color = bp.g.colors()
    assert oracle.canonical_labels(rv_color) == z['rv_label'].tolist()
    assert oracle.canonical_labels(f_color) == z['f_label'].tolist()
    hid = [i for i, rv in enumerate(rvs) if rv.value is None]
    got = bp.belief_rv_batch([rvs[i] for i in hid], z['query_x'][hid])
    np.testing.assert_allclose(got, z['query_logb'][hid], rtol=1e-8, atol=1e-6)
    for i in hid[:5]:
        _check_per_variable_map(bp, rvs[i], z['map'][i], lambda x, i=i: bp.belief_rv_query(float(x), rvs[i]))
        assert bp.belief(z['query_x'][i][2], rvs[i]) == pytest.approx(z['belief_mid'][i], rel=1e-5, abs=1e-7)
    assert 'ground_map' in bp._batched and 'ground_area' in bp._batched      # unstable partition: batched over the GROUND variables
    # the reference's demo loop (Demo/RGM/demo.py:32-35): every rv's MAP -- one batched pass, then dictionary look-ups
    all_maps = [bp.map(rv) for rv in rvs]
    assert len(all_maps) == len(rvs) and all(np.isfinite(m) for m in all_maps)
    for i, a, b, want in z['probability']:
        assert bp.probability(a, b, rvs[int(i)]) == pytest.approx(want, rel=1e-7, abs=1e-300)


def paper_popularity(P, T, seed, points=32):
    """the paper-popularity hybrid MLN of Demo/Data/HMLN/GeneratorPaperPopularity.py:7-72 (atoms, the three parametric
    factors with their weights, evidence pattern of generate_data) built with THIS package's relational API, and the
    demo's domain Domain((-15, 15), integral_points=linspace(0, 10, 32)) (Demo/HMLN/DemoPaperPopularity.py)"""
    from lhvi.graph import Domain
    from lhvi.mln import MLNPotential, eq_op
    from lhvi.relational import LV, Atom, ParamF, RelationalGraph
    rng = np.random.default_rng(seed)
    dom_b = Domain((0, 1))
    dom_r = Domain((-15, 15), continuous=True, integral_points=np.linspace(0, 10, points))
    lvp, lvt = LV(['p%d' % i for i in range(P)]), LV(['t%d' % i for i in range(T)])
    atoms = (Atom(dom_b, (lvt, lvt), 'SameSession'), Atom(dom_b, (lvp, lvt), 'PaperIn'),
             Atom(dom_r, (lvt,), 'TopicPopularity'), Atom(dom_r, (lvp,), 'PaperPopularity'))
    pfs = (ParamF(MLNPotential(lambda x: eq_op(x[0], 1), w=0.3), nb=['PaperPopularity(p)']),
           ParamF(MLNPotential(lambda x: x[0] * eq_op(x[1], x[2]), w=0.5),
                  nb=['SameSession(t1,t2)', 'TopicPopularity(t1)', 'TopicPopularity(t2)'], constrain=lambda s: s['t1'] != s['t2']),
           ParamF(MLNPotential(lambda x: x[0] * eq_op(x[1], x[2]), w=1),
                  nb=['PaperIn(p,t)', 'PaperPopularity(p)', 'TopicPopularity(t)']))
    rel = RelationalGraph(atoms, pfs)
    g, table = rel.ground_graph()
    data = {}
    for i in rng.choice(P, int(P * 0.7), replace=False):
        data[('PaperPopularity', 'p%d' % i)] = float(rng.integers(0, 5)) * 2.5     # few distinct values: lifting merges
    for i in rng.choice(T, int(T * 0.7), replace=False):
        data[('TopicPopularity', 't%d' % i)] = float(rng.uniform(0, 10))
    for i in rng.choice(P, int(P * 0.7), replace=False):
        for j in rng.choice(T, int(rng.integers(T)), replace=False):
            data[('PaperIn', 'p%d' % i, 't%d' % j)] = int(rng.integers(0, 2))
    for i in range(T):
        for j in rng.choice(T, T // 2, replace=False):
            if i != j:
                data[('SameSession', 't%d' % i, 't%d' % j)] = int(rng.integers(0, 2))
    rel.add_evidence(data)
    g.rvs, g.factors = sorted(g.rvs), sorted(g.factors)
    g.init_nb()
    return g, table


def test_cfg3_paper_popularity_hmln_full_size_matches_oracle(api):
    """BASELINE.json cfg 3 at its stated size: the paper-popularity HMLN grounded for 300 papers x 10 topics (V = 3 400,
    F = 3 390, E = 9 570; ternary MLN factors with one boolean and two continuous arguments), HybridLBP n = 10 with the
    demo's 32 integral points, lifted (c2f = -1): every sweep's tables against the C oracle on the same lifted graph with
    the same particles, then beliefs of query atoms within 1e-5"""
    from lhvi.pbp import HybridLBP
    from oracle import oracle
    g, table = paper_popularity(300, 10, seed=3)
    assert len(g.rvs) == 3400 and len(g.factors) == 3390 and sum(len(f.nb) for f in g.factors) == 9570
    n, its = 10, 4
    rng = np.random.default_rng(8)
    samples = []

    def sampler(k, flat, q):
        cont = flat.var_hidden & flat.var_cont
        lo, hi = flat.dom_lo[flat.var_dom], flat.dom_hi[flat.var_dom]
        out = np.zeros((flat.V, n))
        out[cont] = np.clip(rng.standard_normal((int(cont.sum()), n)) * np.sqrt(q[cont, 1:2]) + q[cont, 0:1],
                            lo[cont, None], hi[cont, None])
        samples.append(out)
        return out

    bp = HybridLBP(g, n=n, proposal_approximation='simple', sampler=sampler)
    bp.run(its)
    flat = bp.flat
    assert bp.T == 32 and flat.V < 3400                       # lifted: fewer clusters than ground variables
    o = oracle.PbpOracle(flat, n, ep=False, epbp=False, var_threshold=5)
    o.run(its, samples)
    hid_e = flat.var_hidden[flat.edge_var]
    cont = flat.var_hidden & flat.var_cont
    np.testing.assert_allclose(bp.q_dev.cpu().numpy()[cont], o.q[cont], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(bp.eta.cpu().numpy()[cont[flat.edge_var]], o.eta[cont[flat.edge_var]], rtol=1e-9, atol=1e-12)
    npe = o.np[flat.edge_var]
    live = hid_e[:, None] & (np.arange(n)[None, :] < npe[:, None])
    np.testing.assert_allclose(bp.v2f.cpu().numpy()[live], o.v2f[live], rtol=RTOL, atol=ATOL)
    got, want = bp.f2v.cpu().numpy(), o.f2v
    np.testing.assert_allclose(got[:, :n][live], want[:, :n][live], rtol=RTOL, atol=ATOL)
    ce = cont[flat.edge_var]
    np.testing.assert_allclose(got[ce, n:], want[ce, n:], rtol=RTOL, atol=ATOL)
    # normalised beliefs of the query atoms (PaperPopularity / TopicPopularity, as in the demo) within 1e-5 of the oracle's
    query = [rv for key, rv in table.items() if key[0] in ('PaperPopularity', 'TopicPopularity') and rv.value is None][:12]
    for rv in query:
        c = flat.var_index[rv.cluster]
        lb = lambda xs: o.belief_points(np.array([c]), np.asarray(xs, dtype=float)[None, :])[0]
        x = np.linspace(-15, 15, 20)
        y = lb(x)
        shift = y.mean() if y.max() - y.mean() <= 700 else y.max() - 700
        w = np.exp(y - shift)
        zarea = ((w[:-1] + w[1:]) * (x[1] - x[0])).sum() * 0.5
        for xq in (2.0, 5.5):
            want_b = float(np.exp(lb([xq])[0] - shift) / zarea)
            assert bp.belief(xq, rv) == pytest.approx(want_b, rel=1e-5, abs=1e-12)


@pytest.mark.parametrize('solver, n', [('epbp', 12), ('epbp', 64), ('hlbp', 16)])
def test_conditionally_quadratic_routing_equals_the_generic_kernel(api, solver, n):
    """The reference's HMLN formulas (x[0] * eq_op(x[1], x[2]), MLNPotential.py:26-27) are quadratic in the continuous
    arguments per state of the boolean: with LHVI_PBP_CQ their edges are served by the heavy / light / cq kernels.  Same
    graph, same particles, routing on vs off (every MLN edge through the bytecode interpreter of the generic kernel):
    every table of every sweep within 1e-9, and every route must actually occur."""
    from lhvi.pbp import EPBP, HybridLBP
    g, table = paper_popularity(40, 5, seed=5)
    rng = np.random.default_rng(n)
    its = 4
    draws = []

    def sampler(k, flat, q):
        if k == len(draws):
            cont = flat.var_hidden & flat.var_cont
            lo, hi = flat.dom_lo[flat.var_dom], flat.dom_hi[flat.var_dom]
            out = np.zeros((flat.V, n))
            out[cont] = np.clip(rng.standard_normal((int(cont.sum()), n)) * np.sqrt(q[cont, 1:2]) + q[cont, 0:1],
                                lo[cont, None], hi[cont, None])
            draws.append(out)
        return draws[k]

    runs = []
    for routed in (True, False):
        bp = (EPBP(g, n=n, proposal_approximation='simple', sampler=sampler) if solver == 'epbp'
              else HybridLBP(g, n=n, proposal_approximation='EP', sampler=sampler))
        bp.cq_routing = routed
        bp.run(its)
        runs.append(bp)
    a, b = runs
    assert a.flags & api.PBP_CQ and not (b.flags & api.PBP_CQ)
    assert b.n_cq == 0 and b.n_heavy_class == 0 and b.n_light == 0 and int(b.generic_edges.numel()) > 0
    import torch
    types = a.cq_desc.view(torch.int32).view(a.n_cq, 64)[:, 2].cpu().numpy()
    assert a.n_heavy_class > 0 and a.n_light > 0 and (types == 1).any() and (types == 2).any()
    # what is left on the generic list are the messages to booleans whose other arguments are all observed
    assert int(a.generic_edges.numel()) < int(b.generic_edges.numel())
    flat = a.flat
    hid_e = flat.var_hidden[flat.edge_var] & (flat.edge_canon == np.arange(flat.E))
    npe = a.np_host[flat.edge_var]
    live = hid_e[:, None] & (np.arange(n)[None, :] < npe[:, None])
    ce = hid_e & flat.var_cont[flat.edge_var]
    fa, fb = a.f2v.cpu().numpy(), b.f2v.cpu().numpy()
    np.testing.assert_allclose(fa[:, :n][live], fb[:, :n][live], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(fa[ce, n:], fb[ce, n:], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(a.v2f.cpu().numpy()[live], b.v2f.cpu().numpy()[live], rtol=1e-9, atol=1e-9)
    cont = flat.var_hidden & flat.var_cont
    np.testing.assert_allclose(a.q_dev.cpu().numpy()[cont], b.q_dev.cpu().numpy()[cont], rtol=1e-9, atol=1e-12)
    # and the routed run against the C oracle (bytecode evaluation on the host)
    from oracle import oracle
    o = oracle.PbpOracle(flat, n, ep=(solver == 'hlbp'), epbp=(solver == 'epbp'), var_threshold=3 if solver == 'epbp' else 5)
    o.run(its, draws)
    np.testing.assert_allclose(fa[:, :n][live], o.f2v[:, :n][live], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(fa[ce, n:], o.f2v[ce, n:], rtol=RTOL, atol=ATOL)


def test_batched_kl_of_tabulated_beliefs(api):
    """utils.kl_tables: trapezoid of kl_continuous' integrand for every variable at once (device), against the host quad"""
    from math import exp, pi, sqrt
    from lhvi import utils
    rng = np.random.default_rng(3)
    V, m = 7, 4001
    mu1, mu2 = rng.uniform(-1, 1, V), rng.uniform(-1, 1, V)
    s1, s2 = rng.uniform(0.5, 1.5, V), rng.uniform(0.5, 1.5, V)
    a, b = np.full(V, -14.0), np.full(V, 14.0)
    x = np.linspace(a, b, m, axis=1)
    pdf = lambda x, mu, s: np.exp(-0.5 * ((x - mu) / s) ** 2) / (sqrt(2 * pi) * s)
    got = utils.kl_tables(pdf(x, mu1[:, None], s1[:, None]), pdf(x, mu2[:, None], s2[:, None]), a, b).cpu().numpy()
    for v in range(V):
        want = utils.kl_continuous(lambda t: pdf(t, mu1[v], s1[v]), lambda t: pdf(t, mu2[v], s2[v]), -14, 14)
        assert got[v] == pytest.approx(want, rel=1e-6)
        assert got[v] == pytest.approx(utils.kl_normal(mu1[v], mu2[v], s1[v], s2[v]), rel=1e-6)


def test_heavy_kernel_work_distribution_does_not_change_results(api):
    """chunks claimed through the ticket (default) against static striding (f2v_ticket = NULL): every edge is computed by
    the same code either way, so whole sweeps agree bit for bit; also a list shorter than one chunk per XCD range"""
    import torch
    from lhvi import synth
    from lhvi.pbp import EPBP
    for V in (20000, 40):
        flat = synth.hybrid_mrf_flat(V=V, deg=4, seed=17)
        res = []
        for dynamic in (True, False):
            bp = EPBP(None, n=64, proposal_approximation='simple', sampler='device', seed=9)
            bp.dynamic_f2v = dynamic
            bp._setup(None, flat=flat)
            _init(api, bp)
            for _ in range(3):
                bp.sweep(last=False)
            torch.cuda.synchronize()
            res.append((bp.f2v.clone(), bp.v2f.clone(), bp.q_dev.clone()))
            assert bool(torch.isfinite(bp.f2v).all())
        for a, b in zip(*res):
            assert torch.equal(a, b)


def test_proposal_records_do_not_change_results(api):
    """with the per-variable records (prop_desc) the proposal kernel starts from one scalar load; without them it walks the
    graph arrays over every variable.  Same arithmetic per variable: whole sweeps agree bit for bit ('EP' and 'simple')"""
    import torch
    from lhvi import synth
    from lhvi.pbp import EPBP
    flat = synth.hybrid_mrf_flat(V=6000, deg=4, seed=23, frac_discrete=0.3)
    for mode in ('EP', 'simple'):
        res = []
        for listed in (True, False):
            bp = EPBP(None, n=64, proposal_approximation=mode, sampler='device', seed=11)
            bp.listed_proposal = listed
            bp._setup(None, flat=flat)
            _init(api, bp)
            for _ in range(4):
                bp.sweep(last=False)
            torch.cuda.synchronize()
            res.append((bp.f2v.clone(), bp.q_dev.clone(), bp.eta.clone(), bp.particles.clone()))
        for a, b in zip(*res):
            assert torch.equal(a, b)



def test_lifted_particle_sweep_on_arrays_matches_object_path(api):
    """paper-popularity HMLN: ground_flat -> initial_colors_flat -> refine_flat -> lift_flat -> HybridLBP.on_flat (no Python
    object per ground atom) against ground_graph -> HybridLBP on the objects: same partition size, and with the same
    particles per cluster the same proposals, MAPs and normalised beliefs"""
    from lhvi import lifting
    from lhvi.graph import Domain
    from lhvi.mln import MLNPotential, eq_op
    from lhvi.pbp import HybridLBP
    from lhvi.relational import LV, Atom, ParamF, RelationalGraph
    P_, T_ = 40, 5

    def template():
        dom_b = Domain((0, 1))
        dom_r = Domain((-15, 15), continuous=True, integral_points=np.linspace(0, 10, 32))
        lvp, lvt = LV(['p%d' % i for i in range(P_)]), LV(['t%d' % i for i in range(T_)])
        atoms = (Atom(dom_b, (lvt, lvt), 'SameSession'), Atom(dom_b, (lvp, lvt), 'PaperIn'),
                 Atom(dom_r, (lvt,), 'TopicPopularity'), Atom(dom_r, (lvp,), 'PaperPopularity'))
        pfs = (ParamF(MLNPotential(lambda x: eq_op(x[0], 1), w=0.3), nb=['PaperPopularity(p)']),
               ParamF(MLNPotential(lambda x: x[0] * eq_op(x[1], x[2]), w=0.5),
                      nb=['SameSession(t1,t2)', 'TopicPopularity(t1)', 'TopicPopularity(t2)'], constrain=lambda s: s['t1'] != s['t2']),
               ParamF(MLNPotential(lambda x: x[0] * eq_op(x[1], x[2]), w=1),
                      nb=['PaperIn(p,t)', 'PaperPopularity(p)', 'TopicPopularity(t)']))
        return RelationalGraph(atoms, pfs)
    rng = np.random.default_rng(4)
    ev = {}
    for i in rng.choice(P_, 28, replace=False):
        ev[('PaperPopularity', 'p%d' % i)] = float(rng.integers(0, 3)) * 3.0
    for i in range(P_):
        ev[('PaperIn', 'p%d' % i, 't0')] = int(i % 2)
    rel_o, rel_f = template(), template()
    g, table = rel_o.ground_graph()
    rel_o.add_evidence(ev)
    g.rvs, g.factors = sorted(g.rvs), sorted(g.factors)
    g.init_nb()
    n, its = 10, 4
    samples = {}
    rng2 = np.random.default_rng(9)

    def by_key(tag):
        """the same particles for the same cluster on both paths: keyed by (draw, smallest ground atom key of the cluster)"""
        def sampler(k, flat, q):
            out = np.zeros((flat.V, n))
            for c in range(flat.V):
                if flat.var_hidden[c] and flat.var_cont[c]:
                    key = (k, tag(flat, c))
                    if key not in samples:
                        samples[key] = rng2.uniform(0.5, 9.5, n)
                    out[c] = samples[key]
            return out
        return sampler
    key_of = {id(rv): k for k, rv in table.items()}
    obj = HybridLBP(g, n=n, proposal_approximation='simple',
                    sampler=by_key(lambda flat, c: min(str(key_of[id(r)]) for r in flat.rvs[c].rvs)))
    obj.run(its)
    flat, keys = rel_f.ground_flat(ev)
    rv0, f0, sym = lifting.initial_colors_flat(flat)
    rvc, fc = lifting.refine_flat(flat, sym, rv0, f0)
    assert int(rvc.max()) + 1 == obj.flat.V and int(fc.max()) + 1 == obj.flat.F and obj.flat.V < len(table)
    lflat = lifting.lift_flat(flat, rvc, fc)
    names = {}
    for k in table:
        c = int(rvc[keys.var_id(k)])
        names[c] = min(names.get(c, str(k)), str(k))
    arr = HybridLBP.on_flat(lflat, n=n, proposal_approximation='simple', sampler=by_key(lambda fl, c: names[c]))
    arr.run_flat(its)
    qa, qo = arr.q_dev.cpu().numpy(), obj.q_dev.cpu().numpy()
    ma, _ = arr.map_all(steps=8)
    mo, _ = obj.map_all(steps=8)
    xq = np.full((lflat.V, 1), 4.0)
    ba = arr.belief_all(xq).cpu().numpy()
    bo = obj.belief_all(np.full((obj.flat.V, 1), 4.0)).cpu().numpy()
    checked = 0
    for k, rv in table.items():
        if rv.value is None and rv.domain.continuous:
            ca, co = int(rvc[keys.var_id(k)]), obj.flat.var_index[rv.cluster]
            np.testing.assert_allclose(qa[ca], qo[co], rtol=1e-9)
            assert ma[ca] == pytest.approx(mo[co], abs=1e-6)
            assert ba[ca, 0] == pytest.approx(bo[co, 0], rel=1e-8)
            checked += 1
    assert checked > 10


@pytest.mark.parametrize('frac_discrete,evidence', [(0.3, 0.1), (0.5, 0.4), (0.2, 0.0)])
def test_pair_records_do_not_change_results(api, frac_discrete, evidence):
    """the light edges served per factor (lhvi_pbp_t.pair_desc, pbp_f2v_pair_kernel) against per edge (light_desc): the same
    expressions per edge, so whole sweeps agree bit for bit -- with hidden / observed variables on either side of the factors"""
    import torch
    from lhvi import synth
    from lhvi.pbp import EPBP
    flat = synth.hybrid_mrf_flat(V=8000, deg=4, seed=31, frac_discrete=frac_discrete, evidence_ratio=evidence)
    res = []
    for paired in (True, False):
        bp = EPBP(None, n=64, proposal_approximation='simple', sampler='device', seed=13)
        bp.paired_light = paired
        bp._setup(None, flat=flat)
        assert (bp.n_pair > 0) == paired and bp.n_light > 0
        if paired:       # every light edge appears in exactly one record
            w = bp.pair_desc.view(torch.int32).view(-1, 32)
            edges = torch.cat([w[:, 0][w[:, 0] >= 0], w[:, 1][w[:, 1] >= 0]])
            light_e = bp.light_desc.view(torch.int32).view(-1, 32)[:, 0]
            assert torch.equal(torch.sort(edges).values, torch.sort(light_e).values)
        _init(api, bp)
        for _ in range(4):
            bp.sweep(last=False)
        torch.cuda.synchronize()
        res.append((bp.f2v.clone(), bp.v2f.clone(), bp.q_dev.clone()))
        assert bool(torch.isfinite(bp.f2v).all())
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize('solver', ['epbp', 'hlbp', 'epbp10', 'epbp16', 'epbp24', 'epbp32'])
def test_packed_v2f_kernel_does_not_change_results(api, solver):
    """variables with at most four particles (binary variables, boolean atoms) served sixteen per wavefront by
    pbp_v2f_narrow_kernel against one wavefront each: whole sweeps agree bit for bit (hybrid MRF with binary and observed
    variables for EPBP; the paper-popularity HMLN, lifted with counts, for HybridLBP)"""
    import torch
    from lhvi import synth
    from lhvi.pbp import EPBP, HybridLBP
    runs = []
    for packed in (True, False):
        if solver.startswith('epbp'):
            # (10 ... 32 particles: the continuous variables go four / two to a wavefront as well, pbp_v2f_packed_kernel)
            flat = synth.hybrid_mrf_flat(V=9001, deg=4, seed=31, frac_discrete=0.4)
            bp = EPBP(None, n=int(solver[4:] or 64), proposal_approximation='simple', sampler='device', seed=6)
            bp.packed_v2f = packed
            bp._setup(None, flat=flat)
            _init(api, bp)
            for _ in range(4):
                bp.sweep(last=False)
        else:
            g, table = paper_popularity(30, 4, seed=9)
            np.random.seed(3)
            bp = HybridLBP(g, n=12, proposal_approximation='simple')
            bp.packed_v2f = packed
            bp.run(4)
        runs.append(bp)
    a, b = runs
    assert a.v2f_lists is not None and a.v2f_lists[3] > 0 and a.v2f_lists[5] == 0 and b.v2f_lists is None
    if solver in ('hlbp', 'epbp10', 'epbp16'):
        assert a.v2f_lists[7] > 0 and a.v2f_lists[1] == 0
    if solver in ('epbp24', 'epbp32'):
        assert a.v2f_lists[9] > 0 and a.v2f_lists[1] == 0
    for name in ('v2f', 'f2v', 'q_dev', 'eta', 'particles'):
        assert torch.equal(getattr(a, name), getattr(b, name)), name


@pytest.mark.parametrize('case', ['n10', 'n16', 'n20', 'n32', 'n12 T48', 'n24 T64', 'n10 EP', 'n20 EP', 'hlbp n12', 'hlbp n20 EP', 'hmln hubs n16'])
def test_fused_variable_kernel_equals_the_three_kernels(api, case):
    """few particles (the reference's demos run 10 - 20): ``lhvi_pbp_var_fused`` does a continuous variable's v -> f messages, its
    proposal update and its new sample + first-occurrence mask in one pass over its rows (lane groups of 16 / 32 lanes; T <= 32
    points over 16 lanes, T <= 64 over 32) -- against ``lhvi_pbp_v2f`` + ``lhvi_pbp_proposal`` + ``lhvi_pbp_resample_uniq``: every
    array of the state after whole sweeps, bit for bit ('simple' and 'EP' rule, ground and lifted with counts, hub rows left
    to the three kernels)"""
    import torch
    from lhvi import synth
    from lhvi.pbp import EPBP, HybridLBP
    n = int(case.split()[-2 if case.endswith('EP') else -1][1:]) if not case.startswith('n') else int(case.split()[0][1:])
    approx = 'EP' if case.endswith('EP') else 'simple'
    T = int(case.split()[1][1:]) if ' T' in case else 32
    runs = []
    for fused in (True, False):
        if case.startswith('hlbp'):
            g, table = paper_popularity(30, 4, seed=9)
            bp = HybridLBP(g, n=n, proposal_approximation=approx, sampler='device', seed=4)
            bp.fused_var_kernel, bp.fused_max_particles = fused, 32
            bp.run(5)
        else:
            if case.startswith('hmln'):
                flat, keys = synth.paper_popularity_flat(150, 4, seed=2)          # topics touch > 64 factors: not fused
            else:
                flat = synth.hybrid_mrf_flat(V=9001, deg=4, seed=31, frac_discrete=0.3, T=T)
            bp = EPBP(None, n=n, proposal_approximation=approx, sampler='device', seed=6)
            bp.fused_var_kernel, bp.fused_max_particles = fused, 32
            bp._setup(None, flat=flat)
            _init(api, bp)
            for _ in range(5):
                bp.sweep(last=False)
            bp.sweep(last=True)
        torch.cuda.synchronize()
        runs.append(bp)
    a, b = runs
    assert a._fused is not None and sum(a._fused['counts']) > 0 and b._fused is None
    c16, c32a, c32b = a._fused['counts']
    if not case.startswith('h'):
        assert (c16 > 0) == (n <= 16 and T <= 32) and (c32a > 0) == (n > 16 and T <= 32) and (c32b > 0) == (T > 32)
    if case.startswith('hmln'):
        assert a._fused['n_prop_rest'] > 0 and a.n_prop_hub > 0                    # the hub rows stay with the three kernels
    for name in ('q_dev', 'eta', 'particles', 'old_particles', 'uniq', 'v2f', 'f2v'):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    assert bool(torch.isfinite(a.q_dev[torch.from_numpy(a.flat.var_hidden & a.flat.var_cont).to(a.q_dev.device)]).all())


@pytest.mark.parametrize('n', [5, 9, 10, 12, 16, 18, 20, 24, 32])
def test_packed_pair_kernel_equals_the_one_entry_per_wave_kernel(api, n):
    """few particles: the HybridQuadratic(1 discrete, 1 continuous) factors' list goes four / three / two entries to a
    wavefront for n <= 16 / 20 / 32 (``pbp_f2v_pair_small_kernel<16 / 20 / 32>``; the groups of 20 lanes spell out the summation
    tree of the DPP networks); LHVI_PBP_WIDE_PAIRS keeps the one-entry-per-wave kernel: every array of the
    state after whole sweeps, bit for bit (observed variables on either side, one-sided records included)"""
    import torch
    from lhvi import _abi, synth
    from lhvi.pbp import EPBP
    runs = []
    for wide in (False, True):
        flat = synth.hybrid_mrf_flat(V=6001, deg=4, seed=17, frac_discrete=0.4, evidence_ratio=0.25)
        bp = EPBP(None, n=n, proposal_approximation='simple', sampler='device', seed=3)
        bp._setup(None, flat=flat)
        assert bp.n_pair > 0
        if wide:
            bp.flags |= _abi.PBP_WIDE_PAIRS
        _init(api, bp)
        for _ in range(4):
            bp.sweep(last=False)
        bp.sweep(last=True)
        torch.cuda.synchronize()
        runs.append(bp)
    a, b = runs
    for name in ('f2v', 'v2f', 'q_dev', 'eta', 'particles'):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    assert bool(torch.isfinite(a.f2v).all())


def test_f2v_half_sweep_on_two_streams_gives_the_same_bits(api, monkeypatch):
    """``EPBP.overlap_f2v``: the heavy kernel on the caller's stream (LHVI_PBP_SHARE_CUS: a workgroup per CU left free) and the pair /
    generic kernels beside it on a second stream write disjoint rows of f2v -- every array of the state after whole sweeps equals
    the one-stream run's bit for bit, and the next v -> f half waits for both streams"""
    import torch
    from lhvi import synth
    from lhvi.pbp import EPBP
    runs = []
    for overlap in ('1', '0'):
        monkeypatch.setenv('LHVI_PBP_OVERLAP', overlap)
        flat = synth.hybrid_mrf_flat(V=30000, deg=4, seed=8, frac_discrete=0.3)
        bp = EPBP(None, n=64, proposal_approximation='EP', sampler='device', seed=9)
        bp.overlap_min_heavy = 1
        bp._setup(None, flat=flat)
        assert bp.n_heavy > 0 and bp.n_pair > 0
        _init(api, bp)
        for _ in range(5):
            bp.sweep(last=False)
        bp.sweep(last=True)
        torch.cuda.synchronize()
        runs.append(bp)
    a, b = runs
    assert getattr(a, '_side', None) is not None and getattr(b, '_side', None) is None
    for name in ('f2v', 'v2f', 'q_dev', 'eta', 'particles'):
        assert torch.equal(getattr(a, name), getattr(b, name)), name


def test_v2f_hub_kernel_matches_the_one_wave_path(api):
    """template variables (more than 64 incident factors: the topics of the paper-popularity model) are swept by a workgroup
    each (pbp_v2f_hub_kernel: four partial totals added in a fixed order) instead of one wavefront walking the row: the same
    messages to rounding, and deterministic"""
    import torch
    from lhvi import synth
    from lhvi.pbp import EPBP
    flat, keys = synth.paper_popularity_flat(150, 4, seed=2)
    assert np.diff(flat.var_ptr).max() > 64
    n = 16
    rng = np.random.default_rng(1)
    draws = []

    def sampler(k, fl, q):
        if k == len(draws):
            cont = fl.var_hidden & fl.var_cont
            out = np.zeros((fl.V, n))
            out[cont] = np.clip(rng.standard_normal((int(cont.sum()), n)) * np.sqrt(q[cont, 1:2]) + q[cont, 0:1], -15, 15)
            draws.append(out)
        return draws[k]
    runs = []
    for packed in (True, False, True):
        bp = EPBP(None, n=n, proposal_approximation='simple', sampler=sampler, seed=6)
        bp.packed_v2f = packed
        bp._setup(None, flat=flat)
        _init(api, bp)
        for _ in range(3):
            bp.sweep(last=False)
        runs.append(bp)
    a, b, c = runs
    assert a.v2f_lists[5] > 0
    hid = flat.var_hidden[flat.edge_var]
    live = hid[:, None] & (np.arange(n)[None, :] < a.np_host[flat.edge_var][:, None])
    np.testing.assert_allclose(a.v2f.cpu().numpy()[live], b.v2f.cpu().numpy()[live], rtol=1e-11, atol=1e-10)
    np.testing.assert_allclose(a.q_dev.cpu().numpy(), b.q_dev.cpu().numpy(), rtol=1e-10, atol=1e-12, equal_nan=True)
    assert torch.equal(a.v2f, c.v2f) and torch.equal(a.f2v, c.f2v)


@pytest.mark.gpu
@pytest.mark.parametrize('rule,lifted,points', [('simple', False, 32), ('EP', False, 32), ('simple', True, 32), ('EP', True, 32),
                                               ('simple', False, 48), ('EP', True, 80)])
def test_sliced_proposal_matches_the_one_wave_path(api, rule, lifted, points):
    """rows of more than 64 incident edges are handed to the proposal kernel in slices (lhvi_pbp_t.prop_hub): the sites are the
    same bit for bit, the proposals equal to the rounding of a differently ordered sum, and a repeat gives the same bits"""
    import torch
    from lhvi import synth
    from lhvi.pbp import EPBP
    flat, keys = synth.paper_popularity_flat(150, 4, seed=2, points=points)      # (48 / 80 points: two / one edge per pass)
    if lifted:
        flat.edge_count = 1.0 + (np.arange(flat.E) % 3).astype(np.float64)
        flat.lifted = True
    deg = np.diff(flat.var_ptr)
    assert deg.max() > 130
    n = 16
    rng = np.random.default_rng(1)
    draws = []

    def sampler(k, fl, q):
        if k == len(draws):
            cont = fl.var_hidden & fl.var_cont
            out = np.zeros((fl.V, n))
            out[cont] = np.clip(rng.standard_normal((int(cont.sum()), n)) * np.sqrt(q[cont, 1:2]) + q[cont, 0:1], -15, 15)
            draws.append(out)
        return draws[k]
    runs = []
    for sliced in (True, False, True):
        bp = EPBP(None, n=n, proposal_approximation=rule, sampler=sampler, seed=6)
        bp.sliced_proposal = sliced
        bp._setup(None, flat=flat)
        _init(api, bp)
        for _ in range(3):
            bp.sweep(last=False)
        runs.append(bp)
    a, b, c = runs
    hubs = int(((deg > 64) & flat.var_hidden & flat.var_cont).sum())
    assert a.n_prop_hub == hubs and hubs > 0 and b.n_prop_hub == 0
    assert a.n_prop_desc > b.n_prop_desc
    qa, qb = a.q_dev.cpu().numpy(), b.q_dev.cpu().numpy()
    assert np.isfinite(qa).all()
    np.testing.assert_allclose(qa, qb, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(a.eta.cpu().numpy(), b.eta.cpu().numpy(), rtol=1e-9, atol=1e-12)
    assert torch.equal(a.q_dev, c.q_dev) and torch.equal(a.eta, c.eta) and torch.equal(a.f2v, c.f2v)


@pytest.mark.gpu
@pytest.mark.parametrize('n,points,lifted', [(10, 32, False), (16, 20, False), (24, 32, False), (32, 100, False), (12, 48, True)])
def test_small_particle_kernel_matches_the_one_edge_per_wave_kernels(api, n, points, lifted):
    """with at most 16 / 32 particles on both sides of an edge, four / two edges share a wavefront (pbp_f2v_small_kernel).  Same
    messages as the heavy kernel gives for those edges (to the rounding of a differently ordered sum), whole sweeps agree, ragged
    ends of the lists included"""
    import torch
    from lhvi import synth
    from lhvi.graph import Domain
    from lhvi.pbp import EPBP
    flat = _with_domain(synth.hybrid_mrf_flat(V=3001, deg=4, seed=5, frac_discrete=0.15),
                        Domain((-10.0, 10.0), continuous=True, integral_points=np.linspace(-10, 10, points)))
    if lifted:
        flat.edge_count = 1.0 + (np.arange(flat.E) % 3).astype(np.float64)
        flat.lifted = True
    runs = []
    for small in (True, False):
        bp = EPBP(None, n=n, proposal_approximation='simple', sampler='device', seed=4)
        bp.small_f2v = small
        bp._setup(None, flat=flat)
        _init(api, bp)
        for _ in range(3):
            bp.sweep(last=False)
        runs.append(bp)
    a, b = runs
    assert (a.n_small16 if n <= 16 else a.n_small32) > 1000 and b.n_small16 == 0 and b.n_small32 == 0
    heavy_ok = n + points <= 128                      # (beyond two rounds of output points the general kernel serves the edge instead)
    assert (a.n_small16 + a.n_small32 + a.n_heavy == b.n_heavy) == heavy_ok
    hid = flat.var_hidden[flat.edge_var]
    S = n + a.T
    cols = np.concatenate([np.arange(n), n + np.arange(points)])
    fa, fb = a.f2v.cpu().numpy()[hid][:, cols], b.f2v.cpu().numpy()[hid][:, cols]
    assert np.isfinite(fa).all()
    np.testing.assert_allclose(fa, fb, rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(a.q_dev.cpu().numpy(), b.q_dev.cpu().numpy(), rtol=1e-10, atol=1e-12, equal_nan=True)
    # one launch on the same inputs: the small lists' edges against the heavy kernel, message by message
    l, st = api.lib(), api.stream_ptr()
    out_a, out_b = torch.zeros_like(a.f2v), torch.zeros_like(a.f2v)
    sa = a._struct()
    api.check(l.lhvi_pbp_f2v(a.dg.g, a.dg.p, sa, api.ptr(a.v2f), api.ptr(out_a), st))
    if not heavy_ok:
        return
    sb = a._struct()
    both = torch.cat([a.heavy_desc, a.small16_desc, a.small32_desc]).contiguous()
    sb.heavy_desc, sb.n_heavy = api.ptr(both), int(both.shape[0])
    sb.small16_desc, sb.n_small16, sb.small32_desc, sb.n_small32 = None, 0, None, 0
    api.check(l.lhvi_pbp_f2v(a.dg.g, a.dg.p, sb, api.ptr(a.v2f), api.ptr(out_b), st))
    oa, ob = out_a.cpu().numpy()[hid][:, cols], out_b.cpu().numpy()[hid][:, cols]
    np.testing.assert_allclose(oa, ob, rtol=1e-11, atol=1e-11)


@pytest.mark.parametrize('name', C2F_CASES)
def test_coarse_to_fine_on_arrays_equals_the_object_path(api, golden_dir, name):
    """``HybridLBP.run(c2f >= 0)`` goes through arrays (``lhvi.c2f.run_c2f_flat``: colours, refinement and both re-liftings of
    every sweep on the device); ``c2f_on_objects`` keeps the path through Python objects per cluster.  Same colour arrays at
    every draw and identical tables at the end -- and ``on_flat(ground).run_flat(c2f)`` is the same run without any object."""
    import torch
    from lhvi.flat import flatten
    from lhvi.pbp import HybridLBP
    z, meta = load_npz(golden_dir, name)
    g, rvs, factors = modelio.load_model(meta['model'], API)
    samples = z['samples']
    inject = lambda k, flat, q: samples[k][flat.rep_ground]
    runs = []
    for on_objects in (True, False):
        bp = HybridLBP(g, n=meta['n'], proposal_approximation=meta['approx'], sampler=inject)
        bp.c2f_on_objects = on_objects
        bp.run(meta['iterations'], c2f=meta['c2f'])
        runs.append(bp)
    a, b = runs
    for (ra, fa), (rb, fb) in zip(a.c2f_history, b.c2f_history):
        assert (ra == rb).all() and (fa == fb).all()
    for name_ in ('f2v', 'v2f', 'eta', 'q_dev', 'particles', 'old_particles', 'uniq'):
        assert torch.equal(getattr(a, name_), getattr(b, name_)), name_
    hid = [i for i, rv in enumerate(rvs) if rv.value is None]
    # (b ran last: rv.cluster / f.cluster point at ITS cluster objects -- SURVEY quirk 6 -- so only b can be queried)
    qb = b.belief_rv_batch([rvs[i] for i in hid], z['query_x'][hid])
    np.testing.assert_allclose(qb, z['query_logb'][hid], rtol=1e-8, atol=1e-6)
    # no objects at all
    gflat = flatten(g, require_device_potentials=True)
    c = HybridLBP.on_flat(gflat, n=meta['n'], proposal_approximation=meta['approx'], sampler=inject)
    c.run_flat(meta['iterations'], c2f=meta['c2f'])
    for name_ in ('f2v', 'v2f', 'eta', 'q_dev', 'particles'):
        assert torch.equal(getattr(b, name_), getattr(c, name_)), name_
    qc = c.belief_rv_ground(hid, z['query_x'][hid]).cpu().numpy()
    np.testing.assert_allclose(qc, qb, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize('seed', [5, 6, 13])
def test_coarse_to_fine_on_random_instances_arrays_objects_and_flat_agree_bit_for_bit(api, seed):
    """random RGM instances (template sizes, evidence with tied and distinct values, thresholds, k-means settings; the seeds are
    three of those a soak of 40 found: evidence clusters of three and more members, whose value is a running sum, and runs whose last
    sweep keeps the factor-side state) through arrays, through objects and without objects (``tests/soak/soak_c2f_random.py``)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'tests', 'soak', 'soak_c2f_random.py'), str(seed), '1'], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and '1 of 1 seeds pass' in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]


def test_coarse_to_fine_with_the_device_sampler_listed_draws_change_nothing(api, golden_dir):
    """a coarse-to-fine state is rebuilt every sweep with the buffers of the previous one handed in: its first device draw may
    not touch `old_particles` (the particles the v -> f messages were evaluated at).  The listed draw (hidden continuous
    variables only) against the full one: bit for bit; and the f -> v tables against the C oracle on the final lifted graph"""
    import torch
    from lhvi.flat import flatten
    from lhvi.pbp import HybridLBP
    z, meta = load_npz(golden_dir, 'hlbp_c2f_rgm')
    g, rvs, factors = modelio.load_model(meta['model'], API)
    gflat = flatten(g, require_device_potentials=True)
    runs = []
    for listed in (True, False):
        bp = HybridLBP.on_flat(gflat, n=meta['n'], proposal_approximation=meta['approx'], sampler='device', seed=3)
        bp.listed_resample = listed
        bp.run_flat(4, c2f=0)
        runs.append(bp)
    a, b = runs
    for name_ in ('f2v', 'v2f', 'eta', 'q_dev', 'particles', 'old_particles', 'uniq'):
        assert torch.equal(getattr(a, name_), getattr(b, name_)), name_
    # (the full draw writes `particles` only, so equality with it shows the listed one left `old_particles` alone); the final
    # state's messages at query points against the C oracle on the same lifted graph, particles and v -> f tables
    from oracle import oracle
    o = oracle.PbpOracle(a.flat, meta['n'], ep=meta['approx'] == 'EP', epbp=False, var_threshold=5)
    o.set_particles(a.particles.cpu().numpy())
    o.v2f = a.v2f.cpu().numpy().copy()
    lf = a.flat
    edges = np.flatnonzero(lf.var_hidden[lf.edge_var] & (lf.edge_canon == np.arange(lf.E)))
    x = np.tile(np.linspace(-8, 8, 7), (edges.size, 1))
    got = torch.empty(edges.size, 7, dtype=torch.float64, device=a.v2f.device)
    api.check(api.lib().lhvi_pbp_edge_points(a.dg.g, a.dg.p, a._struct(), api.ptr(a.v2f), int(edges.size),
                                             api.ptr(api.to_dev(edges.astype(np.int32))), 7, api.ptr(api.to_dev(x)), api.ptr(got),
                                             api.stream_ptr()))
    np.testing.assert_allclose(got.cpu().numpy(), o.edge_points(edges, x), rtol=1e-9, atol=1e-8)


def test_first_device_draw_after_host_draws_keeps_the_old_particles(api):
    """mixed samplers: two host draws, then the device sampler -- the listed form's first call fills the static rows of the
    other buffer but leaves its particles (the previous sample) alone"""
    import torch
    from lhvi import synth
    from lhvi.pbp import EPBP
    flat = synth.hybrid_mrf_flat(V=600, deg=4, seed=4)
    rng = np.random.default_rng(0)
    host = [rng.uniform(-5, 5, (flat.V, 16)) for _ in range(2)]
    bp = EPBP(None, n=16, proposal_approximation='simple', sampler=lambda k, f, q: host[k], seed=1)
    bp._setup(None, flat=flat)
    _init(api, bp)
    bp.sweep(last=False)                 # second host draw
    prev = bp.particles.clone()
    bp.sampler = 'device'
    bp._generate_sample()
    cont = torch.from_numpy(flat.var_hidden & flat.var_cont).to(prev.device)
    assert torch.equal(bp.old_particles[cont], prev[cont])          # (rows of observed variables are never read)
    assert not torch.equal(bp.particles[cont], prev[cont])
    disc = torch.from_numpy(flat.var_hidden & ~flat.var_cont).to(prev.device)
    assert torch.equal(bp.particles[disc][:, :2], prev[disc][:, :2]) and torch.equal(bp.old_particles[disc][:, :2], prev[disc][:, :2])
