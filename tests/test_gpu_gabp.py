"""GPU parity: Gaussian sweep (GaBP / GaLBP) and colour refinement through the C ABI vs the oracle and the
golden vectors captured from the reference."""
import json
import os

import numpy as np
import pytest

import modelio
from test_oracle_golden import API, load, nan_equal, _initial

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def api():
    from lhvi import _abi
    _abi.require_gpu()
    return _abi


@pytest.mark.parametrize('name', ['gauss_g1_chain', 'gauss_g2_kalman', 'gauss_g3_rgm0'])
def test_gabp_matches_reference_golden(api, golden_dir, name):
    from lhvi.gabp import GaBP
    rec = load(golden_dir, name)
    g, rvs, factors = modelio.load_model(rec['model'], API)
    for k, want in rec['sweeps'].items():
        bp = GaBP(g)
        bp.run(int(k))
        hidden_edge = bp.flat.var_hidden[bp.flat.edge_var]
        # tolerance: fp64, contraction off, same summation order as CPython -> expect bit-identical;
        # 1e-13 relative leaves room for the device's division / reciprocal rounding
        nan_equal(bp._v2f, want['v2f'], rtol=1e-13)
        nan_equal(bp._f2v[hidden_edge], np.asarray(want['f2v'], dtype=float)[hidden_edge], rtol=1e-13)
        nan_equal(bp._mu_var, want['mu_var'], rtol=1e-13)
        # dict view keeps the reference's key/None conventions
        f, rv = factors[0], factors[0].nb[0]
        assert (f, rv) in bp.message and (rv, f) in bp.message
    for rv, (mu, var) in zip(rvs, rec['sweeps'][k]['mu_var']):
        if rv.value is None:
            assert bp.map(rv) == pytest.approx(mu, rel=1e-13)
            assert bp.get_belief_params(rv)[1] == pytest.approx(var, rel=1e-13)
        else:
            assert bp.map(rv) == rv.value
            assert bp.belief(rv.value, rv) == 1 and bp.belief(rv.value + 1, rv) == 0


def test_gabp_matches_oracle_random_graph(api):
    """larger random pairwise Gaussian MRF, every potential kind, vs the C oracle"""
    from lhvi import synth, _abi
    from oracle import oracle
    flat = synth.random_gaussian_mrf(V=20000, deg=4, seed=3)
    dg = _abi.DeviceGraph(flat)
    f2v, v2f, mv = dg.empty(flat.E, 2), dg.empty(flat.E, 2), dg.empty(flat.V, 2)
    l = _abi.lib()
    _abi.check(l.lhvi_gabp_run(dg.g, dg.p, _abi.ptr(f2v), _abi.ptr(v2f), 8, _abi.stream_ptr()))
    _abi.check(l.lhvi_gabp_marginals(dg.g, _abi.ptr(f2v), _abi.ptr(mv), _abi.stream_ptr()))
    of2v, ov2f, omv = oracle.gabp_run(flat, 8)
    hidden_edge = flat.var_hidden[flat.edge_var]
    nan_equal(v2f.cpu().numpy(), ov2f, rtol=1e-12)
    nan_equal(f2v.cpu().numpy()[hidden_edge], of2v[hidden_edge], rtol=1e-12)
    nan_equal(mv.cpu().numpy(), omv, rtol=1e-12)


def test_color_refinement_matches_reference_partitions(api, golden_dir):
    from lhvi import lifting
    from oracle import oracle
    rec = load(golden_dir, 'color_partitions')
    for name, entry in rec.items():
        g, rvs, factors = modelio.load_model(entry['model'], API)
        cg = lifting.CompressedGraph(g).run()
        rv_color, f_color = cg.colors()
        # partitions are integer objects: bit-exact
        assert oracle.canonical_labels(rv_color) == entry['rv_label'], name
        assert oracle.canonical_labels(f_color) == entry['f_label'], name
        assert len(cg.rvs) == entry['n_rv'] and len(cg.factors) == entry['n_f']
        index = {id(rv): i for i, rv in enumerate(rvs)}
        findex = {id(f): i for i, f in enumerate(factors)}
        for c in cg.rvs:
            key = str(min(index[id(r)] for r in c.rvs))
            got = sorted([min(findex[id(f)] for f in sf.factors), int(n)] for sf, n in c.count.items())
            assert got == entry['counts'][key], name
            assert c.N == entry['N'][key]
            want = entry['value'][key]
            assert (c.value is None) == (want is None)
            if want is not None:
                assert c.value == pytest.approx(want, rel=1e-15)


@pytest.mark.parametrize('name', ['gauss_g1_chain', 'gauss_g2_kalman', 'gauss_g3_rgm0'])
def test_galbp_matches_reference_golden(api, golden_dir, name):
    from lhvi.gabp import GaLBP
    from oracle import oracle
    rec = load(golden_dir, name)
    g, rvs, factors = modelio.load_model(rec['model'], API)
    lbp = GaLBP(g)
    lbp.run(rec['galbp']['iterations'])
    rv_color, f_color = lbp.g.colors()
    assert oracle.canonical_labels(rv_color) == rec['galbp']['rv_label']
    assert oracle.canonical_labels(f_color) == rec['galbp']['f_label']
    got = np.array([lbp.map(rv) for rv in rvs])
    # lifted summation order is a set-iteration artefact in the reference: a few ulp
    np.testing.assert_allclose(got, rec['galbp']['map'], rtol=1e-12, atol=1e-13)


def test_color_refinement_large_random_vs_oracle(api):
    """colour passing on a graph with structure (RGM template) at a size the python oracle still finishes; the template
    variables have 90 / 71 incident factors, i.e. they take the wavefront-per-row path of the fingerprint sums"""
    from lhvi import synth, lifting, _abi
    from oracle import oracle
    flat, sym, rv0, f0 = synth.rgm_flat(C=90, B=70, n_values=4, evidence_ratio=0.2, seed=1)
    assert np.diff(flat.var_ptr).max() > 64
    rv_color, f_color = lifting.refine_flat(flat, sym, rv0, f0)
    orv, of = oracle.color_passing(flat, sym, rv0, f0)
    assert oracle.canonical_labels(rv_color) == oracle.canonical_labels(orv)
    assert oracle.canonical_labels(f_color) == oracle.canonical_labels(of)


def test_color_refinement_hash_table_equals_the_radix_sort(api):
    """the two relabelling methods of lhvi_color_refine_* (fingerprints into a hash table and only the distinct keys sorted /
    radix sort of every item's fingerprint) give the same colour ARRAYS, not just the same partition; a graph with more
    distinct colours than the table holds falls back to the sort inside refine_flat and still agrees"""
    from lhvi import synth, lifting, _abi
    flat, sym, rv0, f0 = synth.rgm_structured_flat(400, 250)          # the 1/25 twin of cfg 5: 9 956 rv clusters
    sa, sb = {}, {}
    ra, fa = lifting.refine_flat(flat, sym, rv0, f0, method=_abi.COLOR_HASH, stats=sa)
    rb, fb = lifting.refine_flat(flat, sym, rv0, f0, method=_abi.COLOR_SORT, stats=sb)
    np.testing.assert_array_equal(ra, rb)
    np.testing.assert_array_equal(fa, fb)
    assert sa['rounds'] == sb['rounds'] and sa['sorted_half_rounds'] == 0 and int(ra.max()) + 1 < flat.V
    # every variable its own initial colour: 600 k distinct keys > 512 k table entries
    big = synth.random_gaussian_mrf(V=600_000, deg=4, seed=3)
    rv1 = np.arange(big.V, dtype=np.int32)
    f1 = np.zeros(big.F, dtype=np.int32)
    symb = np.zeros(big.F, dtype=np.uint8)
    st = {}
    r1, g1 = lifting.refine_flat(big, symb, rv1, f1, stats=st)
    r2, g2 = lifting.refine_flat(big, symb, rv1, f1, method=_abi.COLOR_SORT)
    assert st['sorted_half_rounds'] > 0 and int(r1.max()) + 1 == big.V
    np.testing.assert_array_equal(r1, r2)
    np.testing.assert_array_equal(g1, g2)


def test_flat_grounded_rgm_matches_object_path(api):
    """RelationalGraph.ground_flat -> GaBP on the arrays, against ground_graph -> GaBP on the objects (different rv /
    factor order because the object path keeps sets; same marginals)"""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from test_host_api import _rgm_relational
    from lhvi.gabp import GaBP
    rng = np.random.default_rng(3)
    rel_o, rel_f = _rgm_relational(40, 25), _rgm_relational(40, 25)
    g, table = rel_o.ground_graph()
    keys_all = list(table)
    ev = {keys_all[i]: float(rng.uniform(-30, 30)) for i in rng.choice(len(keys_all), 60, replace=False)}
    rel_o.add_evidence(ev)
    bo = GaBP(g)
    bo.run(15)
    flat, keys = rel_f.ground_flat(ev)
    bf = GaBP(flat)
    bf.run(15)
    checked = 0
    for key, rv in table.items():
        if rv.value is None:
            mo = bo.get_belief_params(rv)
            mf = bf.mu_var[keys.var_id(key)]
            np.testing.assert_allclose(mf, mo, rtol=1e-10, atol=1e-12)
            checked += 1
    assert checked == len(keys_all) - 60


def test_relational_pipeline_on_arrays_matches_object_lifting(api):
    """ground_flat -> initial_colors_flat -> refine_flat -> lift_flat -> lifted Gaussian sweep, against the object path
    (ground_graph -> GaLBP, which lifts with CompressedGraph): same number of clusters, same marginals"""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from test_host_api import _rgm_relational
    from lhvi import lifting
    from lhvi.gabp import GaLBP
    rng = np.random.default_rng(8)
    rel_o, rel_f = _rgm_relational(30, 20), _rgm_relational(30, 20)
    g, table = rel_o.ground_graph()
    keys_all = list(table)
    pool = [-3.0, 0.5, 2.0]                                   # few distinct evidence values -> real lifting
    ev = {keys_all[i]: pool[int(rng.integers(3))] for i in rng.choice(len(keys_all), 80, replace=False)}
    rel_o.add_evidence(ev)
    lbp = GaLBP(g)
    lbp.run(12)
    flat, keys = rel_f.ground_flat(ev)
    rv0, f0, sym = lifting.initial_colors_flat(flat)
    rvc, fc = lifting.refine_flat(flat, sym, rv0, f0)
    orv, of = lbp.g.colors()
    assert int(rvc.max()) + 1 == int(np.max(orv)) + 1 and int(fc.max()) + 1 == int(np.max(of)) + 1
    lflat = lifting.lift_flat(flat, rvc, fc)
    dg = api.DeviceGraph(lflat)
    f2v, v2f, mv = dg.empty(lflat.E, 2), dg.empty(lflat.E, 2), dg.empty(lflat.V, 2)
    l, st = api.lib(), api.stream_ptr()
    api.check(l.lhvi_gabp_run(dg.g, dg.p, api.ptr(f2v), api.ptr(v2f), 12, st))
    api.check(l.lhvi_gabp_marginals(dg.g, api.ptr(f2v), api.ptr(mv), st))
    mv = mv.cpu().numpy()
    for key, rv in table.items():
        if rv.value is None:
            np.testing.assert_allclose(mv[rvc[keys.var_id(key)], 0], lbp.map(rv), rtol=1e-9, atol=1e-10)


def _run_both_forms(api, flat, iterations):
    """(f2v, v2f, marginals) of lhvi_gabp_run (v2f / f2v kernel pair) and of lhvi_gabp_run_pull (one launch per sweep)"""
    import torch
    from lhvi.gabp import pull_plan
    dg = api.DeviceGraph(flat)
    l, st = api.lib(), api.stream_ptr()
    out = []
    host = pull_plan(flat)
    dev = {k: (api.to_dev(a) if a is not None else None) for k, a in host.items()}
    plan = api.GabpPlanStruct()
    plan.pslot, plan.info, plan.count = (api.ptr(dev[k]) for k in ('pslot', 'info', 'count'))
    plan.n_hub_rows = int((np.diff(flat.var_ptr) > 512).sum())
    nbytes = int(l.lhvi_gabp_pull_workspace_bytes(dg.g))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dg.device)
    for pull in (False, True, 'records', 'records, potentials from global memory'):
        f2v, v2f, mv = dg.empty(flat.E, 2), dg.empty(flat.E, 2), dg.empty(flat.V, 2)
        plan.rec = api.ptr(dev['rec']) if isinstance(pull, str) else None      # round 4: one 16-byte record per slot
        plan.pot_words = api.ptr(dev['pot_words'])
        plan.seg, plan.n_seg = api.ptr(dev['seg']), int(host['seg'].shape[0])
        if pull == 'records, potentials from global memory':
            # more potentials than the kernel keeps in LDS: pad the table (the padding rows are never referenced)
            import copy
            wide = copy.copy(flat)
            wide.__dict__.pop('_view_cache', None)
            wide.pot_kind = np.concatenate([flat.pot_kind, np.full(64, flat.pot_kind[0], dtype=flat.pot_kind.dtype)])
            wide.pot_off = np.concatenate([flat.pot_off, np.full(64, flat.pot_off[-1], dtype=flat.pot_off.dtype)])
            dgw = api.DeviceGraph(wide)
            wide_words = api.to_dev(pull_plan(wide)['pot_words'])
            plan.pot_words = api.ptr(wide_words)
            api.check(l.lhvi_gabp_run_pull(dgw.g, dgw.p, plan, api.ptr(f2v), api.ptr(v2f), iterations, api.ptr(ws), nbytes, st))
        elif pull:
            api.check(l.lhvi_gabp_run_pull(dg.g, dg.p, plan, api.ptr(f2v), api.ptr(v2f), iterations, api.ptr(ws), nbytes, st))
        else:
            api.check(l.lhvi_gabp_run(dg.g, dg.p, api.ptr(f2v), api.ptr(v2f), iterations, st))
        api.check(l.lhvi_gabp_marginals(dg.g, api.ptr(f2v), api.ptr(mv), st))
        out.append((f2v.cpu().numpy(), v2f.cpu().numpy(), mv.cpu().numpy()))
    return out


def _ragged_gaussian_flat(seed=0, n_small=6000):
    """every row class of the pull kernel in one graph: thousands of rows of 2-6 entries, rows of 33-480 entries (chunk sums), two
    rows of more than 512 (wave-parallel hub kernel), evidence in each class -- a star forest (each big variable tied to its own
    leaves by linear-Gaussian / XY / Gaussian factors, leaves tied among themselves at random, a unary prior everywhere)"""
    from lhvi import potentials as P
    from lhvi.flat import build_flat
    from lhvi.graph import Domain
    rng = np.random.default_rng(seed)
    big = [700, 1500] + rng.integers(33, 480, 40).tolist() + [32, 33, 64, 65, 511, 512, 513]
    pairs, pot = [], []
    nv = len(big)
    for c, d in enumerate(big):
        leaves = np.arange(nv, nv + d - 1)
        nv += d - 1
        for lf in leaves:
            pairs.append((c, lf) if rng.random() < 0.5 else (lf, c))
            pot.append(int(rng.integers(0, 3)))
    first_small = nv
    nv += n_small
    for _ in range(2 * n_small):
        a, b = rng.integers(len(big), nv, 2)
        if a != b:
            pairs.append((int(a), int(b)))
            pot.append(int(rng.integers(0, 3)))
    F2 = len(pairs)
    edge_var = np.concatenate([np.array(pairs, dtype=np.int32).ravel(), np.arange(nv, dtype=np.int32)])
    fac_ptr = np.concatenate([np.arange(0, 2 * F2 + 1, 2), 2 * F2 + np.arange(1, nv + 1)]).astype(np.int32)
    fac_pot = np.concatenate([np.array(pot), np.full(nv, 3)]).astype(np.int32)
    value = np.full(nv, np.nan)
    obs = rng.random(nv) < 0.1
    obs[[0, 5, 44]] = True                                   # a hub row, two chunked rows
    value[obs] = np.round(rng.uniform(-2, 2, int(obs.sum())), 3)
    dom = Domain((-10, 10), continuous=True, integral_points=np.linspace(-10, 10, 8))
    pots = [(P.POT_LINEAR_GAUSSIAN, [0.7, 2.0]), (P.POT_XY, [0.3, 3.0]), (P.POT_GAUSSIAN, P.GaussianPotential([0.0, 0.0], [[2.0, 0.5], [0.5, 1.5]]).device_spec([dom, dom])[1]),
            (P.POT_X2, [1.0, 4.0])]
    return build_flat(fac_ptr, edge_var, fac_pot, pots, value, np.zeros(nv, dtype=np.int32), [dom])


def test_pull_form_on_a_graph_with_every_row_class(api):
    """rows of 2-6, of 33-512 and of more than 512 entries in ONE graph with evidence in each class: after two sweeps (the first
    one sums equal initial messages: exact either way) the slots of rows of at most 32 and of more than 512 entries carry the bits
    of the kernel pair and the chunk-summed rows agree to 1e-13; after seven sweeps everything agrees to 1e-11 (the rounding of the
    chunked rows has travelled)"""
    flat = _ragged_gaussian_flat()
    deg = np.diff(flat.var_ptr)
    assert (deg > 512).sum() >= 2 and ((deg > 32) & (deg <= 512)).sum() >= 40 and (deg <= 32).sum() > 5000
    chunked = (deg > 32) & (deg <= 512)
    (f_a, v_a, m_a), _, (f_b, v_b, m_b), (f_c, v_c, m_c) = _run_both_forms(api, flat, 2)
    for v_x in (v_b, v_c):
        same = ~chunked[flat.edge_var]
        assert v_a[same].tobytes() == v_x[same].tobytes()
        np.testing.assert_allclose(v_a, v_x, rtol=1e-13, atol=1e-13, equal_nan=True)
        assert v_a[~same].tobytes() != v_x[~same].tobytes()          # (the chunked sums did run)
    (f_a, v_a, m_a), *others = _run_both_forms(api, flat, 7)
    for f_x, v_x, m_x in others[1:]:
        for a, b in ((f_a, f_x), (v_a, v_x), (m_a, m_x)):
            np.testing.assert_allclose(a, b, rtol=1e-11, atol=1e-11, equal_nan=True)
    assert np.isfinite(m_a[flat.var_hidden]).all()


@pytest.mark.parametrize('case', ['random', 'lifted_rgm', 'hub', 'one_sweep', 'no_sweep', 'long_rows'])
def test_pull_form_equals_the_kernel_pair_bit_for_bit(api, case):
    """lhvi_gabp_run_pull (messages in slot order, f -> v recomputed from the partner's v -> f, one launch per sweep) against
    lhvi_gabp_run: same expressions in the same order, so every message and marginal has the same bits -- ground graph
    with every potential kind and evidence, a lifted RGM (counts), a 700-edge hub (wave-parallel path), 1 and 0 sweeps"""
    from lhvi import synth, lifting
    its = 7
    if case == 'long_rows':
        flat = synth.rgm_flat(C=60, B=40, n_values=0, evidence_ratio=0.15, seed=2)[0]
        assert np.diff(flat.var_ptr).max() == 60
    elif case == 'lifted_rgm':
        g, sym, rv0, f0 = synth.rgm_structured_flat(80, 50, A=40, R=25)
        rvc, fc = lifting.refine_flat(g, sym, rv0, f0)
        flat = lifting.lift_flat(g, rvc, fc)
        assert flat.lifted and flat.V < g.V
    elif case == 'hub':
        from lhvi.flat import build_flat
        from lhvi.graph import Domain
        from lhvi import potentials as P
        D = 700
        dom = Domain((-10, 10), continuous=True, integral_points=np.linspace(-10, 10, 8))
        edge_var = np.concatenate([np.stack([np.zeros(D, dtype=np.int32), np.arange(1, D + 1, dtype=np.int32)], axis=1).ravel(),
                                   np.arange(D + 1, dtype=np.int32)])
        fac_ptr = np.concatenate([np.arange(0, 2 * D + 1, 2), 2 * D + np.arange(1, D + 2)]).astype(np.int32)
        value = np.full(D + 1, np.nan)
        value[5::7] = 1.0
        flat = build_flat(fac_ptr, edge_var, np.concatenate([np.zeros(D), np.ones(D + 1)]).astype(np.int32),
                          [(P.POT_LINEAR_GAUSSIAN, [0.7, 2.0]), (P.POT_X2, [1.0, 4.0])], value, np.zeros(D + 1, dtype=np.int32), [dom])
    else:
        flat = synth.random_gaussian_mrf(V=30000, deg=4, seed=5)
        its = {'one_sweep': 1, 'no_sweep': 0}.get(case, its)
    (f_a, v_a, m_a), *others = _run_both_forms(api, flat, its)
    assert len(others) == 3                     # pull form on the graph arrays, on slot records (potentials in LDS / in global memory)
    for which, (f_b, v_b, m_b) in enumerate(others):
        for a, b in ((f_a, f_b), (v_a, v_b), (m_a, m_b)):
            if case == 'long_rows' and which > 0:
                np.testing.assert_allclose(a, b, rtol=1e-13, atol=1e-13, equal_nan=True)
                assert a.tobytes() != b.tobytes() or a.size == 0        # (the chunked sums did run)
            else:
                assert a.tobytes() == b.tobytes()
    assert np.isfinite(m_a[flat.var_hidden]).all() or case == 'no_sweep'


def test_recorded_run_equals_the_direct_run_and_follows_new_evidence(api):
    """GaBP.run on a small graph records its launches once (lhvi_gabp_graph_create) and replays them: same bits as the direct
    call sequence; a second run() on the same solver after the evidence VALUES changed reuses the device state and the recorded
    graph and equals a fresh solver; a changed hidden / observed pattern rebuilds"""
    from lhvi import synth
    from lhvi.gabp import GaBP
    flat, sym, rv0, f0 = synth.rgm_flat(C=60, B=40, n_values=3, evidence_ratio=0.2, seed=4)
    a = GaBP(flat)
    a.run(12)
    b = GaBP(flat)
    b.graph_replay_slots = 0
    b.run(12)
    assert len(a._state['graphs']) == 1 and not b._state['graphs']
    np.testing.assert_array_equal(a._f2v, b._f2v)
    np.testing.assert_array_equal(a._v2f, b._v2f)
    np.testing.assert_array_equal(a._mu_var, b._mu_var)
    # new evidence values, same pattern
    import copy
    flat2 = copy.copy(flat)
    flat2.__dict__.pop('_view_cache', None)
    flat2.var_value = np.where(np.isnan(flat.var_value), np.nan, flat.var_value * 0.5 + 1.0)
    state = a._state
    a.g = flat2
    a.run(12)
    assert a._state is state and len(state['graphs']) == 1
    c = GaBP(flat2)
    c.graph_replay_slots = 0
    c.run(12)
    np.testing.assert_array_equal(a._mu_var, c._mu_var)
    np.testing.assert_array_equal(a._f2v, c._f2v)
    # another pattern: everything rebuilt
    flat3 = copy.copy(flat)
    flat3.__dict__.pop('_view_cache', None)
    v = flat.var_value.copy()
    v[np.flatnonzero(np.isnan(v))[5]] = 0.25
    flat3.var_value = v
    a.g = flat3
    a.run(12)
    assert a._state is not state
    d = GaBP(flat3)
    d.graph_replay_slots = 0
    d.run(12)
    np.testing.assert_array_equal(a._mu_var, d._mu_var)
