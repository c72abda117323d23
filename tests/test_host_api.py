"""CPU suite: host-side object model, potentials, MLN tracing, relational grounding, Kalman builder, flattening."""
import importlib
import json
import os
import sys

import numpy as np
import pytest

from lhvi import expr, graph as G, kalman, mln as M, potentials as P, relational as R
from lhvi.flat import flatten
import modelio
from test_oracle_golden import API, load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_domain_defaults_and_identity():
    d = G.Domain((-1, 1), continuous=True)
    assert len(d.integral_points) == 30 and d.integral_points[0] == -1 and d.integral_points[-1] == 1
    assert G.Domain((0, 1)) != G.Domain((0, 1))          # identity, like the reference


def test_graph_init_nb_orders_by_factor_id():
    d = G.Domain((0, 1))
    a, b = G.RV(d), G.RV(d)
    t = P.TablePotential(np.ones((2, 2)))
    f1, f2, f3 = G.F(t, [a, b]), G.F(t, [b, a]), G.F(t, [a, b])
    g = G.Graph()
    g.rvs, g.factors = {a, b}, {f3, f1, f2}
    g.init_nb()
    assert a.nb == [f1, f2, f3] and a.N == 3 and g.rvs_list == [a, b]
    g.init_rv_indices()
    assert g.Nd == 2 and f1.disc_nb_idx == (0, 1) and f2.disc_nb_idx == (1, 0)


def test_potential_values_and_equality():
    assert P.LinearGaussianPotential(0.5, 2.0) == P.LinearGaussianPotential(0.5, 2.0)
    assert P.XYPotential(0.5, 2.0) != P.X2Potential(0.5, 2.0)
    assert hash(P.X2Potential(1, 4)) == hash(P.X2Potential(1, 4))
    g1, g2 = P.GaussianPotential([0, 0], [[2, 0], [0, 2]]), P.GaussianPotential([0, 0], [[2, 0], [0, 2]])
    assert g1 != g2                                       # identity for everything else
    assert P.LinearGaussianPotential(0.9, 1.0).get((1.0, 1.5)) == pytest.approx(np.exp(-(1.5 - 0.9) ** 2 / 2))
    assert P.XYPotential(0.4, 2.0).symmetric and not P.X2Potential(1, 1).symmetric
    q = P.QuadraticPotential(np.array([[-1.0, 0.2], [0.2, -0.5]]), np.array([0.1, 0.3]), 0.2)
    x = np.array([0.3, -0.7])
    assert q.get(x) == pytest.approx(np.exp(x @ q.A @ x + q.b @ x + 0.2))
    hq = P.HybridQuadraticPotential(np.array([[[-0.2]], [[-0.5]]]), np.array([[0.5], [-0.4]]), np.array([0.0, 0.3]))
    assert hq.get((1, 2.0)) == pytest.approx(np.exp(-0.5 * 4 - 0.4 * 2 + 0.3))
    assert g1.get((1.0, 1.0)) == pytest.approx(np.exp(-0.5))
    A, b, c = g1.get_quadratic_params()
    assert np.allclose(A, -0.25 * np.eye(2)) and np.allclose(b, 0) and c == 0


def test_mln_ops_and_tracing():
    assert M.imp_op(1, 0) == 0 and M.imp_op(0, 0) == 1 and M.bic_op(1, 1) == 1 and M.eq_op(2, 5) == -9
    pot = M.MLNPotential(lambda x: x[0] * M.eq_op(x[1], x[2]), w=0.5)
    assert pot.get((1, 2.0, 3.5)) == pytest.approx(np.e ** (-2.25 * 0.5))
    rng = np.random.default_rng(0)
    for name, f in modelio.FORMULAS.items():
        arity = 5 if name == 'rm_aligned' else 3 if name in ('x0_eq12', 'any3') else 2 if name in ('nand',) or name.startswith('x0_eq') else 1
        prog = expr.trace(f, arity)
        for _ in range(20):
            x = [float(rng.integers(0, 2)) if name in ('nand', 'any3', 'rm_aligned') else float(rng.uniform(-3, 3)) for _ in range(arity)]
            assert expr.run(prog, x) == pytest.approx(float(f(x)), rel=1e-15, abs=1e-15)
    with pytest.raises(expr.FormulaNotTraceable):
        expr.trace(lambda x: 1 if x[0] > 0 else 0, 1)
    hard = M.MLNHardPotential(lambda x: x[0] - 0.5)
    assert hard.get((1,)) == 1 and hard.get((0,)) == 0


def test_formulas_that_branch_on_discrete_arguments_are_traced_per_state():
    """``MLNPotential`` accepts any callable (MLNPotential.py:36-37); the reference keeps ``1 if ... else 0`` variants of two
    robot-mapping formulas beside the arithmetic ones it runs (Demo/Data/HMLN/GeneratorRobotMapping.py:37,42).  A branch on the
    values of DISCRETE arguments is resolved by tracing the formula once per joint discrete state; the traced program, its
    conditional-quadratic view and the device parameter row equal those of the arithmetic variant's values on every state"""
    import itertools
    from lhvi.graph import Domain
    neg = M.neg_op
    boolean = Domain((0, 1), continuous=False)
    cont = Domain((-5, 5), continuous=True, integral_points=np.linspace(-5, 5, 8))
    cases = [
        # GeneratorRobotMapping.py:37 / :38 -- "some type holds"
        (lambda x: 1 if x[0] + x[1] + x[2] > 0 else 0, lambda x: 1 - (x[0] == 0) * (x[1] == 0) * (x[2] == 0), [boolean] * 3),
        # GeneratorRobotMapping.py:42 (its arithmetic neighbour :43 is a different clause: no equivalence to check)
        (lambda x: 1 if neg(x[0]) + neg(x[1]) + x[2] + neg(x[3]) + neg(x[4]) else 0, None, [boolean] * 5),
        # a hybrid one: the branch picks which soft equality applies
        (lambda x: M.eq_op(x[1], 0.1) if x[0] == 1 else 0.5 * M.eq_op(x[1], x[2]), None, [boolean, cont, cont]),
    ]
    for branching, arithmetic, domains in cases:
        with pytest.raises(expr.FormulaNotTraceable):
            expr.trace(branching, len(domains))
        pot = M.MLNPotential(branching, w=1.7)
        kind, par = pot.device_spec(domains)
        nops = int(par[1])
        prog = par[3:3 + 2 * nops]
        disc = [i for i, d in enumerate(domains) if not d.continuous]
        rng = np.random.default_rng(1)
        for states in itertools.product(*[domains[i].values for i in disc]):
            x = [float(rng.uniform(-3, 3)) for _ in domains]
            for i, v in zip(disc, states):
                x[i] = v
            want = float(branching(x))
            assert expr.run(prog, x) == pytest.approx(want, rel=1e-15, abs=1e-15)
            assert pot.get(x) == pytest.approx(np.e ** (1.7 * want))
            if arithmetic is not None:
                assert want == float(arithmetic(x))
        if any(d.continuous for d in domains):
            assert par[2] == 3 + 2 * nops and par[3 + 2 * nops] == expr.CQ_MAGIC          # the hybrid one is conditionally quadratic: routed to the fast kernels
        # same parameter row for the same graph position (cached per domains), and the flattening of a graph accepts it
        assert pot.device_spec(domains)[1] == par
    with pytest.raises(expr.FormulaNotTraceable):              # a branch on a CONTINUOUS value stays untraceable: fail loudly
        M.MLNPotential(lambda x: 1 if x[1] > 0 else x[0], w=1).device_spec([boolean, cont])


def test_relational_grounding_matches_rgm_template():
    d = G.Domain((-50, 50), continuous=True, integral_points=np.linspace(-50, 50, 100))
    p1, p2, p3 = (P.GaussianPotential([0., 0.], s) for s in ([[10., -7.], [-7., 10.]], [[10., 5.], [5., 10.]], [[10., 7.], [7., 10.]]))
    C, B = 6, 4
    lv_r, lv_c, lv_b = R.LV(('all',)), R.LV([f'c{i}' for i in range(C)]), R.LV([f'b{i}' for i in range(B)])
    atoms = (R.Atom(d, (lv_r,), 'recession'), R.Atom(d, (lv_b,), 'revenue'), R.Atom(d, (lv_c, lv_b), 'loss'), R.Atom(d, (lv_c,), 'market'))
    pfs = (R.ParamF(p1, nb=('recession($all)', 'market(c)')), R.ParamF(p2, nb=('market(c)', 'loss(c,b)')),
           R.ParamF(p3, nb=('loss(c,b)', 'revenue(b)')))
    rel = R.RelationalGraph(atoms, pfs)
    g, table = rel.ground_graph()
    assert len(g.rvs) == 1 + C + C * B + B and len(g.factors) == C + 2 * C * B
    assert table[('recession', 'all')].N == C and table[('market', 'c0')].N == 1 + B and table[('loss', 'c1', 'b2')].N == 2
    rel.add_evidence({('market', 'c0'): 1.5, ('loss', 'c1', 'b2'): -2.0})
    assert table[('market', 'c0')].value == 1.5 and table[('revenue', 'b0')].value is None
    # constraints and constants
    db = G.Domain((0, 1))
    lv_t = R.LV(['W', 'D', 'O'])
    lv_s = R.LV(['s1', 's2'])
    a = R.Atom(db, (lv_s, lv_t), 'SegType')
    f0 = R.ParamF(M.MLNPotential(lambda x: M.or_op(M.neg_op(x[0]), M.neg_op(x[1])), w=3), nb=['SegType(s,t1)', 'SegType(s,t2)'],
                  constrain=lambda s: s['t1'] != s['t2'])
    f3 = R.ParamF(M.MLNPotential(lambda x: x[0], w=0.3), nb=['SegType(s,$W)'])
    g2, t2 = R.RelationalGraph((a,), (f0, f3)).ground_graph()
    assert len(g2.factors) == 2 * 6 + 2 and len(g2.rvs) == 6


def test_kalman_builder_matches_reference_structure(golden_dir):
    """same builder arguments as oracle/capture_golden.py::model_kalman -> same graph as the reference built"""
    rec = load(golden_dir, 'gauss_g2_kalman')
    n, T, seed = 4, 5, 0
    rng = np.random.RandomState(seed)
    A = rng.uniform(-0.5, 0.5, size=(n, n)) + np.eye(n) * 0.5
    data = rng.uniform(-2, 2, size=(n, T))
    data[rng.rand(n, T) < 0.3] = 5000
    data[:, 0] = rng.uniform(-2, 2, size=n)
    d = G.Domain((-8, 8), continuous=True, integral_points=np.linspace(-8, 8, 32))
    g, table = kalman.KalmanFilter(d, A, 1.5, np.eye(n), 0.7).grounded_graph(T, data)
    mine = modelio.dump_model(g)
    want = rec['model']
    assert mine['rvs'] == want['rvs']
    assert [nb for _, nb in mine['factors']] == [nb for _, nb in want['factors']]
    mp = [mine['potentials'][i] for i, _ in mine['factors']]
    wp = [want['potentials'][i] for i, _ in want['factors']]
    assert mp == wp


def test_flatten_layout():
    rec_g, rvs = __import__('lhvi.synth', fromlist=['x']).gaussian_chain(5)
    flat = flatten(rec_g)
    assert flat.V == 5 and flat.F == 8 and flat.E == 12 and not flat.lifted
    for f in range(flat.F):
        for e in range(flat.fac_ptr[f], flat.fac_ptr[f + 1]):
            assert flat.edge_fac[e] == f and flat.edge_pos[e] == e - flat.fac_ptr[f]
    for v, rv in enumerate(rvs):
        fs = [flat.factors[flat.edge_fac[e]] for e in flat.var_edge[flat.var_ptr[v]:flat.var_ptr[v + 1]]]
        assert fs == rv.nb                                      # variable CSR keeps rv.nb order
    assert np.isnan(flat.var_value[1:]).all() and flat.var_value[0] == 1.5
    assert len(flat.potentials) == 2


def test_compat_modules_mirror_reference_imports():
    compat = os.path.join(ROOT, 'lifted-hybrid-variational-inference_amd', 'compat')
    sys.path.insert(0, compat)
    try:
        for name in ('Graph', 'Potential', 'MLNPotential', 'RelationalGraph', 'KalmanFilter', 'CompressedGraphWithObs',
                     'CompressedGraphSorted', 'GaBP', 'GaLBP', 'EPBPLogVersion', 'HybridLBPLogVersion', 'VarInference',
                     'LiftedVarInference', 'C2FVarInference', 'utils'):
            sys.modules.pop(name, None)
            importlib.import_module(name)
        ns = {}
        exec('from RelationalGraph import *\nfrom MLNPotential import *\nfrom Potential import GaussianPotential\n'
             'from GaBP import GaBP\nfrom EPBPLogVersion import EPBP\nfrom HybridLBPLogVersion import HybridLBP\n'
             'from VarInference import VarInference as VI\nfrom LiftedVarInference import VarInference as LVI\n'
             'from C2FVarInference import VarInference as C2FVI\n'
             'from CompressedGraphSorted import CompressedGraphSorted\n'
             'd = Domain((-1, 1), continuous=True, integral_points=linspace(-1, 1, 5))\n', ns)
        assert ns['VI'] is not ns['LVI'] and ns['d'].continuous
        assert ns['C2FVI'].update_obs_its == 10 and ns['C2FVI'].gaussian_obs is True        # C2FVI:11-18
    finally:
        sys.path.remove(compat)
        for name in ('Graph', 'Potential', 'MLNPotential', 'RelationalGraph', 'KalmanFilter', 'utils', 'GaBP', 'GaLBP',
                     'VarInference', 'LiftedVarInference', 'C2FVarInference', 'EPBPLogVersion', 'HybridLBPLogVersion',
                     'CompressedGraphWithObs', 'CompressedGraphSorted'):
            sys.modules.pop(name, None)


def test_split_evidence_colors_kmeans():
    from lhvi.lifting import split_evidence_colors
    vals = np.array([np.nan, 3.0, -2.0, 3.0, -2.0, -2.0, np.nan, 7.5])
    col = np.array([0, 1, 1, 1, 1, 1, 0, 2], dtype=np.int32)
    out = split_evidence_colors(vals, col, k=2, iteration=50, epsilon=0.0)
    assert out[0] == out[6] == 0 and out[7] == 2
    assert out[1] == out[3] and out[2] == out[4] == out[5] and out[1] != out[2]
    assert set(out.tolist()) == {0, 1, 2, 3}
    # below the threshold nothing moves; HLBP's variant compares the variance itself
    assert (split_evidence_colors(vals, col, epsilon=10.0) == col).all()
    assert (split_evidence_colors(vals, col, epsilon=6.3, use_sqrt=False) == col).all()     # var = 6.0
    assert (split_evidence_colors(vals, col, epsilon=5.9, use_sqrt=False) != col).any()
    # three groups, k = 2: the nearest-centroid rule merges the two close ones
    vals3 = np.array([0.0, 0.1, 5.0, 5.1, 0.05])
    out3 = split_evidence_colors(vals3, np.zeros(5, dtype=np.int32), k=2, iteration=10)
    assert out3[0] == out3[1] == out3[4] and out3[2] == out3[3] and out3[0] != out3[2]


def _rgm_relational(C, B):
    d = G.Domain((-50, 50), continuous=True, integral_points=np.linspace(-50, 50, 30))
    p1, p2, p3 = (P.GaussianPotential([0., 0.], s) for s in ([[10., -7.], [-7., 10.]], [[10., 5.], [5., 10.]], [[10., 7.], [7., 10.]]))
    lv_r, lv_c, lv_b = R.LV(('all',)), R.LV([f'c{i}' for i in range(C)]), R.LV([f'b{i}' for i in range(B)])
    atoms = (R.Atom(d, (lv_r,), 'recession'), R.Atom(d, (lv_b,), 'revenue'), R.Atom(d, (lv_c, lv_b), 'loss'), R.Atom(d, (lv_c,), 'market'))
    pfs = (R.ParamF(p1, nb=('recession($all)', 'market(c)')), R.ParamF(p2, nb=('market(c)', 'loss(c,b)')),
           R.ParamF(p3, nb=('loss(c,b)', 'revenue(b)')))
    return R.RelationalGraph(atoms, pfs)


def _paper_popularity_relational(n_paper, n_topic):
    """paper-popularity hybrid MLN template (atoms / formulas as in the reference's generator, SURVEY.md cfg 3)"""
    topics, papers = [f't{i}' for i in range(n_topic)], [f'p{i}' for i in range(n_paper)]
    lv_t, lv_p = R.LV(topics), R.LV(papers)
    dr = G.Domain((-15, 15), continuous=True, integral_points=np.linspace(0, 10, 12))
    db = G.Domain((0, 1))
    atoms = (R.Atom(dr, (lv_t,), 'TopicPopularity'), R.Atom(dr, (lv_p,), 'PaperPopularity'),
             R.Atom(db, (lv_p, lv_t), 'AboutTopic'), R.Atom(db, (lv_t, lv_t), 'SameSession'))
    f1 = R.ParamF(M.MLNPotential(lambda x: M.eq_op(x[0], 1.0), w=0.3), nb=('PaperPopularity(p)',))
    f2 = R.ParamF(M.MLNPotential(lambda x: x[0] * M.eq_op(x[1], x[2]), w=1.0),
                  nb=('SameSession(t1,t2)', 'TopicPopularity(t1)', 'TopicPopularity(t2)'), constrain=lambda s: s['t1'] != s['t2'])
    f3 = R.ParamF(M.MLNPotential(lambda x: x[0] * M.eq_op(x[1], x[2]), w=1.0),
                  nb=('AboutTopic(p,t)', 'PaperPopularity(p)', 'TopicPopularity(t)'))
    return R.RelationalGraph(atoms, (f1, f2, f3))


@pytest.mark.parametrize('build,evidence', [
    (lambda: _rgm_relational(6, 4), {('market', 'c0'): 1.5, ('loss', 'c1', 'b2'): -2.0, ('revenue', 'b3'): 7.0}),
    (lambda: _paper_popularity_relational(7, 4), {('AboutTopic', 'p1', 't2'): 1, ('SameSession', 't0', 't1'): 0,
                                                   ('SameSession', 't2', 't2'): 1, ('TopicPopularity', 't3'): 4.5}),
])
def test_flat_grounding_equals_object_grounding(build, evidence):
    """ground_flat() must give, array for array, what ground_graph() + add_evidence() + flatten() give when the object
    path visits rvs and factors in creation order (the reference keeps them in sets)"""
    from lhvi.flat import flatten
    rel = build()
    g, table = rel.ground_graph()
    rel.add_evidence(evidence)
    # creation order: rvs_dict insertion order, factors in grounding order (recover it from the rvs' nb lists)
    g.rvs = list(table.values())
    seen, factors = set(), []
    rel2 = build()
    g2, table2 = rel2.ground_graph()
    flat_f, keys = rel2.ground_flat(evidence)
    # rebuild the ordered factor list of the object path by grounding again with lists
    ordered = []
    for pf in rel.param_factors:
        lvs = dict()
        for e in pf.nb:
            rel.extract_lvs(e, lvs)
        for sub in rel.lvs_iter(lvs):
            if pf.constrain is None or pf.constrain(sub):
                ordered.append((pf.potential, tuple(rel.atom_substitution(rel._parse(e), sub)[0] for e in pf.nb)))
    by_scope = {}
    for f in g.factors:
        by_scope.setdefault((id(f.potential), tuple(id(rv) for rv in f.nb)), []).append(f)
    g.factors = [by_scope[(id(p), tuple(id(table[k]) for k in ks))].pop() for p, ks in ordered]
    for rv in g.rvs:
        rv.nb = []
    g.init_nb()
    flat_o = flatten(g)
    assert (flat_o.V, flat_o.F, flat_o.E) == (flat_f.V, flat_f.F, flat_f.E)
    for name in ('fac_ptr', 'edge_var', 'edge_fac', 'edge_pos', 'edge_canon', 'var_ptr', 'var_edge', 'fac_pot', 'pot_kind', 'pot_off',
                 'pot_param', 'var_dom', 'dom_cont', 'dom_lo', 'dom_hi', 'dom_ptr', 'dom_val', 'edge_count'):
        np.testing.assert_array_equal(getattr(flat_o, name), getattr(flat_f, name), err_msg=name)
    np.testing.assert_array_equal(np.isnan(flat_o.var_value), np.isnan(flat_f.var_value))
    np.testing.assert_array_equal(np.nan_to_num(flat_o.var_value), np.nan_to_num(flat_f.var_value))
    for key, rv in table.items():
        v = keys.var_id(key)
        assert g.rvs[v] is rv and keys.key_of(v) == key
    assert keys.var_id(('SameSession', 't2', 't2')) == -1 if 'SameSession' in keys.atom_ids else True


def test_flat_grounding_vectorised_constraint_and_scale():
    """a vectorised constraint gives the same arrays as the per-substitution callable; 1e5 factors ground in well under a second"""
    import time
    rel_a, rel_b = _paper_popularity_relational(9, 5), _paper_popularity_relational(9, 5)
    vec = lambda s: s['t1'] != s['t2']
    vec.vectorized = True
    rel_b.param_factors[1].constrain = vec
    fa, _ = rel_a.ground_flat()
    fb, _ = rel_b.ground_flat()
    for name in ('fac_ptr', 'edge_var', 'fac_pot', 'var_dom'):
        np.testing.assert_array_equal(getattr(fa, name), getattr(fb, name))
    big = _rgm_relational(400, 250)
    t0 = time.perf_counter()
    flat, keys = big.ground_flat({('market', 'c7'): 3.0})
    dt = time.perf_counter() - t0
    assert flat.F == 400 + 2 * 400 * 250 and flat.E == 2 * flat.F and dt < 2.0
    assert flat.var_value[keys.var_id(('market', 'c7'))] == 3.0 and np.isnan(flat.var_value).sum() == flat.V - 1


def test_flat_grounding_matches_reference_grounding(golden_dir):
    """the reference's own RGM (100 x 10) and paper-popularity (300 x 10) templates, grounded by the reference
    (oracle/capture_grounding.py), against ground_flat(): same ground rvs, same ground factors with the same scopes"""
    import gzip, json
    rec = json.load(gzip.open(os.path.join(golden_dir, 'grounding.json.gz'), 'rt'))
    d = G.Domain((-50, 50), continuous=True, integral_points=np.linspace(-50, 50, 100))
    rgm = _rgm_relational(100, 10)
    dr = G.Domain((-15, 15), continuous=True, integral_points=np.linspace(0, 10, 20))
    db = G.Domain((0, 1))
    lv_p, lv_t = R.LV([f'p{i}' for i in range(300)]), R.LV([f't{i}' for i in range(10)])
    atoms = (R.Atom(db, (lv_t, lv_t), 'SameSession'), R.Atom(db, (lv_p, lv_t), 'PaperIn'),
             R.Atom(dr, (lv_t,), 'TopicPopularity'), R.Atom(dr, (lv_p,), 'PaperPopularity'))
    pp = R.RelationalGraph(atoms, (
        R.ParamF(M.MLNPotential(lambda x: M.eq_op(x[0], 1), w=0.3), nb=['PaperPopularity(p)']),
        R.ParamF(M.MLNPotential(lambda x: x[0] * M.eq_op(x[1], x[2]), w=0.5),
                 nb=['SameSession(t1,t2)', 'TopicPopularity(t1)', 'TopicPopularity(t2)'], constrain=lambda sub: sub['t1'] != sub['t2']),
        R.ParamF(M.MLNPotential(lambda x: x[0] * M.eq_op(x[1], x[2]), w=1), nb=['PaperIn(p,t)', 'PaperPopularity(p)', 'TopicPopularity(t)'])))
    from lhvi import generators
    for name, rel in (('rgm_100x10', rgm), ('paper_popularity_300x10', pp), ('rgm_100x10', generators.rgm()),
                      ('paper_popularity_300x10', generators.paper_popularity()), ('robot_mapping', generators.robot_mapping())):
        flat, keys = rel.ground_flat()
        want = rec[name]
        assert sorted(list(keys.key_of(v)) for v in range(flat.V)) == want['rvs']
        pf_of_pot = {}
        for i, pf in enumerate(rel.param_factors):
            doms = tuple(rel.atoms_dict[rel._parse(e)[0]].domain for e in pf.nb)
            pf_of_pot[[k for k, p in enumerate(flat.potentials) if p is pf.potential][0]] = i
        got = sorted([pf_of_pot[int(flat.fac_pot[f])], [list(keys.key_of(v)) for v in flat.edge_var[flat.fac_ptr[f]:flat.fac_ptr[f + 1]]]]
                     for f in range(flat.F))
        assert got == want['factors']


@pytest.mark.parametrize('n,T,missing', [(4, 5, 0.3), (3, 2, 0.0), (5, 7, 0.6), (2, 1, 0.0)])
def test_kalman_flat_builder_equals_object_builder(n, T, missing):
    """KalmanFilter.grounded_flat against grounded_graph + flatten: same variables, same factors in the same order, the
    same (kind, parameters) behind every factor (the flat builder deduplicates potentials by value)"""
    rng = np.random.default_rng(n * 10 + T)
    A = rng.normal(size=(n, n)) * (rng.random((n, n)) < 0.7)
    Cm = np.diag(rng.uniform(0.5, 1.5, n))
    data = rng.normal(size=(n, max(T, 1)))
    data[rng.random(data.shape) < missing] = kalman.MISSING
    data[:, 0] = rng.normal(size=n)
    dom = G.Domain((-20, 20), continuous=True, integral_points=np.linspace(-20, 20, 8))
    kf = kalman.KalmanFilter(dom, A, 0.7, Cm, 0.4)
    g, table = kf.grounded_graph(T, data)
    fo = flatten(g)
    ff, sid = kf.grounded_flat(T, data)
    assert (fo.V, fo.F, fo.E) == (ff.V, ff.F, ff.E)
    for name in ('fac_ptr', 'edge_var', 'var_ptr', 'var_edge', 'var_dom'):
        np.testing.assert_array_equal(getattr(fo, name), getattr(ff, name), err_msg=name)
    np.testing.assert_array_equal(np.nan_to_num(fo.var_value, nan=-1e9), np.nan_to_num(ff.var_value, nan=-1e9))
    for f in range(fo.F):
        po, pf = fo.fac_pot[f], ff.fac_pot[f]
        assert fo.pot_kind[po] == ff.pot_kind[pf]
        np.testing.assert_array_equal(fo.pot_param[fo.pot_off[po]:fo.pot_off[po + 1]], ff.pot_param[ff.pot_off[pf]:ff.pot_off[pf + 1]])
    for t in range(T):
        for i in range(n):
            assert fo.rvs[sid[t, i]] is table[t][i]


def test_utils_kl_helpers_have_the_reference_signatures():
    """utils.py:18-128: KL, kl_discrete, kl_continuous*, kl_normal(mu1, mu2, sig1, sig2)"""
    from math import exp, log, pi, sqrt
    from lhvi import utils
    from lhvi.graph import Domain

    def pdf(mu, sig):
        return lambda x: exp(-0.5 * ((x - mu) / sig) ** 2) / (sqrt(2 * pi) * sig)

    def logpdf(mu, sig):
        return lambda x: -0.5 * ((x - mu) / sig) ** 2 - log(sqrt(2 * pi) * sig)
    mu1, mu2, sig1, sig2 = 0.3, -0.4, 0.8, 1.5
    closed = utils.kl_normal(mu1, mu2, sig1, sig2)
    assert closed == pytest.approx(log(sig2 / sig1) + (sig1 ** 2 + (mu1 - mu2) ** 2) / (2 * sig2 ** 2) - 0.5, rel=1e-15)
    assert utils.kl_continuous(pdf(mu1, sig1), pdf(mu2, sig2), -12, 12) == pytest.approx(closed, rel=1e-7)
    assert utils.kl_continuous_no_add_const(pdf(mu1, sig1), pdf(mu2, sig2), -5, 5) == pytest.approx(closed, rel=1e-4)
    assert utils.kl_continuous_logpdf(logpdf(mu1, sig1), logpdf(mu2, sig2), -12, 12) == pytest.approx(closed, rel=1e-7)
    p, q = np.array([0.2, 0.5, 0.3]), np.array([0.3, 0.3, 0.4])
    assert utils.kl_discrete(p, q) == pytest.approx(float(np.sum(p * np.log(p / q))), rel=1e-14)
    dom = Domain((-12, 12), continuous=True, integral_points=np.linspace(-12, 12, 2001))
    assert utils.KL(pdf(mu1, sig1), pdf(mu2, sig2), dom) == pytest.approx(closed, rel=1e-3)      # Riemann sum of the grid
    dd = Domain((0, 1, 2))
    assert utils.KL(lambda x: p[int(x)], lambda x: q[int(x)], dd) == pytest.approx(utils.kl_discrete(p, q), abs=1e-6)
    compat_dir = os.path.join(ROOT, 'lifted-hybrid-variational-inference_amd', 'compat')
    sys.path.insert(0, compat_dir)
    try:
        sys.modules.pop('utils', None)
        compat = importlib.import_module('utils')               # `from utils import kl_continuous` of the reference's demos
        assert compat.kl_normal is utils.kl_normal and compat.kl_continuous is utils.kl_continuous
        assert compat.log_likelihood is utils.log_likelihood
    finally:
        sys.path.remove(compat_dir)
        sys.modules.pop('utils', None)


def test_utils_match_reference_values(golden_dir):
    """lhvi.utils (the reference's utils.py surface) against values computed by the reference itself
    (oracle/capture_utils.py -> tests/golden/utils.json): log_likelihood on four models incl. the -inf convention, KL,
    kl_discrete, the three kl_continuous variants, kl_normal"""
    from lhvi import utils
    rec = load(golden_dir, 'utils')
    for case in rec['log_likelihood']:
        g, rvs, factors = modelio.load_model(case['model'], API)
        asg = {rv: (v if rv.domain.continuous else int(v)) for rv, v in zip(rvs, case['x'])}
        got = utils.log_likelihood(g, asg)
        assert got == case['value'] if np.isinf(case['value']) else got == pytest.approx(case['value'], rel=1e-12)

    def pdf(mu, sig):
        return lambda x: np.exp(-((x - mu) / sig) ** 2 * 0.5) / (2.506628274631 * sig)

    def logpdf(mu, sig):
        return lambda x: -((x - mu) / sig) ** 2 * 0.5 - np.log(2.506628274631 * sig)
    for c in rec['kl']:
        dom = G.Domain((c['lo'], c['hi']), continuous=True, integral_points=np.linspace(c['lo'], c['hi'], c['points']))
        p, q = pdf(c['mu1'], c['s1']), pdf(c['mu2'], c['s2'])
        assert utils.KL(p, q, dom) == pytest.approx(c['KL'], rel=1e-12, abs=1e-15)
        assert utils.kl_continuous(p, q, c['lo'], c['hi']) == pytest.approx(c['kl_continuous'], rel=1e-10, abs=1e-14)
        assert utils.kl_continuous_no_add_const(p, q, c['lo'], c['hi']) == pytest.approx(c['kl_continuous_no_add_const'], rel=1e-10, abs=1e-14)
        assert utils.kl_continuous_logpdf(logpdf(c['mu1'], c['s1']), logpdf(c['mu2'], c['s2']), c['lo'], c['hi']) == \
            pytest.approx(c['kl_continuous_logpdf'], rel=1e-10, abs=1e-14)
        assert utils.kl_normal(c['mu1'], c['mu2'], c['s1'], c['s2']) == pytest.approx(c['kl_normal'], rel=1e-13, abs=1e-16)
    d = rec['discrete']
    tp, tq = np.array(d['p']), np.array(d['q'])
    assert utils.kl_discrete(tp, tq) == pytest.approx(d['kl_discrete'], rel=1e-13)
    assert utils.KL(lambda x: tp[x], lambda x: tq[x], G.Domain((0, 1, 2))) == pytest.approx(d['KL'], rel=1e-13)


def test_kalman_flat_builder_matches_reference_structure(golden_dir):
    """the array builder against the graph the reference's own KalmanFilter.grounded_graph built (fixture G2)"""
    rec = load(golden_dir, 'gauss_g2_kalman')
    n, T = 4, 5
    rng = np.random.RandomState(0)
    A = rng.uniform(-0.5, 0.5, size=(n, n)) + np.eye(n) * 0.5
    data = rng.uniform(-2, 2, size=(n, T))
    data[rng.rand(n, T) < 0.3] = 5000
    data[:, 0] = rng.uniform(-2, 2, size=n)
    d = G.Domain((-8, 8), continuous=True, integral_points=np.linspace(-8, 8, 32))
    flat, _ = kalman.KalmanFilter(d, A, 1.5, np.eye(n), 0.7).grounded_flat(T, data)
    model = rec['model']
    # same variables (evidence pattern / values), same multiset of (potential class, parameters, scope) factors
    ref_vals = np.array([np.nan if v is None else v for _, v in model['rvs']])
    ref_f = sorted((model['potentials'][pi]['cls'], round(model['potentials'][pi]['coeff'], 12), round(model['potentials'][pi]['sig'], 12),
                    tuple(nb)) for pi, nb in model['factors'])
    g, rvs, factors = modelio.load_model(model, API)
    ref_flat = flatten(g)
    # the fixture lists rvs in the reference's order; the flat builder numbers them in creation order of grounded_graph:
    # match variables through the object builder's table (pinned to the same fixture by the test above)
    assert flat.V == len(model['rvs']) and flat.F == len(model['factors']) and flat.E == ref_flat.E
    assert np.isnan(flat.var_value).sum() == np.isnan(ref_vals).sum()
    np.testing.assert_allclose(np.sort(flat.var_value[~np.isnan(flat.var_value)]), np.sort(ref_vals[~np.isnan(ref_vals)]), rtol=0, atol=0)
    kinds = {P.POT_LINEAR_GAUSSIAN: 'LinearGaussianPotential', P.POT_X2: 'X2Potential', P.POT_XY: 'XYPotential'}
    got_f = sorted((kinds[int(flat.pot_kind[p])], round(float(flat.pot_param[flat.pot_off[p]]), 12),
                    round(float(flat.pot_param[flat.pot_off[p] + 1]), 12)) for p in flat.fac_pot)
    assert got_f == sorted(t[:3] for t in ref_f)
    assert sorted(np.diff(flat.var_ptr).tolist()) == sorted(np.diff(ref_flat.var_ptr).tolist())      # same degree sequence


def test_design_numbers_are_the_committed_profiles():
    """the measured-numbers block of docs/measurement.md is the verbatim output of scripts/design_numbers.py on the committed
    profiles/r05_* files: a number cannot be quoted there that no file holds"""
    import re
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'design_numbers.py'), 'r05'], capture_output=True, text=True,
                         check=True).stdout.strip()
    text = open(os.path.join(ROOT, 'docs', 'measurement.md')).read()
    m = re.search(r'<!-- numbers:begin \(scripts/design_numbers.py\) -->\n(.*?)\n<!-- numbers:end -->', text, re.S)
    assert m and m.group(1).strip() == out


def test_c2fvi_evidence_bookkeeping():
    """lhvi.c2fvi's restatement of CompressedGraph.split_evidence (CGWO:236-247) on colour arrays: only clusters in
    `clustered_evidence` are examined, the standard deviation decides whether a cluster splits but the VARIANCE whether a
    piece is tracked again, piece 0 keeps the colour, single-member clusters leave the set"""
    from lhvi import c2fvi
    vals = np.array([np.nan, 1.0, 1.0, 9.0, 9.5, np.nan, 4.0, 4.0, 20.0])
    rvc = np.array([0, 1, 1, 1, 1, 0, 2, 2, 3], dtype=np.int32)
    # cluster 1 {1, 1, 9, 9.5}: std 4.1 > 3 -> k-means splits {1, 1} | {9, 9.5}; the second piece (variance 0.0625) is not
    # tracked at epsilon = 3 but piece 0 keeps its place in the set; cluster 2 is not tracked and stays; 3 is a singleton
    out, tracked = c2fvi.split_evidence_pass(vals, rvc, {1, 3}, 2, 10, 3.0)
    assert out.tolist() == [0, 1, 1, 4, 4, 0, 2, 2, 3] and tracked == {1, 3}
    out2, tracked2 = c2fvi.split_evidence(vals, out, {1, 3, 4}, 2, 10, 0.0)
    assert sorted(set(out2.tolist())) == [0, 1, 2, 3, 4, 5] and out2[3] != out2[4]          # {9} | {9.5} at epsilon 0
    assert c2fvi.evidence_variances(vals, out2)[2] == 0.0 and 3 in tracked2                  # (a singleton is only dropped when examined)
    # an untracked cluster is never split, whatever its spread (the path dependence the fixtures avoid)
    out3, _ = c2fvi.split_evidence(vals, rvc, set(), 2, 10, 0.0)
    assert (out3 == rvc).all()


def test_conditional_quadratic_view_of_the_reference_formulas():
    """every hybrid formula the reference ships (Demo/Data/HMLN/GeneratorPaperPopularity.py:28-40,
    GeneratorRobotMapping.py:60-75) has a conditional-quadratic view that reproduces the lambda on random points; formulas
    outside the family (cubic terms, comparisons of continuous arguments, no continuous argument) have none"""
    from lhvi import expr
    from lhvi.graph import Domain
    from lhvi.mln import MLNPotential, eq_op, neg_op, or_op
    rng = np.random.default_rng(0)
    b = Domain((0, 1))
    r = Domain((-15, 15), continuous=True, integral_points=np.linspace(0, 10, 20))
    tri = Domain((0, 1, 2))
    cases = [(lambda x: x[0] * eq_op(x[1], x[2]), 0.5, (b, r, r)), (lambda x: eq_op(x[0], 1), 0.3, (r,)),
             (lambda x: x[0] * eq_op(x[1], 0.341), 3.754, (b, r)), (lambda x: x[1] * eq_op(x[0], x[2]) + 0.5 * x[1], 2.0, (r, b, r)),
             (lambda x: (x[0] == 2) * x[2] * x[3] - x[1] * x[2] ** 2 / 4, 1.5, (tri, b, r, r))]
    for formula, w, doms in cases:
        pot = MLNPotential(formula, w)
        kind, par = pot.device_spec(doms)
        ncode = int(par[1])
        tail = par[3 + 2 * ncode:]
        assert tail and tail[0] == expr.CQ_MAGIC
        arity, nd, nc = int(tail[1]), int(tail[2]), int(tail[3])
        role, dims = [int(v) for v in tail[4:4 + arity]], [int(v) for v in tail[4 + arity:4 + arity + nd]]
        coef = np.array(tail[4 + arity + nd:]).reshape(-1, 6)
        assert arity == len(doms) and coef.shape[0] == int(np.prod(dims)) if dims else 1
        for _ in range(50):
            x, cfg, cont = [], 0, []
            for a, d in enumerate(doms):
                if d.continuous:
                    x.append(float(rng.uniform(-15, 15)))
                    cont.append(x[-1])
                else:
                    k = int(rng.integers(len(d.values)))
                    x.append(d.values[k])
                    cfg = cfg * len(d.values) + k
            u, v = cont[0], (cont[1] if nc == 2 else 0.0)
            c = coef[cfg]
            got = c[0] * u * u + c[1] * u * v + c[2] * v * v + c[3] * u + c[4] * v + c[5]
            assert got == pytest.approx(w * formula(x), rel=1e-12, abs=1e-10)
    for formula, doms in [(lambda x: x[0] * x[1] ** 2 * x[2], (b, r, r)), (lambda x: (x[1] > x[2]) * x[0], (b, r, r)),
                          (lambda x: x[0] * x[1] * x[2], (r, r, r))]:
        kind, par = MLNPotential(formula, 1.0).device_spec(doms)
        assert len(par) == 3 + 2 * int(par[1]) and par[2] == 0
    # a formula over discrete arguments only is the limiting case: a table in log space (Nc = 0, only the constants are set),
    # so that no formula the reference ships is interpreted on the device (GeneratorRobotMapping.py:31-43)
    for formula, w, doms in [(lambda x: or_op(neg_op(x[0]), neg_op(x[1])), 3.0, (b, b)),
                             (lambda x: 1 - (x[0] == 1) * (x[1] == 1) * (x[2] == 0) * (x[3] == 1) * (1 - x[4]), 1.591, (b,) * 5),
                             (lambda x: x[0], -0.737, (tri,))]:
        kind, par = MLNPotential(formula, w).device_spec(doms)
        tail = par[int(par[2]):]
        assert par[2] == 3 + 2 * int(par[1]) and tail[0] == expr.CQ_MAGIC and (int(tail[2]), int(tail[3])) == (len(doms), 0)
        coef = np.array(tail[4 + 2 * len(doms):]).reshape(-1, 6)
        import itertools
        for cfg, states in enumerate(itertools.product(*[d.values for d in doms])):
            assert (coef[cfg, :5] == 0).all() and coef[cfg, 5] == pytest.approx(w * formula(list(states)), rel=1e-15, abs=1e-15)


def test_robot_mapping_model_matches_the_reference_demo(golden_dir, tmp_path):
    """The reference's second HMLN demo (Demo/HMLN/DemoRobotMapping.py:11-27 on Demo/Data/HMLN/GeneratorRobotMapping.py):
    ``generators.robot_mapping()`` grounded by the object path and by ``ground_flat`` against the reference's own grounding,
    the demo's evidence (raw data + closed world; recorded by oracle/capture_robot.py) applied through both paths, and the
    raw-data parser on a text with the same constructs (comment blocks, valued atoms, atoms naming no ground variable)"""
    import gzip, json
    from lhvi import generators
    rec = json.load(gzip.open(os.path.join(golden_dir, 'grounding.json.gz'), 'rt'))['robot_mapping']
    rel = generators.robot_mapping()
    g, table = rel.ground_graph()
    assert sorted(list(k) for k in table) == rec['rvs']
    assert len(g.rvs) == 1591 and len(g.factors) == 3182 and sum(len(f.nb) for f in g.factors) == 14282
    key_of = {id(rv): list(k) for k, rv in table.items()}
    pf_of = {id(pf.potential): i for i, pf in enumerate(rel.param_factors)}
    assert sorted([pf_of[id(f.potential)], [key_of[id(rv)] for rv in f.nb]] for f in g.factors) == rec['factors']
    # arity-5 formula, constants and the depth domain whose integral points leave the domain
    assert max(len(f.nb) for f in g.factors) == 5
    depth = table[('Depth', 'A1_1')].domain
    assert depth.values == (0, 0.5) and depth.integral_points.max() == 1.0
    # evidence: the demo's dict through add_evidence and through ground_flat
    data = {tuple(k): v for k, v in rec['evidence']}
    data.update({tuple(k): 1 for k in rec['raw_keys_without_atom']})       # e.g. SegType(A1_1, Door): names no atom, ignored
    rel.add_evidence(data)
    hidden = [k for k, rv in table.items() if rv.value is None]
    from collections import Counter
    assert Counter(k[0] for k in hidden) == {'SegType': 111, 'PartOf': 55, 'Depth': 20, 'Length': 9}
    flat, keys = generators.robot_mapping().ground_flat(data)
    assert int(flat.var_hidden.sum()) == 195
    for v in range(flat.V):
        val = table[keys.key_of(v)].value
        assert (val is None and np.isnan(flat.var_value[v])) or val == flat.var_value[v]
    # closed world: what the demo adds to the raw data is every non-query discrete atom that the data does not mention
    raw = {k: v for k, v in data.items() if not (k[0] == 'Aligned' and v == 0)}
    filled = generators.closed_world(table, raw, query=('SegType', 'PartOf', 'Length', 'Depth'))
    assert {k: float(v) for k, v in filled.items() if k in table} == {tuple(k): v for k, v in rec['evidence']}
    # the parser
    text = 'PartOf(A1_2,LA1)\n\nSegType(A1_1,Door)\n/*\nSegType(A1_10,Wall)\n*/\nLength(A1_1) 0.0979\n/* x */\nDepth(A1_3) 0.02\nAligned(A1_2,A1_1)\n'
    path = tmp_path / 'raw'
    path.write_text(text)
    # (a one-line comment opens a block that only a later "*/" line closes, exactly as the reference's parser behaves)
    assert generators.load_raw_data(str(path)) == {('PartOf', 'A1_2', 'LA1'): 1, ('SegType', 'A1_1', 'Door'): 1, ('Length', 'A1_1'): 0.0979}


def test_vi_factor_lists_split():
    """``vi.factor_lists``: every factor in exactly one segment; pairwise continuous factors on the fast path; the group kernel's
    budget (axis lengths sum to <= 24, times K <= 48); per-edge axis records agree with the graph"""
    from lhvi import synth
    from lhvi.vi import factor_lists
    flat, _ = synth.paper_popularity_flat(40, 5, seed=1, points=20)
    order, counts, rec = factor_lists(flat, 2, 3, tiny_kernel='always')
    assert sorted(order.tolist()) == list(range(flat.F)) and sum(counts) == flat.F
    assert counts[0] == 0 and counts[1] == flat.F            # MLN formulas of arity <= 3 with at most 2 * 3 * 3 grid nodes: tiny
    for policy in (True, False):                             # ... or, this few of them / without that kernel, the group kernel: K * S <= 2 * 8
        o_g, counts_g, _ = factor_lists(flat, 2, 3, tiny_kernel=policy)
        assert counts_g[2] == flat.F and sorted(o_g.tolist()) == list(range(flat.F))
    order5, counts5, _ = factor_lists(flat, 7, 3, tiny_kernel='always')    # K = 7: no tiny kernel; 7 * 8 slots > 48 for the all-hidden ternary factors
    assert counts5[1] == 0 and counts5[4] > 0 and counts5[2] + counts5[4] == flat.F
    assert (rec[:, 3] == flat.dom_ptr[flat.var_dom[flat.edge_var]]).all()
    hid = flat.var_hidden[flat.edge_var]
    assert (rec[:, 0] == flat.edge_var).all() and (((rec[:, 1] >> 16) & 1) == hid).all()
    lens = rec[:, 1] & 0xffff
    assert (lens[hid & flat.var_cont[flat.edge_var]] == 3).all() and (lens[~hid] == 1).all() and (lens[hid & ~flat.var_cont[flat.edge_var]] == 2).all()
    obs_d = ~hid & ~flat.var_cont[flat.edge_var]
    assert (rec[obs_d, 2] == flat.var_value[flat.edge_var[obs_d]].astype(int)).all()          # states (0, 1): index == value
    rg, _, _, _ = synth.rgm_flat(C=6, B=4, evidence_ratio=0.3, seed=0)
    o2, c2, _ = factor_lists(rg, 2, 3)
    assert c2 == (rg.F, 0, 0, 0, 0, 0)


def test_ground_order_of_set_based_graphs_is_node_id_order():
    """a reference-style Graph holds its rvs / factors in Python sets: the ground numbering (``flat.ground_order``) is creation order,
    not set-iteration order (which follows object hashes and changes from process to process); lists keep their order.  A cluster's
    evidence mean is the running sum of its members in that order -- the array path's order (``lifting.segment_sums``)."""
    from lhvi import lifting
    from lhvi.flat import flatten, ground_order
    from lhvi.graph import Domain, F, Graph, RV
    from lhvi.potentials import GaussianPotential
    dom = Domain((-10, 10), continuous=True, integral_points=np.linspace(-10, 10, 5))
    vals = [0.1, 0.7, 1e16, -1e16, 0.3, 0.2, None, None]
    rvs = [RV(dom, v) for v in vals]
    pot = GaussianPotential([0.0, 0.0], [[2.0, 0.5], [0.5, 2.0]])
    fs = [F(pot, nb=[rvs[i], rvs[6 + i % 2]]) for i in range(6)]
    g = Graph()
    g.rvs, g.factors = set(rvs[::-1]), set(fs[::-1])
    g.init_nb()
    assert ground_order(g.rvs) == rvs and ground_order(g.factors) == fs and ground_order(rvs[::-1]) == rvs[::-1]
    flat = flatten(g)
    assert [flat.var_index[r] for r in rvs] == list(range(8))
    # all observed variables in one cluster: the mean is the running sum in ground order, 1e16 and -1e16 included
    rv_color = np.array([0, 0, 0, 0, 0, 0, 1, 2], dtype=np.int32)
    f_color = np.arange(6, dtype=np.int32)
    super_rvs, _ = lifting.build_lifted_objects(g, rv_color, f_color)
    total = 0
    for v in vals[:6]:
        total += v
    assert super_rvs[0].value == total / 6 and super_rvs[0].value != sum(sorted(vals[:6])) / 6
    R = lifting._lift_reduce_host(flat, rv_color, f_color)
    assert R['val'][0] == total / 6


def test_v2f_records_name_the_row_the_graph_holds():
    """``lhvi_pbp_t.v2f_wide`` as records (LHVI_PBP_V2F_RECORDS): variable, degree, particle count, domain and the first four
    incident edges in row order -- what ``pbp_v2f_kernel`` would otherwise read through var_ptr / var_edge (host side only)"""
    from lhvi import synth
    from lhvi.pbp import EPBP
    flat = synth.hybrid_mrf_flat(V=700, deg=4, seed=3)
    bp = EPBP.__new__(EPBP)
    bp.np_host = np.where(flat.var_cont, 64, 2).astype(np.int64)
    vs = np.flatnonzero(flat.var_hidden & flat.var_cont)
    rec = bp._v2f_records(flat, vs)
    assert rec.shape == (vs.size, 8) and rec.dtype == np.int32
    deg = np.diff(flat.var_ptr)[vs]
    np.testing.assert_array_equal(rec[:, 0], vs)
    np.testing.assert_array_equal(rec[:, 1], deg)
    np.testing.assert_array_equal(rec[:, 2], 64)
    np.testing.assert_array_equal(rec[:, 3], flat.var_dom[vs])
    for i in range(0, vs.size, 37):
        row = flat.var_edge[flat.var_ptr[vs[i]]:flat.var_ptr[vs[i] + 1]]
        want = [row[min(k, row.size - 1)] for k in range(4)] if row.size else [0] * 4
        assert rec[i, 4:].tolist() == [int(x) for x in want]
    assert bp._v2f_records(flat, vs[:0]).shape == (1, 8)          # an empty list keeps a non-null pointer


@pytest.mark.parametrize('model', ['rgm', 'rgm pooled', 'hmln'])
def test_lift_flat_equals_flatten_of_the_cluster_objects(model):
    """a stable partition's lifted graph two ways: ``flatten`` of the SuperRV / SuperF objects and ``lifting.lift_flat`` on the ground
    arrays (what ``CompressedGraph.lifted_flat`` hands to ``flatten`` once the colour passing has reached its fixed point) --
    the same incidences, canonical edges, rows, counts, multiplicities and cluster values, the same potential and domain per
    factor / variable (the tables themselves are numbered differently), cluster ids = indices"""
    from lhvi import flat as F, generators, lifting
    from oracle import oracle
    rng = np.random.default_rng(4)
    if model == 'hmln':
        rel = generators.paper_popularity(30, 4)
        rel.ground_graph()
        data = {k: (int(rng.integers(0, 2)) if k[0] in ('SameSession', 'PaperIn') else float(np.round(rng.uniform(0, 10), 1)))
                for k in sorted(rel.rvs_dict) if rng.random() < 0.3}
    else:
        rel = generators.rgm(40, 20)
        rel.ground_graph()
        keys = sorted(rel.rvs_dict)
        pick = rng.choice(len(keys), len(keys) // 4, replace=False)
        data = {keys[i]: (float(rng.choice([1.5, -2.0, 7.25])) if 'pooled' in model else float(np.round(rng.uniform(-30, 30), 2))) for i in pick}
    g, _ = rel.add_evidence(data)
    fl = F.flatten(g)
    rv0, f0 = lifting.initial_colors(g, True)
    sym = np.array([1 if getattr(f.potential, 'symmetric', False) else 0 for f in fl.factors], dtype=np.uint8)
    rvc, fc = oracle.color_passing(fl, sym, rv0, f0)
    rvs, factors = lifting.build_lifted_objects(g, rvc, fc)[:2]

    class Lifted:
        pass
    lg = Lifted()
    lg.rvs, lg.factors = rvs, factors
    a, b = F.flatten(lg), lifting.lift_flat(fl, rvc, fc)
    for name in ('fac_ptr', 'edge_var', 'edge_fac', 'edge_pos', 'edge_canon', 'var_ptr', 'var_edge', 'edge_count', 'var_mult', 'fac_mult'):
        np.testing.assert_array_equal(getattr(a, name), getattr(b, name), err_msg=name)
    np.testing.assert_array_equal(a.var_value, b.var_value)
    assert all(a.domains[i] is b.domains[j] for i, j in zip(a.var_dom, b.var_dom))
    assert all(a.potentials[i] is b.potentials[j] for i, j in zip(a.fac_pot, b.fac_pot))
    row = lambda t, i: (int(t.pot_kind[i]), tuple(t.pot_param[t.pot_off[i]:t.pot_off[i + 1]]))
    assert [row(a, i) for i in a.fac_pot] == [row(b, i) for i in b.fac_pot]
    assert [c.id for c in a.rvs] == list(range(a.V)) and [c.id for c in a.factors] == list(range(a.F))
