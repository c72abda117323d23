"""CPU suite: the particle-BP oracle reproduces the vectors captured from the reference's EPBP / HybridLBP."""
import json
import os

import numpy as np
import pytest

import modelio
from test_oracle_golden import API, _initial
from lhvi import lifting
from lhvi.flat import flatten
from oracle import oracle

EPBP_CASES = ['epbp_kalman_simple', 'epbp_kalman_ep', 'epbp_kalman_n64', 'epbp_hybrid_ep', 'epbp_hybrid_simple',
              'epbp_hmln', 'epbp_robot']
HLBP_CASES = ['hlbp_rgm_small', 'hlbp_hybrid', 'hlbp_kalman_full', 'hlbp_hmln', 'hlbp_hmln_ep', 'hlbp_hmln_lifted',
              'hlbp_robot']
C2F_CASES = ['hlbp_c2f_rgm', 'hlbp_c2f_rgm_simple', 'hlbp_c2f_hmln']

# fp64 tolerance of the log-message tables: the oracle fuses nothing and follows the reference's operation
# order, so differences come only from libm (pow(e,x) vs CPython's) and the exact-rational mean
RTOL, ATOL = 1e-9, 1e-9


def load_npz(golden_dir, name):
    z = np.load(os.path.join(golden_dir, 'pbp_%s.npz' % name))
    meta = json.loads(str(z['meta']))
    return z, meta


def check_against_snapshots(z, meta, flat, orc_factory, iterations):
    n = meta['n']
    K = z['sample'].shape[0]
    assert K == iterations          # initial draw + one per non-final iteration
    seen = []

    def on_it(i, o):
        want_v2f = z['v2f'][i]
        hid = flat.var_hidden[flat.edge_var]
        npv = o.np[flat.edge_var]
        for e in np.flatnonzero(hid):
            np.testing.assert_allclose(o.v2f[e, :npv[e]], want_v2f[e, :npv[e]], rtol=RTOL, atol=ATOL,
                                       err_msg='v2f it %d edge %d' % (i, e))
        if i < iterations - 1:
            cont = flat.var_hidden & flat.var_cont
            np.testing.assert_allclose(o.q[cont], z['q'][i][cont], rtol=1e-9, atol=1e-12, err_msg='q it %d' % i)
            ce = cont[flat.edge_var]
            np.testing.assert_allclose(o.eta[ce], z['eta'][i][ce], rtol=1e-9, atol=1e-12, err_msg='eta it %d' % i)
        if i > 0:
            # f2v computed in iteration i-1 is tabulated on sample i (+ grid)
            want = z['f2v'][i]
            for e in np.flatnonzero(hid):
                v = flat.edge_var[e]
                got = o.f2v_prev[e]
                np.testing.assert_allclose(got[:npv[e]], want[e, :npv[e]], rtol=RTOL, atol=ATOL)
                if flat.var_cont[v]:
                    T = flat.var_nstates[v]
                    np.testing.assert_allclose(got[n:n + T], want[e, n:n + T], rtol=RTOL, atol=ATOL)
        seen.append(i)

    o = orc_factory()
    orig_f2v = o.step_f2v

    def step_f2v():
        orig_f2v()
        o.f2v_prev = o.f2v.copy()
    o.step_f2v = step_f2v
    o.f2v_prev = None
    o.run(iterations, [z['sample'][k] for k in range(K)], on_it)
    assert seen == list(range(iterations))
    return o


@pytest.mark.parametrize('name', EPBP_CASES)
def test_epbp_oracle_matches_reference(golden_dir, name):
    z, meta = load_npz(golden_dir, name)
    g, rvs, factors = modelio.load_model(meta['model'], API)
    flat = flatten(g, require_device_potentials=True)
    o = check_against_snapshots(
        z, meta, flat,
        lambda: oracle.PbpOracle(flat, meta['n'], ep=meta['approx'] == 'EP', epbp=True, var_threshold=3),
        meta['iterations'])
    # post-sweep queries: belief_rv at recorded points
    hid = np.flatnonzero(flat.var_hidden)
    got = o.belief_points(hid, z['query_x'][hid])
    np.testing.assert_allclose(got, z['query_logb'][hid], rtol=1e-9, atol=1e-8)
    check_probability(z, rvs, lambda i, xs: o.belief_points(np.array([i]), np.asarray(xs)[None, :])[0])


def check_probability(z, rvs, log_belief_of):
    """the reference's recorded probability(a, b, rv) (EPBP:356-375 / HLBP:384-403); `log_belief_of(i, xs)`"""
    if 'probability' not in z.files:
        return
    for i, a, b, want in z['probability']:
        rv = rvs[int(i)]
        got = oracle.interval_probability(lambda xs: log_belief_of(int(i), xs), a, b, rv.domain.values[0], rv.domain.values[1])
        assert got == pytest.approx(want, rel=1e-8, abs=1e-300)


def check_draw_tables(z, k, rvs, n, edge_of, f2v, eta, c2f=False):
    """message f -> rv at the integral points and the sites (eta) the reference held at its k-th generate_sample call
    (recorded per ground rv and per ground factor of it, HLBP:100-118,182-215): `edge_of(rv, f)` -> row of f2v / eta.
    In a coarse-to-fine run a factor cluster created by the split_factors just before the draw has inherited its sites but
    not its f -> rv table (HLBP:292-308; it is recomputed right after the draw): those rows are absent from the record."""
    want_m, want_e = z['draw_f2v_grid'][k], z['draw_eta'][k]
    checked = 0
    for i, rv in enumerate(rvs):
        if rv.value is not None or not rv.domain.continuous:
            continue
        T = len(rv.domain.integral_points)
        for s_, f in enumerate(rv.nb):
            e = edge_of(rv, f)
            missing = np.isnan(want_m[i, s_, :T])
            assert missing.all() if (c2f and missing.any()) else not missing.any()
            if not missing.any():
                np.testing.assert_allclose(f2v[e, n:n + T], want_m[i, s_, :T], rtol=RTOL, atol=ATOL,
                                           err_msg='f2v grid at draw %d rv %d factor %d' % (k, i, s_))
            np.testing.assert_allclose(eta[e], want_e[i, s_], rtol=1e-9, atol=1e-12,
                                       err_msg='eta at draw %d rv %d factor %d' % (k, i, s_))
            checked += 1
    assert checked


def lifted_edge_of(flat):
    def edge_of(rv, f):
        fi = flat.fac_index[f.cluster]
        pos = next(i for i, r in enumerate(f.nb) if r is rv)
        return int(flat.edge_canon[flat.fac_ptr[fi] + pos])
    return edge_of


@pytest.mark.parametrize('name', HLBP_CASES)
def test_hlbp_oracle_matches_reference(golden_dir, name):
    z, meta = load_npz(golden_dir, name)
    g, rvs, factors = modelio.load_model(meta['model'], API)
    gflat = flatten(g)
    sym, rv0, f0 = _initial(gflat, g)
    rv_color, f_color = oracle.color_passing(gflat, sym, rv0, f0)
    assert oracle.canonical_labels(rv_color) == z['rv_label'].tolist()
    assert oracle.canonical_labels(f_color) == z['f_label'].tolist()
    cg = lifting.CompressedGraph(g)
    cg.set_colors(rv_color, f_color)
    flat = flatten(cg, require_device_potentials=True)
    rep = np.array([rvs.index(min(c.rvs)) for c in flat.rvs])
    samples = [z['samples'][k][rep] for k in range(z['samples'].shape[0])]
    o = oracle.PbpOracle(flat, meta['n'], ep=meta['approx'] == 'EP', epbp=False, var_threshold=5)
    # at draw k = i + 1 the reference holds f2v of sweep i - 1 (zeros for i = 0) and the sites of sweep i
    o.run(meta['iterations'], samples,
          lambda i, o: i < meta['iterations'] - 1 and check_draw_tables(z, i + 1, rvs, meta['n'], lifted_edge_of(flat), o.f2v, o.eta))
    cl = np.array([flat.var_index[rv.cluster] for rv in rvs])
    hid = np.flatnonzero(gflat.var_hidden)
    np.testing.assert_allclose(o.q[cl][gflat.var_hidden & gflat.var_cont],
                               z['final_q'][gflat.var_hidden & gflat.var_cont], rtol=1e-9)
    got = o.belief_points(cl[hid], z['query_x'][hid])
    np.testing.assert_allclose(got, z['query_logb'][hid], rtol=1e-9, atol=1e-8)
    check_probability(z, rvs, lambda i, xs: o.belief_points(cl[[i]], np.asarray(xs)[None, :])[0])


class OracleRefiner:
    """lhvi.c2f.Refiner backed by the exact CPU colour refinement"""

    def __init__(self, g):
        self.gflat = flatten(g)
        self.sym = np.array([1 if getattr(f.potential, 'symmetric', False) else 0 for f in self.gflat.factors])

    def factors(self, rvc, fc):
        return oracle.refine_factors(self.gflat, self.sym, rvc, fc)[0]

    def rvs(self, fc, rvc):
        return oracle.refine_rvs(self.gflat, fc, rvc)[0]


from oracle.engines import OracleEngine, OracleTensorRefiner      # noqa: E402  (shared with the scripts' CPU-baseline legs)


def ground_edges(flat, ground_rv):
    acc = {}
    for f in ground_rv.nb:
        fi = flat.fac_index[f.cluster]
        pos = next(i for i, r in enumerate(f.nb) if r is ground_rv)
        e = int(flat.edge_canon[flat.fac_ptr[fi] + pos])
        acc[e] = acc.get(e, 0) + 1
    return acc


def c2f_table_observer(z, rvs, factors, n, host):
    """lhvi.c2f observer: the variable-side tables at every draw against the reference's recorded messages / sites"""
    ridx = {id(rv): i for i, rv in enumerate(rvs)}
    fidx = {id(f): i for i, f in enumerate(factors)}

    def observer(k, rvc, old_fc, G1, pair_phi, st1):
        pair = {(int(A), int(phi)): e for e, (A, phi) in enumerate(zip(G1.edge_var, pair_phi))}
        check_draw_tables(z, k, rvs, n, lambda rv, f: pair[(int(rvc[ridx[id(rv)]]), int(old_fc[fidx[id(f)]]))],
                          host(st1.f2v), host(st1.eta), c2f=True)
    return observer


@pytest.mark.parametrize('name', C2F_CASES)
def test_hlbp_c2f_oracle_matches_reference(golden_dir, name):
    """HybridLBP.run(c2f=0) driven through lhvi.c2f with the CPU oracle as the engine vs the reference: partitions at
    every draw (exact), proposals at every draw, final log-beliefs through the ground variables' factors"""
    from lhvi import c2f
    z, meta = load_npz(golden_dir, name)
    g, rvs, factors = modelio.load_model(meta['model'], API)
    samples = z['samples']
    q_at_draw = []

    def draw(k, flat, q):
        q_at_draw.append(q[np.searchsorted(np.sort(flat.rep_ground), flat.rep_ground)] if False else (flat.rep_ground.copy(), q.copy()))
        return samples[k][flat.rep_ground]

    st, flat, cg, rvc, fc, history = c2f.run_c2f(g, OracleEngine(meta['n'], meta['approx'] == 'EP'), OracleRefiner(g),
                                                meta['iterations'], meta['c2f'], 2, 10, draw,
                                                observer=c2f_table_observer(z, rvs, factors, meta['n'], lambda a: a))
    assert len(history) == z['draw_rv_labels'].shape[0]
    for k, (r, f) in enumerate(history):
        assert oracle.canonical_labels(r) == z['draw_rv_labels'][k].tolist(), 'rv partition at draw %d' % k
        assert oracle.canonical_labels(f) == z['draw_f_labels'][k].tolist(), 'factor partition at draw %d' % k
        # proposals per ground rv at the draw
        cl = history[k][0]
        want = z['draw_q'][k]
        got = q_at_draw[k][1][cl]
        m = ~np.isnan(want[:, 0])
        np.testing.assert_allclose(got[m], want[m], rtol=1e-8, atol=1e-10, err_msg='q at draw %d' % k)
    assert oracle.canonical_labels(rvc) == z['rv_label'].tolist() and oracle.canonical_labels(fc) == z['f_label'].tolist()
    gflat = flatten(g)
    hid = np.flatnonzero(gflat.var_hidden)
    for i in hid:
        acc = ground_edges(flat, rvs[i])
        edges = np.array(list(acc))
        vals = st.edge_points(edges, np.tile(z['query_x'][i], (edges.size, 1)))
        got = (vals * np.array([acc[e] for e in acc], dtype=float)[:, None]).sum(axis=0)
        np.testing.assert_allclose(got, z['query_logb'][i], rtol=1e-8, atol=1e-6)

    def log_belief(i, xs):
        acc = ground_edges(flat, rvs[i])
        edges = np.array(list(acc))
        vals = st.edge_points(edges, np.tile(np.asarray(xs), (edges.size, 1)))
        return (vals * np.array([acc[e] for e in acc], dtype=float)[:, None]).sum(axis=0)
    check_probability(z, rvs, log_belief)


@pytest.mark.parametrize('name', ['epbp_kalman_simple', 'epbp_hybrid_ep', 'epbp_hmln'])
def test_pure_python_restatement_matches_reference(golden_dir, name):
    """oracle/pyref.py (the dict-of-dicts sweep bench.py times as ``cpu_baseline.python``) against the reference's recorded
    messages, proposals and sites of every iteration, and its log-beliefs"""
    from oracle import pyref
    z, meta = load_npz(golden_dir, name)
    g, rvs, factors = modelio.load_model(meta['model'], API)
    n, its = meta['n'], meta['iterations']
    edges = [(f, rv) for f in factors for rv in f.nb]
    samples = [pyref.sample_dicts(rvs, z['sample'][k]) for k in range(z['sample'].shape[0])]
    seen = []

    def on_it(i, bp):
        for e, (f, rv) in enumerate(edges):
            if rv.value is not None:
                continue
            pts = samples[i][rv]
            np.testing.assert_allclose([bp.message[(rv, f)][x] for x in pts], z['v2f'][i][e, :len(pts)], rtol=1e-12, atol=1e-10)
            if i > 0:       # f2v of sweep i-1 lives on sample i (+ the integral points); still in place before this sweep's f2v
                np.testing.assert_allclose([bp.message[(f, rv)][x] for x in pts], z['f2v'][i][e, :len(pts)], rtol=1e-12, atol=1e-10)
                if rv.domain.continuous:
                    grid = rv.domain.integral_points
                    np.testing.assert_allclose([bp.message[(f, rv)][x] for x in grid], z['f2v'][i][e, n:n + len(grid)],
                                               rtol=1e-12, atol=1e-10)
            if i < its - 1 and rv.domain.continuous:
                np.testing.assert_allclose(bp.eta_message[(f, rv)], z['eta'][i][e], rtol=1e-12)
        if i < its - 1:
            for k, rv in enumerate(rvs):
                if rv.value is None and rv.domain.continuous:
                    np.testing.assert_allclose(bp.q[rv], z['q'][i][k], rtol=1e-12)
        seen.append(i)

    bp = pyref.DictEPBP(g, n, meta['approx'])
    # the snapshot of iteration i holds v2f of sweep i and f2v of sweep i-1, both keyed by sample i: check before the install
    bp.start(samples[0])
    for i in range(its):
        bp.v2f_half()
        if i < its - 1:
            bp.update_proposal()
            on_it(i, bp)
            bp.install(samples[i + 1])
            bp.f2v_half()
        else:
            on_it(i, bp)
    assert seen == list(range(its))
    hid = [i for i, rv in enumerate(rvs) if rv.value is None]
    for i in hid[:4]:
        got = [bp.belief_rv(x, rvs[i]) for x in z['query_x'][i]]
        np.testing.assert_allclose(got, z['query_logb'][i], rtol=1e-10, atol=1e-9)


@pytest.mark.parametrize('name', C2F_CASES)
def test_hlbp_c2f_on_arrays_equals_the_object_path(golden_dir, name):
    """``run_c2f_flat`` (ground FlatGraph in, colour tensors, both lifted graphs of every sweep built from them with tensor
    operations, inheritance through vectorised pair maps) against ``run_c2f`` on the objects with the same engine and refiner:
    identical partitions at every draw, identical lifted graphs, identical tables -- and through the same observer against the
    reference's recorded tables"""
    from lhvi import c2f
    z, meta = load_npz(golden_dir, name)
    g, rvs, factors = modelio.load_model(meta['model'], API)
    samples = z['samples']
    draw = lambda k, flat, q: samples[k][flat.rep_ground]
    graphs_a, graphs_b = [], []

    class Eng(OracleEngine):
        def __init__(self, log, *a):
            super().__init__(*a)
            self.log = log

        def make(self, flat, sides='vf'):
            self.log.append((sides, flat))
            return super().make(flat, sides)
    ref = c2f.run_c2f(g, Eng(graphs_a, meta['n'], meta['approx'] == 'EP'), OracleRefiner(g), meta['iterations'], meta['c2f'], 2, 10, draw)
    gflat = flatten(g, require_device_potentials=True)
    rvc0, fc0, sym = lifting.initial_colors_flat(gflat, is_split_cont_evidence=False)
    tg = lifting.TensorGraph(gflat)
    got = c2f.run_c2f_flat(gflat, tg, Eng(graphs_b, meta['n'], meta['approx'] == 'EP'), OracleTensorRefiner(gflat, sym),
                           meta['iterations'], meta['c2f'], 2, 10, draw, rvc0, fc0,
                           observer=c2f_table_observer(z, rvs, factors, meta['n'], lambda a: a), keep_history=True)
    st_a, flat_a, cg, rvc_a, fc_a, hist_a = ref
    st_b, flat_b, rvc_b, fc_b, hist_b = got
    assert len(hist_a) == len(hist_b) == z['draw_rv_labels'].shape[0]
    for (ra, fa), (rb, fb) in zip(hist_a, hist_b):
        assert (ra == rb).all() and (fa == fb).all()                  # same colours, not only the same partition
    assert (rvc_a == rvc_b.numpy()).all() and (fc_a == fc_b.numpy()).all()
    # every lifted graph of every sweep, both sides.  The object path builds two graphs per sweep; the array path builds one only
    # when a refinement split something since the last one of that side (it reuses graph, state and maps otherwise)
    fields = ('fac_ptr', 'edge_var', 'edge_fac', 'edge_canon', 'var_ptr', 'var_edge', 'edge_count', 'var_mult', 'fac_mult')

    def same(A, B):
        return all(getattr(A, f) == getattr(B, f) for f in ('V', 'F', 'E')) and all(np.array_equal(getattr(A, f), getattr(B, f)) for f in fields)
    assert len(graphs_b) <= len(graphs_a) and len(graphs_b) < len(graphs_a) or meta['iterations'] < 3
    for side in ('v', 'vf'):
        seq_a = []
        for sd, G in graphs_a:
            if sd == side and not (seq_a and same(seq_a[-1], G)):
                seq_a.append(G)
        seq_b = [G for sd, G in graphs_b if sd == side]
        assert len(seq_a) == len(seq_b), side
        for A, B in zip(seq_a, seq_b):
            assert same(A, B)
            np.testing.assert_allclose(A.var_value, B.var_value, rtol=1e-14, equal_nan=True)
    for f in ('f2v', 'v2f', 'eta', 'q', 'particles'):
        np.testing.assert_allclose(getattr(st_a, f), getattr(st_b, f), rtol=1e-12, atol=1e-12, err_msg=f)


def test_split_evidence_on_observed_members_equals_the_colour_pass():
    from lhvi import c2f
    rng = np.random.default_rng(0)
    V = 400
    values = np.where(rng.random(V) < 0.6, rng.integers(0, 7, V) * 1.5 + rng.integers(0, 2, V) * 0.01, np.nan)
    rvc = rng.integers(0, 9, V).astype(np.int32)
    rvc[np.isnan(values)] = 9 + rng.integers(0, 3, int(np.isnan(values).sum()))
    for eps, use_sqrt in ((0.0, True), (2.0, False), (0.5, True), (100.0, False)):
        want = lifting.split_evidence_colors(values, rvc, 2, 10, eps, use_sqrt=use_sqrt)
        obs = np.flatnonzero(~np.isnan(values))
        oc, nc = c2f.split_evidence_observed(values[obs], rvc[obs], int(rvc.max()) + 1, 2, 10, eps, use_sqrt)
        got = rvc.copy()
        got[obs] = oc
        np.testing.assert_array_equal(got, want)
        assert nc == int(want.max()) + 1


def test_split_evidence_by_tensor_grouping_equals_the_colour_pass():
    """``c2f.split_evidence_tensors`` (grouping by weighted bincounts / unique on the colours' device, k-means on the distinct
    (colour, value) pairs only) against ``lifting.split_evidence_colors`` over all variables: same pieces, same numbering"""
    import torch
    from lhvi import c2f
    rng = np.random.default_rng(1)
    V = 600
    values = np.where(rng.random(V) < 0.6, rng.integers(0, 9, V) * 1.25 + rng.integers(0, 2, V) * 0.01, np.nan)
    rvc = rng.integers(0, 7, V).astype(np.int32)
    rvc[np.isnan(values)] = 7 + rng.integers(0, 3, int(np.isnan(values).sum()))
    obs = np.flatnonzero(~np.isnan(values))
    distinct, code = np.unique(values[obs], return_inverse=True)
    for eps, use_sqrt in ((0.0, True), (2.0, False), (0.5, True), (100.0, False), (0.9, False)):
        want = lifting.split_evidence_colors(values, rvc, 2, 10, eps, use_sqrt=use_sqrt)
        oc, nc = c2f.split_evidence_tensors(torch.from_numpy(values[obs]), torch.from_numpy(code.astype(np.int64)), distinct,
                                            torch.from_numpy(rvc[obs].astype(np.int64)), int(rvc.max()) + 1, 2, 10, eps, use_sqrt)
        got = rvc.copy()
        got[obs] = oc.numpy()
        np.testing.assert_array_equal(got, want)
        assert nc == int(want.max()) + 1
