"""Synthetic model generators for the BASELINE.json configurations.

Small object-graph builders (used by tests / smoke) and vectorised flat-array builders for the large
benchmark graphs, which never materialise per-variable Python objects.
"""
from __future__ import annotations

import numpy as np

from .graph import Domain, F, Graph, RV
from .potentials import LinearGaussianPotential, X2Potential


def gaussian_chain(n=10, coeff=0.9, sig=1.0, evidence=1.5):
    """cfg 1: n-node Gaussian chain; every hidden node also gets a unary X2 prior so that no hidden
    variable has degree 1 (the reference divides by zero there)."""
    d = Domain((-10, 10), continuous=True, integral_points=np.linspace(-10, 10, 32))
    rvs = [RV(d, evidence if i == 0 else None) for i in range(n)]
    lin = LinearGaussianPotential(coeff, sig)
    x2 = X2Potential(1.0, 4.0)
    fs = [F(lin, [rvs[i], rvs[i + 1]]) for i in range(n - 1)]
    fs += [F(x2, [rvs[i]]) for i in range(1, n)]
    g = Graph()
    g.rvs, g.factors = rvs, fs
    g.init_nb()
    return g, rvs


# ---------------------------------------------------------------------------------------------
# flat-array generators (no Python objects per node)
# ---------------------------------------------------------------------------------------------
from .flat import build_flat  # noqa: E402
from . import potentials as _P  # noqa: E402


def _random_pairing(V, deg, rng):
    """random `deg`-regular multigraph by pairing variable stubs; self-pairs are re-drawn"""
    stubs = np.repeat(np.arange(V, dtype=np.int32), deg)
    rng.shuffle(stubs)
    a, b = stubs[0::2].copy(), stubs[1::2].copy()
    bad = a == b
    while bad.any():
        b[bad] = rng.integers(0, V, size=int(bad.sum()))
        bad = a == b
    return a, b


def random_gaussian_mrf(V=20000, deg=4, seed=0, evidence_ratio=0.1):
    """random pairwise Gaussian MRF mixing every closed-form potential kind of GaBP.message_f_to_rv, plus one
    unary X2 prior per variable (keeps precisions positive and degrees >= 2)"""
    rng = np.random.default_rng(seed)
    a, b = _random_pairing(V, deg, rng)
    Fp = a.size
    dom = Domain((-10, 10), continuous=True, integral_points=np.linspace(-10, 10, 32))
    specs = []
    for _ in range(8):    # Gaussian 2x2 with |rho| < 1
        s0, s1 = rng.uniform(4, 12, size=2)
        rho = rng.uniform(-0.7, 0.7)
        specs.append(_P.GaussianPotential([0., 0.], [[s0, rho * np.sqrt(s0 * s1)], [rho * np.sqrt(s0 * s1), s1]])
                     .device_spec((dom, dom)))
    for _ in range(8):
        specs.append((_P.POT_LINEAR_GAUSSIAN, [float(rng.uniform(0.2, 0.9)), float(rng.uniform(1, 3))]))
    for _ in range(8):
        specs.append((_P.POT_XY, [float(rng.uniform(-0.3, 0.3)), float(rng.uniform(1, 3))]))
    x2_first = len(specs)
    for _ in range(4):
        specs.append((_P.POT_X2, [float(rng.uniform(0.5, 1.5)), float(rng.uniform(1, 3))]))
    fac_pot = np.concatenate([rng.integers(0, x2_first, size=Fp), rng.integers(x2_first, len(specs), size=V)])
    fac_ptr = np.concatenate([np.arange(0, 2 * Fp + 1, 2), 2 * Fp + np.arange(1, V + 1)]).astype(np.int32)
    edge_var = np.concatenate([np.stack([a, b], axis=1).ravel(), np.arange(V, dtype=np.int32)])
    value = np.full(V, np.nan)
    ev = rng.random(V) < evidence_ratio
    value[ev] = rng.uniform(-3, 3, size=int(ev.sum()))
    return build_flat(fac_ptr, edge_var, fac_pot, specs, value, np.zeros(V, dtype=np.int32), [dom])


def rgm_flat(C=100, B=50, n_values=0, evidence_ratio=0.2, seed=0):
    """cfg 2 / cfg 5: the RGM template (Demo/Data/RGM/Generator.py:18-33) grounded straight into arrays:
    recession(1) -p1- market(C) -p2- loss(C,B) -p3- revenue(B).  Returns (flat, symmetric, rv_color0, f_color0).
    Evidence on a random subset; values ~U(-30,30), or drawn from ``n_values`` distinct values when
    ``n_values`` > 0 so that colour passing compresses."""
    rng = np.random.default_rng(seed)
    V = 1 + C + C * B + B
    rec, market, loss, revenue = 0, 1, 1 + C, 1 + C + C * B
    c_idx = np.arange(C, dtype=np.int32)
    cb_c = np.repeat(c_idx, B)
    cb_b = np.tile(np.arange(B, dtype=np.int32), C)
    f1 = np.stack([np.full(C, rec, dtype=np.int32), market + c_idx], axis=1)
    f2 = np.stack([market + cb_c, loss + cb_c * B + cb_b], axis=1)
    f3 = np.stack([loss + cb_c * B + cb_b, revenue + cb_b], axis=1)
    edge_var = np.concatenate([f1.ravel(), f2.ravel(), f3.ravel()]).astype(np.int32)
    F = C + 2 * C * B
    fac_ptr = np.arange(0, 2 * F + 1, 2, dtype=np.int32)
    fac_pot = np.concatenate([np.zeros(C), np.ones(C * B), np.full(C * B, 2)]).astype(np.int32)
    dom = Domain((-50, 50), continuous=True, integral_points=np.linspace(-50, 50, 100))
    specs = [_P.GaussianPotential([0., 0.], s).device_spec((dom, dom)) for s in
             ([[10., -7.], [-7., 10.]], [[10., 5.], [5., 10.]], [[10., 7.], [7., 10.]])]
    value = np.full(V, np.nan)
    ev = rng.random(V) < evidence_ratio
    # keep every hidden variable's degree >= 2 (loss nodes have degree 2 already; all others larger)
    if n_values > 0:
        pool = np.round(rng.uniform(-30, 30, size=n_values), 3)
        value[ev] = rng.choice(pool, size=int(ev.sum()))
    else:
        value[ev] = rng.uniform(-30, 30, size=int(ev.sum()))
    flat = build_flat(fac_ptr, edge_var, fac_pot, specs, value, np.zeros(V, dtype=np.int32), [dom])
    # initial colours (CompressedGraph.init_cluster): hidden vs one colour per distinct evidence value
    rv_color = np.zeros(V, dtype=np.int32)
    if ev.any():
        _, inv = np.unique(value[ev], return_inverse=True)
        rv_color[ev] = 1 + inv
    sym = np.zeros(F, dtype=np.uint8)
    return flat, sym, rv_color, fac_pot.copy()


def rgm_structured_flat(C=2000, B=1250, A=400, R=250, distinct=False):
    """cfg 5 (BASELINE.json: "10 M ground / ~10 k lifted clusters"): the RGM template with evidence that is a function of
    (c mod A, b mod R) only -- a third of the market classes and a quarter of the revenue classes observed from small value
    pools, a tenth of the loss atoms observed on a lattice of the two class indices.  Colour passing then converges to a
    partition whose size depends on (A, R) but not on (C, B) as long as A | C and R | B: the defaults give 9 956 rv clusters
    and 19 630 factor clusters at any scale, 10.0 M ground edges at C = 2 000, B = 1 250.
    ``distinct``: every observed market / revenue class and every observed (market class, revenue class) pair of loss atoms gets a
    value of its own (instead of one from a pool of 11 / 9 / 6), so the
    partition has about one loss cluster per (market class, revenue class) pair: A = 2 000, R = 250 lifts the 10 M-edge graph
    to ~1 M edges.  Returns (flat, symmetric, rv_color0, f_color0) like ``rgm_flat``."""
    if C % A or B % R:
        raise ValueError('A must divide C and R must divide B')
    flat, sym, _, f0 = rgm_flat(C=C, B=B, n_values=0, evidence_ratio=0.0, seed=0)
    V = flat.V
    val = np.full(V, np.nan)
    market, loss, revenue = 1, 1 + C, 1 + C + C * B
    c, b = np.arange(C), np.arange(B)
    ac, rb = c % A, b % R
    mo = ac % 3 == 0
    val[market + c[mo]] = ac[mo].astype(float) * 0.01 - 5.0 if distinct else (ac[mo] % 11).astype(float) - 5.0
    ro = rb % 4 == 0
    val[revenue + b[ro]] = rb[ro].astype(float) * 0.02 if distinct else (rb[ro] % 9).astype(float) * 1.5
    acc, rbb = np.repeat(ac, B), np.tile(rb, C)
    lo = (acc * 7 + rbb * 3) % 10 == 0
    val[loss + np.flatnonzero(lo)] = (acc * R + rbb)[lo].astype(float) * 1e-4 - 2.5 if distinct else ((acc + 2 * rbb) % 6)[lo].astype(float) - 2.5
    flat.var_value = val
    rv_color = np.zeros(V, dtype=np.int32)
    ob = ~np.isnan(val)
    _, inv = np.unique(val[ob], return_inverse=True)
    rv_color[ob] = 1 + inv
    return flat, sym, rv_color, f0


def hybrid_mrf_flat(V=250000, deg=4, seed=0, frac_discrete=0.2, evidence_ratio=0.1, T=32, pool=64):
    """cfg 4 (and, at V=2.5M, the 10M-edge headline graph): random sparse hybrid pairwise MRF.

    80 % continuous variables on [-10, 10] (T integral points), 20 % binary; `deg`-regular random pairing;
    potentials by scope type, coefficients ~U(0.2, 1):
      (cont, cont)  QuadraticPotential  exp(-a x^2 - b y^2 + c x y + ...), |c| < sqrt(ab)  (XY / X2 style)
      (disc, cont)  HybridQuadraticPotential, one (A, b, c) per state, discrete argument first
      (disc, disc)  2x2 TablePotential
    10 % of the variables are observed."""
    rng = np.random.default_rng(seed)
    dc = Domain((-10, 10), continuous=True, integral_points=np.linspace(-10, 10, T))
    db = Domain((0, 1))
    is_disc = rng.random(V) < frac_discrete
    a, b = _random_pairing(V, deg, rng)
    # discrete argument first for mixed scopes
    swap = ~is_disc[a] & is_disc[b]
    a, b = np.where(swap, b, a), np.where(swap, a, b)
    F = a.size
    specs = []
    for _ in range(pool):
        p, q = rng.uniform(0.2, 1.0, size=2) * 0.5
        c = rng.uniform(-0.9, 0.9) * np.sqrt(p * q)
        specs.append((_P.POT_QUADRATIC, [2.0, -p, c / 2, c / 2, -q, rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), 0.0]))
    for _ in range(pool):
        A = -rng.uniform(0.2, 1.0, size=2) * 0.5
        specs.append((_P.POT_HYBRID_QUADRATIC, [1.0, 1.0, 2.0] + A.tolist() + rng.uniform(-1, 1, size=2).tolist()
                      + rng.uniform(-0.5, 0.5, size=2).tolist()))
    for _ in range(pool):
        specs.append((_P.POT_TABLE, [2.0, 2.0, 2.0] + rng.uniform(0.2, 1.0, size=4).tolist()))
    kind = np.where(is_disc[a] & is_disc[b], 2, np.where(is_disc[a], 1, 0))
    fac_pot = (kind * pool + rng.integers(0, pool, size=F)).astype(np.int32)
    fac_ptr = np.arange(0, 2 * F + 1, 2, dtype=np.int32)
    edge_var = np.stack([a, b], axis=1).ravel().astype(np.int32)
    value = np.full(V, np.nan)
    ev = rng.random(V) < evidence_ratio
    value[ev & ~is_disc] = rng.uniform(-5, 5, size=int((ev & ~is_disc).sum()))
    value[ev & is_disc] = rng.integers(0, 2, size=int((ev & is_disc).sum()))
    return build_flat(fac_ptr, edge_var, fac_pot, specs, value, is_disc.astype(np.int32), [dc, db])



def paper_popularity_flat(P=300, T=10, seed=0, points=32):
    """cfg 3 at any size, on arrays: the paper-popularity hybrid MLN (Demo/Data/HMLN/GeneratorPaperPopularity.py:7-40: atoms,
    the three weighted formulas, the t1 != t2 constraint) grounded by ``RelationalGraph.ground_flat`` with the evidence pattern
    of its ``generate_data`` (:51-72: 70 % of the paper / topic popularities ~ U(0, 10); for 70 % of the papers a random
    subset of their PaperIn atoms; SameSession pairs) and the demo's domain ``Domain((-15, 15), integral_points =
    linspace(0, 10, points))``.  Returns (flat, keys)."""
    from .mln import MLNPotential, eq_op
    from .relational import LV, Atom, ParamF, RelationalGraph
    rng = np.random.default_rng(seed)
    dom_b = Domain((0, 1))
    dom_r = Domain((-15, 15), continuous=True, integral_points=np.linspace(0, 10, points))
    lvp, lvt = LV(['p%d' % i for i in range(P)]), LV(['t%d' % i for i in range(T)])
    atoms = (Atom(dom_b, (lvt, lvt), 'SameSession'), Atom(dom_b, (lvp, lvt), 'PaperIn'),
             Atom(dom_r, (lvt,), 'TopicPopularity'), Atom(dom_r, (lvp,), 'PaperPopularity'))
    differ = lambda s: s['t1'] != s['t2']
    differ.vectorized = True
    pfs = (ParamF(MLNPotential(lambda x: eq_op(x[0], 1), w=0.3), nb=['PaperPopularity(p)']),
           ParamF(MLNPotential(lambda x: x[0] * eq_op(x[1], x[2]), w=0.5),
                  nb=['SameSession(t1,t2)', 'TopicPopularity(t1)', 'TopicPopularity(t2)'], constrain=differ),
           ParamF(MLNPotential(lambda x: x[0] * eq_op(x[1], x[2]), w=1),
                  nb=['PaperIn(p,t)', 'PaperPopularity(p)', 'TopicPopularity(t)']))
    flat, keys = RelationalGraph(atoms, pfs).ground_flat(None)
    value = np.full(flat.V, np.nan)

    def observe(name, lin, vals):
        v = keys.var_ids(name, lin)
        value[v[v >= 0]] = np.asarray(vals, dtype=np.float64)[v >= 0]
    pp = rng.choice(P, int(P * 0.7), replace=False)
    observe('PaperPopularity', pp, rng.uniform(0, 10, pp.size))
    tp = rng.choice(T, int(T * 0.7), replace=False)
    observe('TopicPopularity', tp, rng.uniform(0, 10, tp.size))
    papers = rng.choice(P, int(P * 0.7), replace=False)
    cnt = rng.integers(0, T, papers.size)                       # np.random.randint(num_topic) topics per chosen paper
    mask = np.argsort(rng.random((papers.size, T)), axis=1) < cnt[:, None]      # a random subset of that size per row
    pi, ti = np.nonzero(mask)
    observe('PaperIn', papers[pi] * T + ti, rng.integers(0, 2, pi.size))
    t1 = np.repeat(np.arange(T), T // 2)
    t2 = rng.integers(0, T, t1.size)
    keep = t1 != t2
    observe('SameSession', t1[keep] * T + t2[keep], rng.integers(0, 2, int(keep.sum())))
    flat.var_value = value
    return flat, keys


def concat_flats(flats):
    """disjoint union of ground ``FlatGraph``s that share their potential table and domains (e.g. one relational template
    grounded several times with different evidence): variables, factors and edges of copy k follow those of copy k - 1"""
    import copy
    f0 = flats[0]
    for f in flats[1:]:
        if not (np.array_equal(f.pot_param, f0.pot_param) and np.array_equal(f.pot_off, f0.pot_off)
                and np.array_equal(f.dom_val, f0.dom_val) and np.array_equal(f.dom_ptr, f0.dom_ptr)) or f.lifted:
            raise ValueError('concat_flats needs ground graphs over the same potentials and domains')
    vo = np.cumsum([0] + [f.V for f in flats])
    fo = np.cumsum([0] + [f.F for f in flats])
    eo = np.cumsum([0] + [f.E for f in flats])
    cat = lambda name, off=None, tail=False: np.concatenate(
        [(getattr(f, name)[1:] if tail and k else getattr(f, name)) + (0 if off is None else off[k]) for k, f in enumerate(flats)]
    ).astype(getattr(f0, name).dtype)
    out = copy.copy(f0)
    out.__dict__.pop('_view_cache', None)
    out.V, out.F, out.E = int(vo[-1]), int(fo[-1]), int(eo[-1])
    out.fac_ptr, out.var_ptr = cat('fac_ptr', eo, True), cat('var_ptr', eo, True)
    out.edge_var, out.edge_fac, out.edge_pos = cat('edge_var', vo), cat('edge_fac', fo), cat('edge_pos')
    out.edge_canon, out.var_edge = cat('edge_canon', eo), cat('var_edge', eo)
    for name in ('edge_count', 'fac_pot', 'var_value', 'var_dom', 'var_mult', 'fac_mult'):
        setattr(out, name, cat(name))
    out.rvs, out.factors, out.var_index, out.fac_index = [], [], {}, {}
    return out


def paper_popularity_copies(copies, P=300, T=10, seed=0, points=32):
    """`copies` independent groundings of the paper-popularity HMLN at its reference size (cfg 3: 300 papers x 10 topics,
    E = 9 570 each), each with its own evidence draw -- the "same graph x k" scaling BASELINE.json uses for cfg 4, which keeps
    every variable's degree (and with it the range of the log messages) what the reference's own model has"""
    return concat_flats([paper_popularity_flat(P, T, seed + k, points)[0] for k in range(copies)])
