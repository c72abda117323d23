"""Object graph -> flat CSR arrays (the layout the HIP kernels read).

Edges are (factor, argument position) incidences stored factor-major, so the edges of factor ``f``
are ``fac_ptr[f] .. fac_ptr[f+1]-1`` in argument order.  The variable side is a CSR over those edge
ids in ``rv.nb`` order -- the order the reference's Python loops visit neighbours
(``GaBP.py:23``, ``EPBPLogVersion.py:171``), which fixes the floating-point summation order.

Works for a ground ``Graph`` and for a lifted ``CompressedGraph`` alike: anything exposing ``rvs`` and
``factors`` whose members have ``nb``, ``value``, ``domain`` / ``potential`` (and ``count`` when lifted).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from .potentials import MAX_ARITY, POT_GENERIC


@dataclass
class FlatGraph:
    V: int
    F: int
    E: int
    fac_ptr: np.ndarray        # int32 [F+1]
    edge_var: np.ndarray       # int32 [E]   variable of edge e
    edge_fac: np.ndarray       # int32 [E]   factor of edge e
    edge_pos: np.ndarray       # int32 [E]   argument position of e in its factor
    edge_canon: np.ndarray     # int32 [E]   canonical edge of the (factor, variable) pair (== e unless the
                               #             factor's scope repeats a lifted cluster)
    var_ptr: np.ndarray        # int32 [V+1]
    var_edge: np.ndarray       # int32 [nnz] canonical edge ids, rv.nb order
    edge_count: np.ndarray     # float64 [E] lifted multiplicity rv.count[f] (1 on a ground graph)
    lifted: bool
    fac_pot: np.ndarray        # int32 [F]   index into the potential table
    pot_kind: np.ndarray       # int32 [P]
    pot_off: np.ndarray        # int32 [P+1] offsets into pot_param
    pot_param: np.ndarray      # float64
    var_value: np.ndarray      # float64 [V]  NaN = hidden
    var_dom: np.ndarray        # int32 [V]   domain id
    var_mult: np.ndarray       # float64 [V]  |cluster| (1 on a ground graph)
    fac_mult: np.ndarray       # float64 [F]
    dom_cont: np.ndarray       # int32 [D]   1 = continuous
    dom_lo: np.ndarray         # float64 [D]
    dom_hi: np.ndarray         # float64 [D]
    dom_ptr: np.ndarray        # int32 [D+1] into dom_val: states (discrete) or integral points (continuous)
    dom_val: np.ndarray        # float64
    rvs: list = field(repr=False, default_factory=list)
    factors: list = field(repr=False, default_factory=list)
    var_index: dict = field(repr=False, default_factory=dict)
    fac_index: dict = field(repr=False, default_factory=dict)
    potentials: list = field(repr=False, default_factory=list)
    domains: list = field(repr=False, default_factory=list)

    # ---- derived helpers -------------------------------------------------------------------
    # The three per-variable views below are cached per source array object (callers index them inside loops; evaluating
    # an O(V) expression per access made such loops quadratic).  Replace `var_value` / `var_dom` wholesale to change
    # them, as all code here does -- in-place edits would not be seen.
    def _cached(self, name, sources, fn):
        cache = self.__dict__.setdefault('_view_cache', {})
        hit = cache.get(name)
        if hit is None or any(a is not b for a, b in zip(hit[0], sources)):
            hit = (tuple(sources), fn())
            cache[name] = hit
        return hit[1]

    @property
    def var_hidden(self):
        return self._cached('hidden', (self.var_value,), lambda: np.isnan(self.var_value))

    @property
    def var_cont(self):
        return self._cached('cont', (self.dom_cont, self.var_dom), lambda: self.dom_cont[self.var_dom].astype(bool))

    @property
    def var_nstates(self):
        """number of tabulation points a domain owns: states (discrete) or grid points (continuous)"""
        return self._cached('nstates', (self.dom_ptr, self.var_dom),
                            lambda: (self.dom_ptr[1:] - self.dom_ptr[:-1])[self.var_dom])

    def edge_of(self, f, rv):
        """canonical edge id of the (factor object, variable object) pair"""
        fi, vi = self.fac_index[f], self.var_index[rv]
        for e in range(self.fac_ptr[fi], self.fac_ptr[fi + 1]):
            if self.edge_var[e] == vi:
                return int(self.edge_canon[e])
        raise KeyError((f, rv))


def ground_order(items):
    """the order in which a graph's rvs / factors are numbered: a list as it stands; a Python ``set`` (what the reference's
    ``Graph`` / ``RelationalGraph.ground_graph`` hold) sorted by node id, i.e. in creation order -- iterating the set itself would
    number the ground graph by object hashes, differently in every process"""
    if not isinstance(items, (set, frozenset)):
        return list(items)
    try:
        return sorted(items, key=_node_id)        # (the nodes' own __lt__ compares ids: the same order, without a Python call per comparison)
    except AttributeError:
        return sorted(items)


_node_id = __import__('operator').attrgetter('id')


def _value_or_nan(v):
    return np.nan if v is None else float(v)


class gc_paused:
    """no cyclic garbage collection while a builder allocates its tens of thousands of small containers (the collector's
    generation-2 passes over the ground graph's objects cost more than the builder itself: 57 of 85 ms of ``build_lifted_objects``
    on the 300 x 10 paper-popularity model were spent in them); nothing built here is garbage"""

    def __enter__(self):
        import gc
        self.was = gc.isenabled()
        gc.disable()

    def __exit__(self, *exc):
        import gc
        if self.was:
            gc.enable()


def flatten(g, require_device_potentials=False):
    """Build a ``FlatGraph`` from ``g`` (see ``_flatten``); the garbage collector rests meanwhile."""
    if isinstance(g, FlatGraph):
        return g
    if getattr(g, 'array_flat', False):
        # a compressed graph whose partition is stable (lifting.CompressedGraph after run / the solvers' colour-passing loops):
        # the lifted arrays straight from the ground arrays and the colours, the cluster objects only attached
        lf = g.lifted_flat(require_device_potentials)
        if lf is not None:
            return lf
    with gc_paused():
        return _flatten(g, require_device_potentials)


def _flatten(g, require_device_potentials=False):
    """Build a ``FlatGraph`` from ``g``.

    ``require_device_potentials``: raise if a potential has no device encoding (particle BP and VI
    evaluate potentials on the GPU; Gaussian BP only needs the closed-form kinds and maps anything
    else to the vacuous message like the reference does, ``GaBP.py:138``).  A ``FlatGraph`` (e.g. from
    ``RelationalGraph.ground_flat`` or ``build_flat``) is passed through unchanged.
    """
    if isinstance(g, FlatGraph):
        return g
    rvs = ground_order(g.rvs)
    factors = ground_order(g.factors)
    var_index = {rv: i for i, rv in enumerate(rvs)}
    fac_index = {f: i for i, f in enumerate(factors)}
    V, F = len(rvs), len(factors)
    lifted = V > 0 and hasattr(rvs[0], 'count') and hasattr(rvs[0], 'rvs')

    # domains (identity keyed, like the reference's colour table)
    domains, dom_index = [], {}
    var_dom = np.zeros(V, dtype=np.int32)
    for i, rv in enumerate(rvs):
        d = rv.domain
        if id(d) not in dom_index:
            dom_index[id(d)] = len(domains)
            domains.append(d)
        var_dom[i] = dom_index[id(d)]
    D = len(domains)
    dom_cont = np.array([1 if d.continuous else 0 for d in domains], dtype=np.int32).reshape(D)
    dom_lo = np.array([float(d.values[0]) if d.continuous else 0.0 for d in domains], dtype=np.float64).reshape(D)
    dom_hi = np.array([float(d.values[1]) if d.continuous else 0.0 for d in domains], dtype=np.float64).reshape(D)
    dom_ptr = np.zeros(D + 1, dtype=np.int32)
    vals = []
    for k, d in enumerate(domains):
        pts = np.asarray(d.integral_points if d.continuous else d.values, dtype=np.float64).ravel()
        vals.append(pts)
        dom_ptr[k + 1] = dom_ptr[k] + pts.size
    dom_val = np.concatenate(vals) if vals else np.zeros(0)

    # edges, factor-major.  The per-edge Python work is one dictionary lookup; duplicates (a lifted factor whose scope repeats a
    # cluster) and the variable rows are resolved on arrays: an object graph of 20 000 edges is flattened twice per lifted run
    # (the ground graph for the refinement, the lifted graph for the sweep), and element-wise NumPy stores made that 2 x 27 ms
    nbs = [f.nb for f in factors]
    arity = np.fromiter((len(nb) for nb in nbs), dtype=np.int64, count=F)
    if F and arity.max() > MAX_ARITY:
        raise NotImplementedError('factor arity %d exceeds LHVI_MAX_ARITY=%d' % (arity.max(), MAX_ARITY))
    fac_ptr = np.zeros(F + 1, dtype=np.int32)
    np.cumsum(arity, out=fac_ptr[1:])
    E = int(fac_ptr[-1])
    edge_var = np.fromiter((var_index[rv] for nb in nbs for rv in nb), dtype=np.int32, count=E)
    edge_fac = np.repeat(np.arange(F, dtype=np.int32), arity)
    edge_pos = (np.arange(E, dtype=np.int64) - fac_ptr[:-1].astype(np.int64)[edge_fac]).astype(np.int32)
    # canonical edge of a (factor, variable) pair = its first edge
    pair_key = edge_fac.astype(np.int64) * max(V, 1) + edge_var
    uniq_key, first_edge, inverse = np.unique(pair_key, return_index=True, return_inverse=True)
    edge_canon = first_edge[inverse].astype(np.int32)
    if not lifted and uniq_key.size != E:
        e_dup = int(np.flatnonzero(edge_canon != np.arange(E))[0])
        raise NotImplementedError('ground factor #%d repeats a variable in its scope' % int(edge_fac[e_dup]))

    # variable CSR in rv.nb order
    vnbs = [rv.nb for rv in rvs]
    deg = np.fromiter((len(nb) for nb in vnbs), dtype=np.int64, count=V)
    var_ptr = np.zeros(V + 1, dtype=np.int32)
    np.cumsum(deg, out=var_ptr[1:])
    nnz = int(var_ptr[-1])
    row_fac = np.fromiter((fac_index[f] for nb in vnbs for f in nb), dtype=np.int64, count=nnz)
    row_key = row_fac * max(V, 1) + np.repeat(np.arange(V, dtype=np.int64), deg)
    at = np.searchsorted(uniq_key, row_key)
    if nnz and (at.max(initial=0) >= uniq_key.size or (uniq_key[np.minimum(at, uniq_key.size - 1)] != row_key).any()):
        bad = int(np.flatnonzero((at >= uniq_key.size) | (uniq_key[np.minimum(at, uniq_key.size - 1)] != row_key))[0])
        raise KeyError((int(row_fac[bad]), int(np.repeat(np.arange(V), deg)[bad])))       # rv.nb names a factor whose scope lacks rv
    var_edge = first_edge[at].astype(np.int32).reshape(-1)
    edge_count = np.ones(E, dtype=np.float64)
    if lifted:
        edge_count[var_edge] = np.fromiter((rv.count[f] for rv, nb in zip(rvs, vnbs) for f in nb), dtype=np.float64, count=nnz)
        edge_count = edge_count[edge_canon]          # alias edges carry their canonical count so kernels can read either

    # potentials
    potentials, pot_index = [], {}
    pot_kind, pot_off, params = [], [0], []
    fac_pot = np.zeros(F, dtype=np.int32)
    for fi, (f, nb) in enumerate(zip(factors, nbs)):
        key = (id(f.potential),) + tuple(id(rv.domain) for rv in nb)
        if key not in pot_index:
            doms = tuple(rv.domain for rv in nb)
            spec = getattr(f.potential, 'device_spec', None)
            if spec is None:
                if require_device_potentials:
                    raise NotImplementedError(
                        'potential %r has no device encoding (device_spec); refusing to fall back to the CPU'
                        % type(f.potential).__name__)
                kind, par = POT_GENERIC, []
            else:
                kind, par = spec(doms)
            pot_index[key] = len(potentials)
            potentials.append(f.potential)
            pot_kind.append(kind)
            params.extend(par)
            pot_off.append(len(params))
        fac_pot[fi] = pot_index[key]

    var_value = np.array([_value_or_nan(rv.value) for rv in rvs], dtype=np.float64).reshape(V)
    var_mult = np.array([float(len(rv.rvs)) if lifted else 1.0 for rv in rvs], dtype=np.float64).reshape(V)
    fac_mult = np.array([float(len(f.factors)) if lifted else 1.0 for f in factors], dtype=np.float64).reshape(F)

    return FlatGraph(
        V=V, F=F, E=E, fac_ptr=fac_ptr, edge_var=edge_var, edge_fac=edge_fac, edge_pos=edge_pos,
        edge_canon=edge_canon, var_ptr=var_ptr, var_edge=var_edge, edge_count=edge_count, lifted=lifted,
        fac_pot=fac_pot, pot_kind=np.array(pot_kind, dtype=np.int32).reshape(-1),
        pot_off=np.array(pot_off, dtype=np.int32), pot_param=np.array(params, dtype=np.float64).reshape(-1),
        var_value=var_value, var_dom=var_dom, var_mult=var_mult, fac_mult=fac_mult,
        dom_cont=dom_cont, dom_lo=dom_lo, dom_hi=dom_hi, dom_ptr=dom_ptr, dom_val=dom_val,
        rvs=rvs, factors=factors, var_index=var_index, fac_index=fac_index,
        potentials=potentials, domains=domains)


def build_flat(fac_ptr, edge_var, fac_pot, pot_specs, var_value, var_dom, domains):
    """Vectorised ``FlatGraph`` constructor for ground graphs given as arrays (no per-node Python objects).

    ``pot_specs``: list of (kind, params) rows; ``domains``: list of ``Domain`` objects (for grids / states).
    The variable CSR lists a variable's incident edges in ascending factor order, which is what
    ``Graph.init_nb`` produces (``Graph.py:148-152``).
    """
    fac_ptr = np.asarray(fac_ptr, dtype=np.int32)
    edge_var = np.asarray(edge_var, dtype=np.int32)
    F, E, V = fac_ptr.size - 1, edge_var.size, int(np.asarray(var_value).size)
    arity = np.diff(fac_ptr)
    edge_fac = np.repeat(np.arange(F, dtype=np.int32), arity)
    edge_pos = (np.arange(E, dtype=np.int64) - fac_ptr[:-1][edge_fac]).astype(np.int32)
    order = np.argsort(edge_var, kind='stable').astype(np.int32)
    var_ptr = np.zeros(V + 1, dtype=np.int32)
    np.cumsum(np.bincount(edge_var, minlength=V), out=var_ptr[1:])
    pot_kind = np.array([k for k, _ in pot_specs], dtype=np.int32)
    pot_off = np.zeros(len(pot_specs) + 1, dtype=np.int32)
    np.cumsum([len(p) for _, p in pot_specs], out=pot_off[1:])
    pot_param = np.array([x for _, p in pot_specs for x in p], dtype=np.float64).reshape(-1)
    D = len(domains)
    dom_ptr = np.zeros(D + 1, dtype=np.int32)
    vals = []
    for k, d in enumerate(domains):
        pts = np.asarray(d.integral_points if d.continuous else d.values, dtype=np.float64).ravel()
        vals.append(pts)
        dom_ptr[k + 1] = dom_ptr[k] + pts.size
    return FlatGraph(
        V=V, F=F, E=E, fac_ptr=fac_ptr, edge_var=edge_var, edge_fac=edge_fac, edge_pos=edge_pos,
        edge_canon=np.arange(E, dtype=np.int32), var_ptr=var_ptr, var_edge=order,
        edge_count=np.ones(E), lifted=False, fac_pot=np.asarray(fac_pot, dtype=np.int32),
        pot_kind=pot_kind, pot_off=pot_off, pot_param=pot_param,
        var_value=np.asarray(var_value, dtype=np.float64), var_dom=np.asarray(var_dom, dtype=np.int32),
        var_mult=np.ones(V), fac_mult=np.ones(F),
        dom_cont=np.array([1 if d.continuous else 0 for d in domains], dtype=np.int32),
        dom_lo=np.array([float(d.values[0]) if d.continuous else 0.0 for d in domains]),
        dom_hi=np.array([float(d.values[1]) if d.continuous else 0.0 for d in domains]),
        dom_ptr=dom_ptr, dom_val=np.concatenate(vals) if vals else np.zeros(0), domains=list(domains))
