"""Colour passing (lifting): ``CompressedGraph`` with the reference's object surface, refined on the GPU.

Reference: ``CompressedGraphWithObs.py:8-271`` (used by GaLBP / HybridLBP / LiftedVarInference) and
``CompressedGraphSorted.py:12-209`` (deterministic ids, ``sharing_count``, ``rvs_list``).  SURVEY.md note
N1 asks for the union of both surfaces on one object, which ``SuperRV`` / ``SuperF`` here provide.

Split of labour
  host   : initial colours (domain identity / hidden-vs-evidence / evidence value; potential ``__eq__``),
           building ``SuperRV`` / ``SuperF`` objects from the final colour arrays;
  device : every refinement half-round (``lhvi_color_refine_factors`` / ``lhvi_color_refine_rvs``,
           ``csrc/color.hip``) on the ground graph's CSR arrays, colours resident in HBM.
Only the induced partition is defined by the reference (cluster objects carry no stable ids), so new
colours are dense ranks; ``SuperRV.id`` / ``SuperF.id`` are those ranks.
"""
from __future__ import annotations

from collections import Counter

import numpy as np

from . import _abi
from .flat import flatten, ground_order


class SuperRV:
    """Cluster of ground rvs (``CompressedGraphWithObs.py:8-45``; ids as in ``CompressedGraphSorted.py:13-32``)."""

    def __init__(self, rvs, domain=None, value=None, cid=-1, order=None):
        self.rvs = rvs
        self._order = order                      # the members in ground order (None: sorted by node id)
        first = next(iter(rvs))
        self.domain = first.domain if domain is None else domain
        self.value = self.get_value(rvs if order is None else order) if value is None and first.value is not None else value
        self._variance = None                    # (np.var of the members' values, formed when first asked for)
        self.nb = None
        self.N = 0
        self.count = None
        self.id = cid
        for rv in rvs:
            rv.cluster = self

    def __lt__(self, other):
        return self.id < other.id

    @staticmethod
    def get_value(rvs):
        """running sum of the members' values over the size (CGWO:24-28).  The reference walks a Python ``set`` of RV objects, i.e.
        in an order that changes from process to process, and the last bit of the mean with it; here the members are walked in
        ground order (node ids), which is also the order of the array path (``lifting.segment_sums``) -- the two paths then give
        a cluster the same value bit for bit."""
        total = 0
        for rv in (sorted(rvs) if isinstance(rvs, (set, frozenset)) else rvs):
            total += rv.value
        return total / len(rvs)

    def get_variance(self):
        return np.var(tuple(rv.value for rv in (sorted(self.rvs) if self._order is None else self._order)))

    @property
    def variance(self):
        """``SuperRV.variance`` (CGWO:22,38-39): None for a hidden cluster"""
        if self.value is None:
            return None
        if self._variance is None:
            self._variance = self.get_variance()
        return self._variance

    @variance.setter
    def variance(self, v):
        self._variance = v

    @property
    def sharing_count(self):
        return len(self.rvs)

    @property
    def domain_type(self):
        return next(iter(self.rvs)).domain_type

    @property
    def dstates(self):
        return next(iter(self.rvs)).dstates

    @property
    def values(self):
        return next(iter(self.rvs)).values

    def update_nb(self, representative=None):
        rv = representative if representative is not None else (min(self.rvs) if self._order is None else self._order[0])
        self.count = Counter(f.cluster for f in rv.nb)
        self.nb = tuple(self.count)
        self.N = rv.N

    def __repr__(self):
        return 'SuperRV#%d(%d rvs)' % (self.id, len(self.rvs))


class SuperF:
    """Cluster of ground factors (``CompressedGraphWithObs.py:133-150``)."""

    def __init__(self, factors, cid=-1, order=None):
        self.factors = factors
        self._order = order
        first = next(iter(factors))
        self.potential = first.potential
        self.log_potential_fun = getattr(first, 'log_potential_fun', None)
        self.nb = None
        self.id = cid
        for f in factors:
            f.cluster = self

    def __lt__(self, other):
        return self.id < other.id

    @property
    def sharing_count(self):
        return len(self.factors)

    @property
    def domain_type(self):
        return next(iter(self.factors)).domain_type

    @property
    def nb_domain_types(self):
        return next(iter(self.factors)).nb_domain_types

    def update_nb(self, representative=None):
        f = representative if representative is not None else (min(self.factors) if self._order is None else self._order[0])
        self.nb = tuple(rv.cluster for rv in f.nb)

    def __repr__(self):
        return 'SuperF#%d(%d factors)' % (self.id, len(self.factors))


def initial_colors(g, is_split_cont_evidence=True):
    """Host side of ``CompressedGraph.init_cluster`` (``CompressedGraphWithObs.py:187-234``) as dense ids.

    rvs: one colour per (Domain object, hidden) and per (Domain object, evidence value) -- or one per
    (Domain object, 'evidence') for continuous domains when ``is_split_cont_evidence`` is False.
    factors: one colour per potential under the potential's own ``__hash__`` / ``__eq__``.
    """
    rvs, factors = ground_order(g.rvs), ground_order(g.factors)
    table, rv_color = {}, np.zeros(len(rvs), dtype=np.int32)
    for i, rv in enumerate(rvs):
        if rv.value is None:
            key = (id(rv.domain), 'h')
        elif not is_split_cont_evidence and rv.domain.continuous:
            key = (id(rv.domain), 'e')
        else:
            key = (id(rv.domain), 'v', rv.value)
        rv_color[i] = table.setdefault(key, len(table))
    ptable, f_color = {}, np.zeros(len(factors), dtype=np.int32)
    for i, f in enumerate(factors):
        f_color[i] = ptable.setdefault(f.potential, len(ptable))
    return rv_color, f_color


def initial_colors_flat(flat, is_split_cont_evidence=True):
    """``initial_colors`` for a ``FlatGraph`` without rv / factor objects (``ground_flat``, ``build_flat``), vectorised:
    variables by (domain id, hidden) / (domain id, evidence value), factors by their row of the potential table -- plus the
    per-factor ``symmetric`` flags ``refine_flat`` wants.  ``flat.potentials`` (kept by ``ground_flat``) supplies
    ``symmetric`` and the potentials' own ``__eq__`` where they define one; otherwise table rows are distinct potentials."""
    hidden = flat.var_hidden
    dom = flat.var_dom.astype(np.int64)
    val = np.where(hidden, 0.0, flat.var_value)
    if not is_split_cont_evidence:
        val = np.where(flat.var_cont, 0.0, val)
    # colours numbered in order of first appearance, like the reference's dict: hash-based factorisation (O(V), no sort), first
    # of the values, then of the (value code, domain, observed) triples
    try:                                                   # pandas (optional): O(V) hash factorisation; NumPy: a sort
        import pandas as pd
        factorize = lambda a: pd.factorize(a)[0]
    except ImportError:
        def factorize(a):
            _, first, inv = np.unique(a, return_index=True, return_inverse=True)
            rank = np.empty(first.size, dtype=np.int64)
            rank[np.argsort(first, kind='stable')] = np.arange(first.size)       # renumbered by first appearance
            return rank[inv]
    vcode = factorize(val + 0.0).astype(np.int64)          # (+ 0.0: -0.0 and 0.0 are one evidence value)
    nd = int(dom.max()) + 1 if dom.size else 1
    rv_color = factorize((vcode * nd + dom) * 2 + (~hidden)).astype(np.int32)
    pots = list(getattr(flat, 'potentials', []) or [])
    row_color = np.arange(int(flat.pot_kind.size), dtype=np.int32)
    sym_row = np.zeros(int(flat.pot_kind.size), dtype=np.uint8)
    if len(pots) == flat.pot_kind.size:
        table = {}
        for i, p in enumerate(pots):
            row_color[i] = table.setdefault(p, len(table))
            sym_row[i] = 1 if getattr(p, 'symmetric', False) else 0
    f_color = row_color[flat.fac_pot].astype(np.int32)
    return rv_color, f_color, sym_row[flat.fac_pot]


def _by_first_appearance(keys):
    """codes of `keys` (a 1-d tensor) numbered in order of first appearance, and their number"""
    import torch
    uniq, inv = torch.unique(keys, return_inverse=True)
    n = int(uniq.numel())
    first = first_members(inv.to(torch.int32), n, int(keys.numel()))
    rank = torch.empty(n, dtype=torch.int64, device=keys.device)
    rank[torch.argsort(first)] = torch.arange(n, device=keys.device)
    return rank[inv], n


def initial_colors_device(flat, tg, is_split_cont_evidence=True):
    """``initial_colors_flat`` on `tg`'s device (``DeviceGraph`` / ``TensorGraph`` of `flat`): the same colour ids -- numbered by
    first appearance -- as tensors (rv colours int32, factor colours int32, symmetric flags uint8); what is computed on the host is
    the table of potential rows (``flat.potentials``: their ``__eq__`` and ``symmetric``), which has one entry per distinct
    potential, not per factor.  At 10 M edges the host version is 39 ms of the coarse-to-fine run; this one is a handful of
    launches."""
    import torch
    dev = tg.device
    value = tg.t['var_value']
    as_t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    dom = (tg.t['var_dom'] if 'var_dom' in tg.t else as_t(flat.var_dom)).long()
    hidden = torch.isnan(value)
    val = torch.where(hidden, torch.zeros_like(value), value)
    if not is_split_cont_evidence:
        cont = (tg.t['dom_cont'] if 'dom_cont' in tg.t else as_t(flat.dom_cont)).long()[dom] != 0
        val = torch.where(cont, torch.zeros_like(val), val)
    vcode, _ = _by_first_appearance(val + 0.0)             # (+ 0.0: -0.0 and 0.0 are one evidence value)
    nd = int(flat.var_dom.max()) + 1 if flat.var_dom.size else 1
    rv_color, _ = _by_first_appearance((vcode * nd + dom) * 2 + (~hidden).long())
    pots = list(getattr(flat, 'potentials', []) or [])
    row_color = np.arange(int(flat.pot_kind.size), dtype=np.int32)
    sym_row = np.zeros(int(flat.pot_kind.size), dtype=np.uint8)
    if len(pots) == flat.pot_kind.size:
        table = {}
        for i, p in enumerate(pots):
            row_color[i] = table.setdefault(p, len(table))
            sym_row[i] = 1 if getattr(p, 'symmetric', False) else 0
    fac_pot = (tg.t['fac_pot'] if 'fac_pot' in tg.t else as_t(flat.fac_pot)).long()
    return rv_color.to(torch.int32), as_t(row_color)[fac_pot], as_t(sym_row)[fac_pot]


def kmeans_assign(vals, k=2, iteration=10, order=None):
    """the k-means of ``SuperRV.split_by_evidence`` (``CompressedGraphWithObs.py:78-130``) on the member values of one
    evidence cluster: distinct values with multiplicities, centroids seeded with the first k distinct values, ``iteration``
    Lloyd rounds, members assigned to the nearest final centroid (first on ties).  Returns the piece index of every member,
    or None when the cluster cannot be split (a single member, or fewer than two distinct values).

    The reference seeds the centroids in the iteration order of a Python ``set`` of RV objects, i.e. in an order that
    changes from run to run; here the order is the order of ``vals`` (ground-variable order), which makes the split
    deterministic (identical to the reference whenever the outcome does not depend on the seeding, e.g. when a cluster
    holds at most k distinct values).  ``order``: a permutation of the members to walk them in instead -- a test that has
    recorded the reference's set order for this cluster replays it through this argument."""
    vals = np.asarray(vals, dtype=np.float64)
    if vals.size <= 1:
        return None
    distinct, counts = [], {}
    for v in (vals if order is None else vals[np.asarray(order)]).tolist():   # insertion order = ground order
        if v not in counts:
            distinct.append(v)
            counts[v] = 0
        counts[v] += 1
    kk = min(k, len(distinct))
    if kk <= 1:
        return None
    centroids = np.array(distinct[:kk], dtype=np.float64)
    table = np.zeros((kk, 2))
    for _ in range(iteration):
        for v in distinct:
            idx = int(np.abs(centroids - v).argmin())
            table[idx, 0] += v * counts[v]
            table[idx, 1] += counts[v]
        for idx in range(kk):
            centroids[idx] = table[idx, 0] / table[idx, 1]
        table.fill(0)
    return np.array([int(np.abs(centroids - v).argmin()) for v in vals.tolist()])


def split_evidence_colors(values, rv_color, k=2, iteration=10, epsilon=0.0, use_sqrt=True):
    """``SuperRV.split_by_evidence`` (``CompressedGraphWithObs.py:78-130``, ``kmeans_assign``) applied to every evidence
    cluster whose spread exceeds ``epsilon`` -- ``sqrt(variance) > epsilon`` in ``CompressedGraph.split_evidence``
    (CGWO:236-247), ``variance > epsilon`` in ``HybridLBP.split_evidence`` (HLBP:250-266).  Piece 0 keeps the old colour,
    the others get fresh colours."""
    values = np.asarray(values, dtype=np.float64)
    rv_color = np.array(rv_color, dtype=np.int32)
    next_color = int(rv_color.max()) + 1 if rv_color.size else 0
    observed = ~np.isnan(values)
    for c in np.unique(rv_color[observed]):
        members = np.flatnonzero((rv_color == c) & observed)
        if members.size <= 1:
            continue
        vals = values[members]
        var = np.var(vals)
        if not ((np.sqrt(var) if use_sqrt else var) > epsilon):
            continue
        assign = kmeans_assign(vals, k, iteration)
        if assign is None:
            continue
        for idx in range(1, int(assign.max()) + 1):
            sel = members[assign == idx]
            if sel.size:
                rv_color[sel] = next_color
                next_color += 1
    return rv_color


class CompressedGraph:
    """Colour-passing compression of a ground ``Graph`` (``CompressedGraphWithObs.py:178-271``)."""

    def __init__(self, graph):
        self.g = graph
        self.clustered_evidence = set()
        self._rv_color = None
        self._f_color = None
        self._objects = None
        self._dev = None

    # ---- device plumbing ---------------------------------------------------------------------
    def _device(self):
        if self._dev is None:
            flat = flatten(self.g)
            dg = _abi.DeviceGraph(flat)
            sym = np.array([1 if getattr(f.potential, 'symmetric', False) else 0 for f in flat.factors], dtype=np.uint8)
            l = _abi.lib()
            ws_bytes = int(l.lhvi_color_workspace_bytes(dg.g))
            torch = _abi.require_gpu()
            self._dev = dict(flat=flat, dg=dg, sym=_abi.to_dev(sym) if sym.size else None,
                             ws=torch.empty(ws_bytes, dtype=torch.uint8, device=dg.device), ws_bytes=ws_bytes,
                             res=torch.zeros(4, dtype=torch.int32, device=dg.device))
        return self._dev

    def _upload_colors(self):
        d = self._device()
        if 'rvc' not in d:
            d['rvc'] = _abi.to_dev(self._rv_color)
            d['fc'] = _abi.to_dev(self._f_color)
            d['rvc2'] = d['rvc'].clone()
            d['fc2'] = d['fc'].clone()
        return d

    # ---- reference API -----------------------------------------------------------------------
    def init_cluster(self, is_split_cont_evidence=True):
        self.array_flat = False
        self._rv_color, self._f_color = initial_colors(self.g, is_split_cont_evidence)
        self.num_rv_clusters = int(self._rv_color.max()) + 1 if self._rv_color.size else 0
        self.num_factor_clusters = int(self._f_color.max()) + 1 if self._f_color.size else 0
        self._objects = None
        if self._dev is not None:
            for k in ('rvc', 'fc', 'rvc2', 'fc2'):
                self._dev.pop(k, None)
        self.clustered_evidence = set()
        self._coarse_evidence = not is_split_cont_evidence

    def split_evidence(self, k=2, iteration=10, epsilon=0, use_sqrt=True):
        """k-means split of evidence clusters by value (CGWO:236-247; ``use_sqrt=False`` gives HLBP:250-266)"""
        rv_color, f_color = self.colors()
        values = np.array([np.nan if rv.value is None else float(rv.value) for rv in ground_order(self.g.rvs)], dtype=np.float64)
        new = split_evidence_colors(values, rv_color, k, iteration, epsilon, use_sqrt)
        self.set_colors(new, f_color)

    def evidence_variances(self):
        """np.var of the member values of every evidence cluster (``SuperRV.get_variance``)"""
        rv_color, _ = self.colors()
        values = np.array([np.nan if rv.value is None else float(rv.value) for rv in ground_order(self.g.rvs)], dtype=np.float64)
        out = []
        for c in np.unique(rv_color[~np.isnan(values)]):
            out.append(float(np.var(values[(rv_color == c) & ~np.isnan(values)])))
        return out

    def _half_round(self, call, what):
        self.array_flat = False                  # (the partition moves: the loops that reach its fixed point set the flag again)
        """one refinement half round on the device: hash-table relabelling, repeated through the radix sort if the table
        overflowed (more than half a million distinct colours); returns the number of colours"""
        d = self._upload_colors()
        for method in (self.method, _abi.COLOR_SORT):
            call(d, method)
            n, collision, overflow = (int(x) for x in d['res'].cpu()[:3])
            if collision:
                raise _abi.LhviError('colour refinement fingerprint collision (%s side)' % what)
            if not overflow:
                return n
        raise _abi.LhviError('colour refinement: table overflow reported by the sort path')

    method = _abi.COLOR_HASH

    def split_factors(self):
        l = _abi.lib()
        n = self._half_round(lambda d, m: _abi.check(l.lhvi_color_refine_factors(
            d['dg'].g, _abi.ptr(d['sym']), _abi.ptr(d['rvc']), _abi.ptr(d['fc']), _abi.ptr(d['fc2']), _abi.ptr(d['res']),
            _abi.ptr(d['ws']), d['ws_bytes'], m, _abi.stream_ptr())), 'factor')
        d = self._dev
        d['fc'], d['fc2'] = d['fc2'], d['fc']
        self.num_factor_clusters = n
        self._objects = None

    def split_rvs(self):
        l = _abi.lib()
        n = self._half_round(lambda d, m: _abi.check(l.lhvi_color_refine_rvs(
            d['dg'].g, _abi.ptr(d['fc']), _abi.ptr(d['rvc']), _abi.ptr(d['rvc2']), _abi.ptr(d['res']), _abi.ptr(d['ws']),
            d['ws_bytes'], m, _abi.stream_ptr())), 'variable')
        d = self._dev
        d['rvc'], d['rvc2'] = d['rvc2'], d['rvc']
        self.num_rv_clusters = n
        self._objects = None

    def run(self):
        """``CompressedGraph.run`` (``CompressedGraphWithObs.py:264-271``): refine until #rv clusters is stable."""
        self.init_cluster(is_split_cont_evidence=True)
        prev = -1
        while prev != self.num_rv_clusters:
            prev = self.num_rv_clusters
            self.split_factors()
            self.split_rvs()
        self.array_flat = True                   # stable: flatten() may lift on arrays (lifted_flat)
        return self

    # ---- colours / objects -------------------------------------------------------------------
    def colors(self):
        """(rv_color, f_color) as host int32 arrays aligned with ``list(g.rvs)`` / ``list(g.factors)``"""
        if self._dev is not None and 'rvc' in self._dev:
            self._rv_color = self._dev['rvc'].cpu().numpy()
            self._f_color = self._dev['fc'].cpu().numpy()
        return self._rv_color, self._f_color

    def set_colors(self, rv_color, f_color):
        """Inject a partition (tests use this to exercise the host logic without a GPU)."""
        self.array_flat = False
        self._rv_color = np.asarray(rv_color, dtype=np.int32)
        self._f_color = np.asarray(f_color, dtype=np.int32)
        self.num_rv_clusters = int(self._rv_color.max()) + 1 if self._rv_color.size else 0
        self.num_factor_clusters = int(self._f_color.max()) + 1 if self._f_color.size else 0
        self._objects = None
        if self._dev is not None:
            for k in ('rvc', 'fc', 'rvc2', 'fc2'):
                self._dev.pop(k, None)

    array_flat = False          # set by the callers that ran the colour passing to its fixed point: flatten() may take lifted_flat()

    def lifted_flat(self, require_device_potentials=False):
        """The lifted ``FlatGraph`` from the ground arrays and the colours (``lift_flat``) with the cluster objects attached --
        what ``flatten`` builds from the ``SuperRV`` / ``SuperF`` objects, edge for edge (same incidences, canonical edges, rows
        and counts: a cluster's edges are its representative's either way; the potential and domain tables are the ground
        graph's), without walking 15 000 objects a second time.  None when there is no uploaded ground graph, or the partition is
        not stable (the caller then flattens the objects)."""
        if self._dev is None or self._rv_color is None:
            return None
        from .potentials import POT_GENERIC
        rvc, fc = self.colors()
        gflat = self._dev['flat']
        # (the ground arrays were made when the graph was first uploaded: evidence set on the rv objects since then is taken over)
        now = np.fromiter((np.nan if rv.value is None else rv.value for rv in gflat.rvs), dtype=np.float64, count=gflat.V)
        if not np.array_equal(now, gflat.var_value, equal_nan=True):
            gflat.var_value = now
        try:
            lf = lift_flat(gflat, np.asarray(rvc), np.asarray(fc))
        except _abi.LhviError:
            return None
        rvs, factors = self.rvs_list, self.factors_list
        if len(rvs) != lf.V or len(factors) != lf.F or (lf.V and (rvs[0].id != 0 or rvs[-1].id != lf.V - 1)) \
                or (lf.F and (factors[0].id != 0 or factors[-1].id != lf.F - 1)):
            return None
        if require_device_potentials:
            used = np.unique(lf.fac_pot)
            generic = used[lf.pot_kind[used] == POT_GENERIC]
            if generic.size:
                raise NotImplementedError('potential %r has no device encoding (device_spec); refusing to fall back to the CPU'
                                          % type(lf.potentials[int(generic[0])]).__name__)
        lf.rvs, lf.factors = rvs, factors
        lf.var_index = {c: i for i, c in enumerate(rvs)}
        lf.fac_index = {c: i for i, c in enumerate(factors)}
        return lf

    def _build(self):
        if self._objects is None:
            rv_color, f_color = self.colors()
            self._objects = build_lifted_objects(self.g, rv_color, f_color)
        return self._objects

    @property
    def rvs(self):
        return self._build()[0]

    @property
    def factors(self):
        return self._build()[1]

    @property
    def rvs_list(self):
        return sorted(self.rvs)

    @property
    def factors_list(self):
        return sorted(self.factors)


def build_lifted_objects(g, rv_color, f_color):
    """``SuperRV`` / ``SuperF`` sets for a partition; cluster representative = member with the smallest id."""
    from .flat import gc_paused
    with gc_paused():
        return _build_lifted_objects(g, rv_color, f_color)


def _build_lifted_objects(g, rv_color, f_color):
    rvs, factors = ground_order(g.rvs), ground_order(g.factors)
    groups = {}
    for rv, c in zip(rvs, rv_color.tolist()):
        groups.setdefault(c, []).append(rv)
    fgroups = {}
    for f, c in zip(factors, f_color.tolist()):
        fgroups.setdefault(c, []).append(f)
    super_rvs = [SuperRV(set(members), cid=c, order=members) for c, members in sorted(groups.items())]
    super_fs = [SuperF(set(members), cid=c, order=members) for c, members in sorted(fgroups.items())]
    for s in super_rvs:
        s.update_nb()
    for s in super_fs:
        s.update_nb()
    # keep a deterministic iteration order (the reference uses sets; solvers only rely on membership)
    return _OrderedSet(super_rvs), _OrderedSet(super_fs)


class _OrderedSet(list):
    """list with set-like membership helpers; iteration order = ascending cluster id"""

    def __contains__(self, x):
        return any(x is y for y in self)


def _lift_reduce_host(flat, rv_color, f_color):
    """the O(V + F) part of ``lift_flat`` in NumPy: representatives, cluster sizes, evidence means, and the factor colours
    along the representatives' adjacency rows"""
    rv_color = np.asarray(rv_color, dtype=np.int64)
    f_color = np.asarray(f_color, dtype=np.int64)
    nV, nF = int(rv_color.max()) + 1, int(f_color.max()) + 1
    rep_v = np.full(nV, flat.V, dtype=np.int64)
    np.minimum.at(rep_v, rv_color, np.arange(flat.V))
    rep_f = np.full(nF, flat.F, dtype=np.int64)
    np.minimum.at(rep_f, f_color, np.arange(flat.F))
    mult_v = np.bincount(rv_color, minlength=nV).astype(np.float64)
    mult_f = np.bincount(f_color, minlength=nF).astype(np.float64)
    # evidence value of a cluster: running sum of member values / size (SuperRV.get_value)
    val = flat.var_value[rep_v].copy()
    obs = ~np.isnan(val)
    if obs.any():
        sums = np.zeros(nV)
        vv = np.where(np.isnan(flat.var_value), 0.0, flat.var_value)
        np.add.at(sums, rv_color, vv)      # sequential accumulation in ground order
        val[obs] = sums[obs] / mult_v[obs]
    deg = (flat.var_ptr[rep_v + 1] - flat.var_ptr[rep_v]).astype(np.int64)
    start = np.zeros(nV + 1, dtype=np.int64)
    np.cumsum(deg, out=start[1:])
    slots = np.repeat(flat.var_ptr[rep_v].astype(np.int64) - start[:-1], deg) + np.arange(int(start[-1]), dtype=np.int64)
    fcol = f_color[flat.edge_fac[flat.var_edge[slots]]]
    owner = np.repeat(np.arange(nV, dtype=np.int64), deg)
    key = owner * nF + fcol                                   # (cluster, factor colour) pairs along the rows
    uniq, first, cnt = np.unique(key, return_index=True, return_counts=True)
    return dict(nV=nV, nF=nF, rep_v=rep_v, rep_f=rep_f, mult_v=mult_v, mult_f=mult_f, val=val, pairs=(uniq, first, cnt),
                color_of_edge_var=lambda g_edge: rv_color[flat.edge_var[g_edge]])


class TensorGraph:
    """the ground arrays the lifting reductions read, as torch tensors on any device (``DeviceGraph.t`` on the GPU; CPU tensors
    in the CPU tests of the host logic)"""

    def __init__(self, flat, device='cpu'):
        import torch
        self.flat = flat
        self.t = {name: torch.from_numpy(np.ascontiguousarray(getattr(flat, name))).to(device)
                  for name in ('fac_ptr', 'edge_var', 'edge_fac', 'var_ptr', 'var_edge', 'var_value')}
        self.device = self.t['fac_ptr'].device


def first_members(colors, n_colors, size):
    """first (smallest-index) member of every colour as an int64 tensor on the colours' device (``lhvi_color_first_members`` on
    the GPU; CPU tensors, which only the CPU tests of the host logic use, go through torch)"""
    import torch
    if colors.device.type != 'cuda':
        return torch.full((n_colors,), size, dtype=torch.int64).scatter_reduce_(0, colors.long(), torch.arange(size), 'amin')
    c32 = colors if colors.dtype == torch.int32 and colors.is_contiguous() else colors.to(torch.int32).contiguous()
    out = torch.empty(n_colors, dtype=torch.int32, device=colors.device)
    _abi.check(_abi.lib().lhvi_color_first_members(_abi.ptr(c32), int(size), int(n_colors), _abi.ptr(out), _abi.stream_ptr()))
    return out.long()


def segment_sums(values, lengths):
    """sums of consecutive runs of `values` (run s has lengths[s] entries), each a running sum in index order
    (``lhvi_color_segment_sums`` on the GPU: torch's segmented reduction adds in a tree there, so the last bit of a cluster's
    evidence value would depend on the device; CPU tensors: ``torch.segment_reduce``, which is sequential)"""
    import torch
    if values.device.type != 'cuda':
        return torch.segment_reduce(values, 'sum', lengths=lengths, unsafe=True)
    n = int(lengths.numel())
    offsets = torch.zeros(n + 1, dtype=torch.int64, device=values.device)
    torch.cumsum(lengths, 0, out=offsets[1:])
    out = torch.empty(n, dtype=torch.float64, device=values.device)
    vals = values.contiguous()
    _abi.check(_abi.lib().lhvi_color_segment_sums(_abi.ptr(vals), _abi.ptr(offsets), n, _abi.ptr(out), _abi.stream_ptr()))
    return out


def _lift_reduce_device(flat, dg, rvc, fc):
    """the same reductions on the device, for colour arrays that are already there (``refine_flat(device_out=True)``): only
    lifted-size arrays come back to the host.  Evidence sums run over each cluster's observed members in ground order (one
    thread per cluster segment), i.e. in the order of the host path."""
    import torch
    dev = rvc.device
    rl, fl = rvc.long(), fc.long()
    nV, nF = int(rl.max().item()) + 1, int(fl.max().item()) + 1
    rep_v, rep_f = first_members(rvc, nV, flat.V), first_members(fc, nF, flat.F)
    mult_v = torch.bincount(rl, minlength=nV).to(torch.float64)
    mult_f = torch.bincount(fl, minlength=nF).to(torch.float64)
    value = dg.t['var_value']
    val = value[rep_v].clone()
    obs_members = torch.nonzero(~torch.isnan(value)).flatten()
    if obs_members.numel():
        oc = rl[obs_members]
        order = torch.sort(oc, stable=True).indices                   # members grouped by cluster, ground order inside
        lengths = torch.bincount(oc, minlength=nV)
        sums = segment_sums(value[obs_members][order], lengths)
        ob = ~torch.isnan(val)
        val[ob] = sums[ob] / mult_v[ob]
    var_ptr = dg.t['var_ptr'].long()
    deg = var_ptr[rep_v + 1] - var_ptr[rep_v]
    start = torch.zeros(nV + 1, dtype=torch.int64, device=dev)
    torch.cumsum(deg, 0, out=start[1:])
    total = int(start[-1].item())
    slots = torch.repeat_interleave(var_ptr[rep_v] - start[:-1], deg, output_size=total) + torch.arange(total, device=dev)
    fcol = fl[dg.t['edge_fac'].long()[dg.t['var_edge'].long()[slots]]]
    owner = torch.repeat_interleave(torch.arange(nV, device=dev), deg, output_size=total)
    uniq, inv, cnt = torch.unique(owner * nF + fcol, return_inverse=True, return_counts=True)
    first = first_members(inv, int(uniq.numel()), total)
    edge_var_d = dg.t['edge_var']
    host = lambda t: t.cpu().numpy()
    return dict(nV=nV, nF=nF, rep_v=host(rep_v), rep_f=host(rep_f), mult_v=host(mult_v), mult_f=host(mult_f), val=host(val),
                pairs=(host(uniq), host(first), host(cnt)),
                color_of_edge_var=lambda g_edge: host(rl[edge_var_d[torch.from_numpy(np.asarray(g_edge, dtype=np.int64)).to(dev)].long()]))


SMALL_LIFT_EDGES = 1 << 15


def lift_flat(flat, rv_color, f_color, dg=None):
    """Lifted ``FlatGraph`` straight from a ground ``FlatGraph`` and a partition, without Python objects
    (vectorised; the 10M-edge path).  Representative of a cluster = its first ground member; lifted edges
    are the representative factor's incidences; ``count`` comes from the representative variable's
    incident factor colours (``SuperRV.update_nb``, ``CompressedGraphWithObs.py:41-45``).
    With device colour tensors (``refine_flat(..., device_out=True)``) and the graph's ``DeviceGraph`` the reductions over
    the ground graph run on the device and only lifted-size arrays reach the host."""
    from .flat import FlatGraph
    on_device = dg is not None and not isinstance(rv_color, np.ndarray) and hasattr(rv_color, 'device')
    if on_device and rv_color.device.type == 'cuda' and flat.E < SMALL_LIFT_EDGES:
        # a reference-size graph: the ~40 small launches and five synchronisations of the device reductions cost 2 ms, the same
        # reductions in NumPy 0.3 ms (same sums in the same order) -- two small copies instead
        rv_color, f_color, on_device = rv_color.cpu().numpy(), f_color.cpu().numpy(), False
    R = _lift_reduce_device(flat, dg, rv_color, f_color) if on_device else _lift_reduce_host(flat, rv_color, f_color)
    nV, nF, rep_v, rep_f = R['nV'], R['nF'], R['rep_v'], R['rep_f']
    arity = (flat.fac_ptr[1:] - flat.fac_ptr[:-1])[rep_f]
    fac_ptr = np.zeros(nF + 1, dtype=np.int32)
    np.cumsum(arity, out=fac_ptr[1:])
    E = int(fac_ptr[-1])
    edge_fac = np.repeat(np.arange(nF, dtype=np.int32), arity)
    edge_pos = (np.arange(E) - fac_ptr[edge_fac]).astype(np.int32)
    g_edge = flat.fac_ptr[rep_f][edge_fac] + edge_pos
    edge_var = R['color_of_edge_var'](g_edge).astype(np.int32)
    # canonical edge of each (factor, variable) pair
    pair = edge_fac.astype(np.int64) * nV + edge_var
    order = np.argsort(pair, kind='stable')
    first = np.ones(E, dtype=bool)
    first[1:] = pair[order][1:] != pair[order][:-1]
    # within a run the first element (stable sort) is the smallest edge id
    run_start = np.maximum.accumulate(np.where(first, np.arange(E), 0))
    edge_canon = np.empty(E, dtype=np.int32)
    edge_canon[order] = order[run_start]
    # variable side: the representative ground rv's incident factors, grouped by factor colour in first-seen order
    # (vectorised: one pass over the representatives' adjacency rows instead of a Python loop per cluster)
    uniq, first, cnt = R['pairs']                             # distinct (cluster, factor colour) pairs along the rows
    order = np.argsort(first, kind='stable')                  # first-seen order (rows are contiguous per cluster)
    uniq, cnt = uniq[order], cnt[order]
    pc, pf = uniq // nF, uniq % nF
    canon_e = np.flatnonzero(edge_canon == np.arange(E))
    ckey = edge_fac[canon_e].astype(np.int64) * nV + edge_var[canon_e]
    csort = np.argsort(ckey, kind='stable')
    pos = np.searchsorted(ckey[csort], pf * nV + pc)
    if pos.size and ((pos >= ckey.size).any() or (ckey[csort][np.minimum(pos, ckey.size - 1)] != pf * nV + pc).any()):
        raise _abi.LhviError('lift_flat: a (cluster, factor colour) pair of a representative has no lifted edge -- the partition is not stable')
    var_edge = canon_e[csort][pos].astype(np.int32)
    counts = np.ones(E, dtype=np.float64)
    counts[var_edge] = cnt
    var_ptr = np.zeros(nV + 1, dtype=np.int32)
    np.cumsum(np.bincount(pc, minlength=nV), out=var_ptr[1:])
    edge_count = counts[edge_canon]
    return FlatGraph(
        V=nV, F=nF, E=E, fac_ptr=fac_ptr, edge_var=edge_var, edge_fac=edge_fac, edge_pos=edge_pos,
        edge_canon=edge_canon, var_ptr=var_ptr, var_edge=var_edge.reshape(-1),
        edge_count=edge_count, lifted=True, fac_pot=flat.fac_pot[rep_f].astype(np.int32),
        pot_kind=flat.pot_kind, pot_off=flat.pot_off, pot_param=flat.pot_param,
        var_value=R['val'], var_dom=flat.var_dom[rep_v].astype(np.int32), var_mult=R['mult_v'], fac_mult=R['mult_f'],
        dom_cont=flat.dom_cont, dom_lo=flat.dom_lo, dom_hi=flat.dom_hi, dom_ptr=flat.dom_ptr, dom_val=flat.dom_val,
        potentials=flat.potentials, domains=flat.domains)


def refine_flat(flat, symmetric, rv_color, f_color, max_rounds=1000, dg=None, stats=None, method=None, device_out=False):
    """Colour passing on flat arrays entirely on the device (the large-graph path: no Python objects).
    Same loop as ``CompressedGraph.run``: factor half-round, rv half-round, until #rv colours is stable; one host
    synchronisation per round (both halves' result words are read together).
    ``dg``: an already uploaded ``DeviceGraph`` of ``flat``; ``stats``: dict that receives the number of rounds;
    ``method``: ``_abi.COLOR_HASH`` (default; a half round whose table overflows is repeated by sorting) or ``_abi.COLOR_SORT``;
    ``device_out``: return the colour arrays as device tensors instead of NumPy arrays."""
    torch = _abi.require_gpu()
    dg = dg or _abi.DeviceGraph(flat)
    l = _abi.lib()
    method = _abi.COLOR_HASH if method is None else method
    ws_bytes = int(l.lhvi_color_workspace_bytes(dg.g))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dg.device)
    res = torch.zeros(8, dtype=torch.int32, device=dg.device)          # [0:4] factor half, [4:8] variable half
    res_f, res_v = res[:4], res[4:]
    sym = symmetric.to(torch.uint8).contiguous() if torch.is_tensor(symmetric) else _abi.to_dev(np.asarray(symmetric, dtype=np.uint8))
    # (device tensors are copied: the loop ping-pongs between two buffers per side and must not write into the caller's)
    rvc = rv_color.to(torch.int32).clone() if torch.is_tensor(rv_color) else _abi.to_dev(np.asarray(rv_color, dtype=np.int32))
    fc = f_color.to(torch.int32).clone() if torch.is_tensor(f_color) else _abi.to_dev(np.asarray(f_color, dtype=np.int32))
    rvc2, fc2 = rvc.clone(), fc.clone()
    n_rv = int(rvc.max().item()) + 1 if flat.V else 0
    prev = -1
    rounds = sorted_halves = 0
    st = _abi.stream_ptr()

    def factors(m):
        _abi.check(l.lhvi_color_refine_factors(dg.g, _abi.ptr(sym), _abi.ptr(rvc), _abi.ptr(fc), _abi.ptr(fc2),
                                               _abi.ptr(res_f), _abi.ptr(ws), ws_bytes, m, st))

    def rvs(m):
        _abi.check(l.lhvi_color_refine_rvs(dg.g, _abi.ptr(fc2), _abi.ptr(rvc), _abi.ptr(rvc2), _abi.ptr(res_v),
                                           _abi.ptr(ws), ws_bytes, m, st))
    while prev != n_rv and rounds < max_rounds:
        prev = n_rv
        factors(method)
        rvs(method)
        r = [int(x) for x in res.cpu()]
        if r[2] or r[6]:                               # a table overflowed: the round again through the sort
            sorted_halves += 2
            factors(_abi.COLOR_SORT)
            rvs(_abi.COLOR_SORT)
            r = [int(x) for x in res.cpu()]
        if r[1] or r[5]:
            raise _abi.LhviError('colour refinement fingerprint collision')
        fc, fc2 = fc2, fc
        rvc, rvc2 = rvc2, rvc
        n_rv = r[4]
        rounds += 1
    if stats is not None:
        stats['rounds'] = rounds
        stats['sorted_half_rounds'] = sorted_halves
    if device_out:
        return rvc, fc
    return rvc.cpu().numpy(), fc.cpu().numpy()
