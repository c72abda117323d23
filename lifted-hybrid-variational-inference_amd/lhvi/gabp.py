"""Gaussian belief propagation on the GPU: ``GaBP`` (ground) and ``GaLBP`` (lifted).

Same constructor / method surface as the reference (``GaBP.py:7-216``, ``GaLBP.py:8-217``):
``GaBP(g).run(iteration)``, ``.belief(x, rv)``, ``.get_belief_params(rv)``, ``.map(rv)``, attribute
``.message``.  ``run`` flattens the object graph once, keeps both message buffers resident in HBM and
issues the flooding sweeps through ``lhvi_gabp_run`` (hand-written HIP, ``csrc/gabp.hip``).
"""
from __future__ import annotations

import os

import numpy as np

from . import _abi
from .flat import flatten


def _norm_pdf_var(x, mu, sig):
    # GaBP.norm_pdf (GaBP.py:14-18): note the normaliser uses the variance, not its square root
    # (`sig` is the reference's parameter name; in GaBP / GaLBP it holds a variance)
    u = x - mu
    return np.exp(-u * u * 0.5 / sig) / (2.506628274631 * sig)


def pull_plan(flat):
    """host arrays of ``lhvi_gabp_plan_t`` (include/lhvi.h): per variable-CSR slot the slot of the partner argument's
    edge (observed partner: -1 - its variable), 4 * potential index + position code, and on a lifted graph the slot's count"""
    ve = flat.var_edge.astype(np.int64)
    nnz = ve.size
    edge_slot = np.full(max(flat.E, 1), -1, dtype=np.int64)
    edge_slot[ve] = np.arange(nnz)
    f = flat.edge_fac[ve].astype(np.int64)
    base = flat.fac_ptr[f].astype(np.int64)
    arity = flat.fac_ptr[f + 1] - base
    pos = ve - base
    code = np.where(arity == 1, 0, np.where(arity == 2, 1 + pos, 3))
    pe = np.where(arity == 2, base + (1 - np.minimum(pos, 1)), ve)       # partner edge of a pairwise factor
    pce = flat.edge_canon[pe].astype(np.int64)
    pvar = flat.edge_var[pe].astype(np.int64)
    hidden_partner = (arity == 2) & np.isnan(flat.var_value[pvar])
    if (hidden_partner & (edge_slot[pce] < 0)).any():
        raise _abi.LhviError('a hidden partner argument has no variable-side slot')
    pslot = np.where(hidden_partner, edge_slot[pce], -1 - pvar)
    info = flat.fac_pot[f].astype(np.int64) * 4 + code
    # the 16-byte slot records of lhvi_gabp_plan_t.rec
    deg = np.diff(flat.var_ptr).astype(np.int64)
    svar = np.repeat(np.arange(flat.V, dtype=np.int64), deg)
    within = np.arange(nnz, dtype=np.int64) - flat.var_ptr[svar]
    long_row = deg[svar] > 512
    rec = np.zeros((nnz, 4), dtype=np.int32)
    rec[:, 0], rec[:, 1] = pslot, info
    rec[:, 2] = np.where(long_row, 0, within | (deg[svar] << 10)) | (np.isnan(flat.var_value[svar]).astype(np.int64) << 20) | (long_row.astype(np.int64) << 21)
    # twelve words per potential: par[0 .. 10] (zero padded) and the kind (lhvi_gabp_plan_t.pot_words)
    P = int(flat.pot_kind.size)
    words = np.zeros((max(P, 1), 12))
    npar = np.diff(flat.pot_off)
    for j in range(11):
        has = npar > j
        words[:P][has, j] = flat.pot_param[flat.pot_off[:-1][has] + j]
    words[:P, 11] = flat.pot_kind
    # segments of the slot order (lhvi_gabp_plan_t.seg): rows of at most 512 entries grouped by the 256-slot window their first
    # slot lies in, runs broken at every longer row
    rows = np.flatnonzero(deg > 0)
    start = flat.var_ptr[rows].astype(np.int64)
    is_hub = deg[rows] > 512
    # (a small graph is latency bound: narrower windows give it more, shorter workgroups)
    # (at most 256: the kernel stages a whole segment -- window - 1 + 512 slots at most -- in 768 LDS slots)
    window = min(max(int(os.environ.get('LHVI_GABP_WINDOW', 256 if nnz >= (1 << 17) else 64)), 1), 256)
    group = np.cumsum(is_hub) * (nnz // window + 2) + start // window
    keep = ~is_hub
    r_start, r_end, r_group = start[keep], start[keep] + deg[rows][keep], group[keep]
    if r_start.size:
        first_of = np.flatnonzero(np.concatenate([[True], r_group[1:] != r_group[:-1]]))
        last_of = np.concatenate([first_of[1:] - 1, [r_start.size - 1]])
        seg = np.stack([r_start[first_of], r_end[last_of]], axis=1).astype(np.int32)
    else:
        seg = np.zeros((0, 2), dtype=np.int32)
    return dict(pslot=pslot.astype(np.int32), info=info.astype(np.int32), rec=rec, pot_words=words, seg=seg,
                count=np.ascontiguousarray(flat.edge_count[ve], dtype=np.float64) if flat.lifted else None)


class _GaussianSweep:
    """State shared by the ground and the lifted solver: flat graph, device buffers, result cache."""

    verbose = False
    pull = True       # one launch per sweep, messages in slot order (lhvi_gabp_run_pull); False: the v2f / f2v kernel pair
    slot_records = True   # the pull kernel reads one 16-byte record per slot (lhvi_gabp_plan_t.rec) instead of walking the graph arrays

    def _check_degrees(self, flat):
        # the reference raises ZeroDivisionError (0 ** -1) when a hidden variable has no other incoming
        # message to multiply (GaBP.py:31) -- keep the error instead of producing NaN
        hidden = flat.var_hidden
        incoming = np.zeros(flat.V)
        np.add.at(incoming, flat.edge_var[flat.var_edge], flat.edge_count[flat.var_edge])
        if np.any(hidden & (incoming <= 1) & (np.diff(flat.var_ptr) > 0)):
            raise ZeroDivisionError('0.0 cannot be raised to a negative power')

    graph_replay_slots = 1 << 20   # runs on graphs of up to this many variable-side slots are recorded as a hipGraph and replayed

    @staticmethod
    def _same_graph(a, b):
        """same structure and potentials (evidence may differ: it lives in one device buffer that is refreshed)"""
        if (a.V, a.F, a.E, a.lifted) != (b.V, b.F, b.E, b.lifted) or a.var_edge.size != b.var_edge.size:
            return False
        return all(np.array_equal(getattr(a, k), getattr(b, k)) for k in
                   ('fac_ptr', 'edge_var', 'var_edge', 'fac_pot', 'pot_kind', 'pot_off', 'pot_param', 'edge_count', 'var_dom',
                    'dom_ptr', 'dom_val')) and np.array_equal(np.isnan(a.var_value), np.isnan(b.var_value))

    def _device_state(self, flat):
        """graph, pull plan, message buffers and workspace on the device, built once per graph and kept on the solver: a
        second ``run()`` on the same graph (same structure, potentials and hidden / observed pattern) only refreshes the
        evidence values"""
        st = getattr(self, '_state', None)
        if st is not None and self._same_graph(st['flat'], flat):
            if not np.array_equal(st['flat'].var_value, flat.var_value, equal_nan=True):
                st['dg'].t['var_value'].copy_(_abi.to_dev(flat.var_value))
                if st['dg'].t['edge_value'] is not None:
                    st['dg'].t['edge_value'].copy_(_abi.to_dev(np.ascontiguousarray(flat.var_value[flat.edge_var])))
            st['flat'] = flat
            st['dg'].flat = flat
            return st
        if st is not None:
            for h in st['graphs'].values():
                _abi.lib().lhvi_gabp_graph_destroy(h)
        torch = _abi.require_gpu()
        l = _abi.lib()
        dg = _abi.DeviceGraph(flat)
        st = dict(flat=flat, dg=dg, f2v=dg.empty(flat.E, 2), v2f=dg.empty(flat.E, 2), mv=dg.empty(flat.V, 2), plan=None, graphs={})
        if self.pull and flat.var_edge.size:
            host = pull_plan(flat)
            st['plan_dev'] = {k: (_abi.to_dev(a) if a is not None else None) for k, a in host.items()}
            plan = _abi.GabpPlanStruct()
            plan.pslot, plan.info, plan.count = (_abi.ptr(st['plan_dev'][k]) for k in ('pslot', 'info', 'count'))
            plan.rec = _abi.ptr(st['plan_dev']['rec']) if self.slot_records else None
            plan.pot_words = _abi.ptr(st['plan_dev']['pot_words'])
            plan.seg, plan.n_seg = _abi.ptr(st['plan_dev']['seg']), int(host['seg'].shape[0])
            plan.n_hub_rows = int((np.diff(flat.var_ptr) > 512).sum())
            st['plan'] = plan
            st['ws_bytes'] = int(l.lhvi_gabp_pull_workspace_bytes(dg.g))
            st['ws'] = torch.empty(st['ws_bytes'], dtype=torch.uint8, device=dg.device)
        self._state = st
        return st

    def _sweep(self, graph_like, iteration):
        import ctypes
        flat = flatten(graph_like)
        self._check_degrees(flat)
        st = self._device_state(flat)
        dg, f2v, v2f, mv = st['dg'], st['f2v'], st['v2f'], st['mv']
        l = _abi.lib()
        s = _abi.stream_ptr()
        if st['plan'] is not None:
            if flat.var_edge.size <= self.graph_replay_slots:
                # launch-bound sizes: the whole run (sweeps, conversion to edge order, marginals) is one graph launch
                h = st['graphs'].get(int(iteration))
                if h is None:
                    out = ctypes.c_void_p()
                    _abi.check(l.lhvi_gabp_graph_create(dg.g, dg.p, st['plan'], _abi.ptr(f2v), _abi.ptr(v2f), _abi.ptr(mv), int(iteration),
                                                        _abi.ptr(st['ws']), st['ws_bytes'], ctypes.byref(out)))
                    h = st['graphs'][int(iteration)] = out
                    if len(st['graphs']) > 8:             # (a caller sweeping over iteration counts: keep the cache small)
                        old = next(iter(st['graphs']))
                        l.lhvi_gabp_graph_destroy(st['graphs'].pop(old))
                _abi.check(l.lhvi_gabp_graph_launch(h, s))
            else:
                _abi.check(l.lhvi_gabp_run_pull(dg.g, dg.p, st['plan'], _abi.ptr(f2v), _abi.ptr(v2f), int(iteration), _abi.ptr(st['ws']),
                                                st['ws_bytes'], s))
                _abi.check(l.lhvi_gabp_marginals(dg.g, _abi.ptr(f2v), _abi.ptr(mv), s))
        else:
            _abi.check(l.lhvi_gabp_run(dg.g, dg.p, _abi.ptr(f2v), _abi.ptr(v2f), int(iteration), s))
            _abi.check(l.lhvi_gabp_marginals(dg.g, _abi.ptr(f2v), _abi.ptr(mv), s))
        self.flat, self.dg = flat, dg
        self.f2v_dev, self.v2f_dev, self.mu_var_dev = f2v, v2f, mv
        self._f2v = f2v.cpu().numpy()
        self._v2f = v2f.cpu().numpy()
        self._mu_var = mv.cpu().numpy()
        self._message = None

    def __del__(self):
        st = getattr(self, '_state', None)
        if st:
            try:
                for h in st['graphs'].values():
                    _abi.lib().lhvi_gabp_graph_destroy(h)
            except Exception:
                pass

    @property
    def message(self):
        """dict keyed (f, rv) / (rv, f) -> [mu, var] like the reference; built lazily from the device arrays"""
        if self._message is None:
            self._message = _message_dict(self.flat, self._f2v, self._v2f)
        return self._message

    @message.setter
    def message(self, value):
        self._message = value

    def _params(self, rv_flat):
        # an int is a variable index of a flat graph (RelationalGraph.ground_flat / build_flat have no rv objects)
        i = int(rv_flat) if isinstance(rv_flat, (int, np.integer)) else self.flat.var_index[rv_flat]
        return float(self._mu_var[i, 0]), float(self._mu_var[i, 1])

    @property
    def mu_var(self):
        """(V, 2) array of the marginals' (mean, variance), variable-index order (extension: batched get_belief_params)"""
        return self._mu_var


def _message_dict(flat, f2v, v2f):
    out = {}
    none = lambda m: [float(m[0]), None if np.isnan(m[1]) else float(m[1])]
    for k in range(flat.var_edge.size):
        e = int(flat.var_edge[k])
        rv, f = flat.rvs[flat.edge_var[e]], flat.factors[flat.edge_fac[e]]
        if rv.value is None:
            out[(f, rv)] = none(f2v[e])
            out[(rv, f)] = none(v2f[e])
        else:
            out[(f, rv)] = [0, 1]       # never updated by the reference (GaBP.py:163-165)
            out[(rv, f)] = None
    return out


class GaBP(_GaussianSweep):
    """Ground Gaussian BP (``GaBP.py``)."""

    def __init__(self, g=None):
        self.g = g
        self._message = {}

    def run(self, iteration=10, log_enable=False):
        self._sweep(self.g, iteration)

    def get_belief_params(self, rv):
        assert rv.value is None
        return self._params(rv)

    def belief(self, x, rv):
        if rv.value is not None:
            return 1 if x == rv.value else 0
        mu, var = self._params(rv)
        return _norm_pdf_var(x, mu, var)

    def map(self, rv):
        return rv.value if rv.value is not None else self._params(rv)[0]

    norm_pdf = staticmethod(_norm_pdf_var)


class GaLBP(_GaussianSweep):
    """Lifted Gaussian BP (``GaLBP.py``): colour passing, then the same sweep with ``count`` exponents."""

    def __init__(self, g=None):
        from .lifting import CompressedGraph
        self.g = CompressedGraph(g)
        self._message = {}

    def run(self, iteration=10, log_enable=False):
        # colour passing exactly as GaLBP.run does it (GaLBP.py:146-151)
        self.g.init_cluster()
        prev = -1
        while self.g.num_rv_clusters != prev:           # (= len(self.g.rvs), without building the cluster objects every round)
            prev = self.g.num_rv_clusters
            self.g.split_factors()
            self.g.split_rvs()
        self.g.array_flat = True                        # (a stable partition: the lifted arrays come from the ground arrays, lifting.lifted_flat)
        self._sweep(self.g, iteration)

    def belief(self, x, ground_rv):
        rv = ground_rv.cluster
        if rv.value is not None:
            return 1 if x == rv.value else 0
        mu, var = self._params(rv)
        return _norm_pdf_var(x, mu, var)

    def map(self, ground_rv):
        rv = ground_rv.cluster
        return rv.value if rv.value is not None else self._params(rv)[0]

    norm_pdf = staticmethod(_norm_pdf_var)
