"""Mixture-of-products variational inference on the GPU: ``VarInference`` and ``LiftedVarInference``.

Same surface as the reference (``VarInference.py:10-456``, ``LiftedVarInference.py:11-380``):
``VarInference(g, num_mixtures, num_quadrature_points).run(iteration, lr, is_log, log_fe)``, ``.free_energy()``,
``.belief(x, rv)``, ``.map(rv)``, ``.rvs_map(rvs)``, ``.GD_update(iteration, lr)``, attributes ``.w .eta .time_log``.
The expectation step (``gradient_w_tau`` / ``gradient_mu_var`` / ``gradient_category_tau`` / ``free_energy``) and the
ADAM update run as HIP kernels (``csrc/vi.hip``) on flat parameter arrays resident in HBM:
``w_tau [K]``, ``eta_c [V, K, 2]`` (mu, var), ``eta_tau [V, K, Dmax]``.

``reference_quirks=True`` (default) reproduces ``VarInference.py:147-150`` so that results match
the reference; ``False`` integrates the neighbours of a discrete variable with their own quadrature rules.

Initial parameters follow the reference's NumPy stream (``init_param``, ``VI:197-213``: ``np.random.rand`` while
iterating ``g.rvs``), so a seeded run on an ordered graph starts from the reference's parameters.
"""
from __future__ import annotations

import time
from math import sqrt, pi

import numpy as np
from numpy.polynomial.hermite import hermgauss

from . import _abi
from .flat import flatten


# (factor, k) items from which the thread-per-item kernel beats the 8-lane groups (measured: 6.8 k items 0.19 vs 0.085 ms per update,
# 1.9 M items 0.87 vs 2.43 ms; the single threads' floor is ~0.15 ms, their slope 0.45 vs 1.25 ns per item)
TINY_MIN_ITEMS = 1 << 17


def factor_lists(flat, K, T, obs_var=None, tiny_kernel=True):
    """The split of the factors among the kernels of the expectation step (``lhvi_vi_t.fac_list``, include/lhvi.h): a
    permutation of the factor ids in six segments and their lengths (n_cc, n_tiny, n_grp3, n_grp6, n_rest3, n_rest6).
    Inside the segments factors are ordered by (potential row, pattern of hidden arguments), so the lanes of a
    wavefront interpret the same formula on the same kind of grid."""
    from .potentials import POT_GAUSSIAN, POT_LINEAR_GAUSSIAN, POT_QUADRATIC, POT_XY
    F = flat.F
    if F == 0:
        return np.zeros(0, dtype=np.int32), (0, 0, 0, 0, 0, 0), np.zeros((0, 4), dtype=np.int32)
    arity = np.diff(flat.fac_ptr).astype(np.int64)
    ev = flat.edge_var
    hid, cont, nst = flat.var_hidden[ev], flat.var_cont[ev], flat.var_nstates[ev]
    gobs = np.zeros(ev.size, dtype=bool) if obs_var is None else (~hid & (np.asarray(obs_var)[ev] > 0))
    axis_len = np.where(hid, np.where(cont, T, nst), np.where(gobs, T, 1)).astype(np.int64)
    first = flat.fac_ptr[:-1].astype(np.int64)
    nonempty = arity > 0
    S = np.zeros(F, dtype=np.int64)
    S[nonempty] = np.add.reduceat(axis_len, first[nonempty])[:int(nonempty.sum())] if ev.size else 0
    # reduceat over consecutive non-empty segments: recompute exactly with a cumulative sum (robust to empty factors)
    csum = np.concatenate([[0], np.cumsum(axis_len)])
    S = csum[flat.fac_ptr[1:]] - csum[flat.fac_ptr[:-1]]
    kind = flat.pot_kind[flat.fac_pot]
    pair = arity == 2
    e0 = np.minimum(first, max(ev.size - 1, 0))
    e1 = np.minimum(first + 1, max(ev.size - 1, 0))
    ok0 = ~hid[e0] | cont[e0]
    ok1 = ~hid[e1] | cont[e1]
    cc = pair & np.isin(kind, (POT_GAUSSIAN, POT_QUADRATIC, POT_LINEAR_GAUSSIAN, POT_XY)) & (ev[e0] != ev[e1]) & ok0 & ok1
    small = arity <= 3
    # grid nodes of a factor (product of its axis lengths, capped: only "<= VI_TINY_NODES" matters)
    lg = np.concatenate([[0.0], np.cumsum(np.log2(axis_len.astype(np.float64)))])
    nodes = lg[flat.fac_ptr[1:]] - lg[flat.fac_ptr[:-1]]
    tiny = ~cc & small & (arity >= 1) & (nodes <= np.log2(_abi.VI_TINY_NODES) + 1e-9) & (K <= _abi.VI_TINY_K) & bool(tiny_kernel)
    if _abi.device_potentials(flat)[1].size > _abi.VI_TINY_PAR:
        tiny[:] = False                   # (the kernel keeps every parameter row of the device table in LDS)
    if tiny_kernel != 'always' and int(tiny.sum()) * K < TINY_MIN_ITEMS:
        tiny[:] = False                   # too few items to fill the device with single threads: the 8-lane groups finish sooner
    grp = ~cc & ~tiny & (arity >= 1) & (S <= _abi.VI_GROUP_SLOTS) & (K * S <= _abi.VI_GROUP_COMP)
    seg = np.where(cc, 0, np.where(tiny, 1, np.where(grp & small, 2, np.where(grp, 3, np.where(small, 4, 5)))))
    # hidden pattern of a factor: bit a set when argument a is hidden (arity <= 6)
    pos = np.arange(ev.size, dtype=np.int64) - np.repeat(first, arity)
    bits = np.zeros(F, dtype=np.int64)
    np.add.at(bits, np.repeat(np.arange(F), arity), hid.astype(np.int64) << np.minimum(pos, 30))
    order = np.lexsort((np.arange(F), bits, flat.fac_pot, seg))
    counts = np.bincount(seg, minlength=6)
    # per-edge axis records (lhvi_vi_t.edge_axis)
    rec = np.zeros((ev.size, 4), dtype=np.int32)
    rec[:, 0] = ev
    rec[:, 1] = axis_len | (hid.astype(np.int64) << 16) | (cont.astype(np.int64) << 17) | (gobs.astype(np.int64) << 18)
    rec[:, 3] = flat.dom_ptr[flat.var_dom[ev]]
    obs_d = np.flatnonzero(~hid & ~cont)
    if obs_d.size:                                  # state index of an observed discrete value (vi_state_index of csrc/vi.hip)
        v = ev[obs_d]
        dom = flat.var_dom[v]
        lo, n = flat.dom_ptr[dom].astype(np.int64), nst[obs_d].astype(np.int64)
        val = flat.var_value[v]
        tt = np.arange(int(n.max()))[None, :]
        states = flat.dom_val[np.minimum(lo[:, None] + tt, flat.dom_val.size - 1)]
        match = (tt < n[:, None]) & (states == val[:, None])
        fix = np.where(match.any(axis=1), match.argmax(axis=1), val.astype(np.int64))      # first matching state, else (int) x
        rec[obs_d, 2] = fix
    return order.astype(np.int32), tuple(int(c) for c in counts), rec


class _Variational:
    var_threshold = 0.1
    reference_quirks = True
    verbose = False
    fused_loop = True           # ADAM_update enqueues its whole loop through lhvi_vi_adam_run (else: one call per array and step)
    tiny_kernel = True          # factors with a handful of grid nodes take the thread-per-(factor, k) kernel when there are >= 2^17
                                # such items ('always': whatever their number; False: the group kernel)
    factor_lists = True         # the factors are split among the expectation kernels on the host (else: every kernel classifies them itself)

    def _init_common(self, num_mixtures, num_quadrature_points):
        self.K = num_mixtures
        self.T = num_quadrature_points
        self.quad_x, self.quad_w = hermgauss(self.T)
        self.quad_w = self.quad_w / sqrt(pi)
        self.time_log = []
        self._dev = None
        self._cache = {}

    # ---- device state -------------------------------------------------------------------------
    def _setup(self, graph_like):
        self._setup_flat(flatten(graph_like, require_device_potentials=True))

    def _setup_flat(self, flat):
        """device state for a ready-made FlatGraph (large graphs built without Python objects)"""
        self.flat, self.dg = flat, _abi.DeviceGraph(flat)
        torch = _abi.require_gpu()
        disc = flat.var_hidden & ~flat.var_cont
        self._cont, self._disc = flat.var_hidden & flat.var_cont, disc
        self.Dmax = int(flat.var_nstates[disc].max()) if disc.any() else 1
        dg, K = self.dg, self.K
        d = dict(gh_x=_abi.to_dev(self.quad_x), gh_w=_abi.to_dev(self.quad_w),
                 w_tau=dg.zeros(K), w=dg.zeros(K), eta_c=dg.zeros(flat.V, K, 2), tau_d=dg.zeros(flat.V, K, self.Dmax),
                 eta_d=dg.zeros(flat.V, K, self.Dmax), g_w=dg.zeros(K), g_c=dg.zeros(flat.V, K, 2),
                 g_d=dg.zeros(flat.V, K, self.Dmax), fe=dg.zeros(1))
        for name in ('w_tau', 'eta_c', 'tau_d'):
            d['m_' + name] = torch.zeros_like(d[name])
            d['s_' + name] = torch.zeros_like(d[name])
        # rv.N per variable (LVI:64-67): the sum of the row's counts, in row order like the kernel's own loop
        if flat.lifted and flat.var_edge.size:
            n_row = np.zeros(flat.V)
            np.add.at(n_row, np.repeat(np.arange(flat.V), np.diff(flat.var_ptr)), flat.edge_count[flat.var_edge])
            d['var_N'] = _abi.to_dev(n_row)
        self._dev = d
        self._fac_counts = None
        if self.factor_lists:
            order, self._fac_counts, rec = factor_lists(flat, K, self.T, getattr(self, '_obs_var_host', None), self.tiny_kernel)
            d['fac_list'] = _abi.to_dev(order if order.size else np.zeros(1, dtype=np.int32))
            d['edge_axis'] = _abi.to_dev(rec if rec.size else np.zeros((1, 4), dtype=np.int32))
        ws_bytes = int(_abi.lib().lhvi_vi_workspace_bytes(dg.g, self._struct()))
        d['ws'] = torch.empty(ws_bytes, dtype=torch.uint8, device=dg.device)
        d['ws_bytes'] = ws_bytes
        # masks so that rows of observed / other-type variables never move
        d['mask_c'] = _abi.to_dev(np.repeat(self._cont[:, None, None], K, 1).repeat(2, 2).astype(np.float64))
        nst = np.where(disc, flat.var_nstates, 0)
        md = (np.arange(self.Dmax)[None, :] < nst[:, None]).astype(np.float64)
        d['mask_d'] = _abi.to_dev(np.repeat(md[:, None, :], K, 1))
        # one softmax launch serves every discrete row when they all have the same number of states
        kinds = set(nst[disc].tolist())
        self._uniform_states = int(next(iter(kinds))) if len(kinds) == 1 else 0
        self._has_disc = bool(disc.any())

    def _struct(self):
        d = self._dev
        p = _abi.ViStruct()
        p.K, p.T, p.Dmax, p.quirks = self.K, self.T, self.Dmax, 1 if self.reference_quirks else 0
        p.gh_x, p.gh_w, p.w = _abi.ptr(d['gh_x']), _abi.ptr(d['gh_w']), _abi.ptr(d['w'])
        p.eta_c, p.eta_d = _abi.ptr(d['eta_c']), _abi.ptr(d['eta_d'])
        p.var_N = _abi.ptr(d.get('var_N'))
        if getattr(self, '_fac_counts', None) is not None:
            p.fac_list, p.edge_axis = _abi.ptr(d['fac_list']), _abi.ptr(d['edge_axis'])
            p.n_cc, p.n_tiny, p.n_grp3, p.n_grp6, p.n_rest3, p.n_rest6 = self._fac_counts
            p.tiny_par_words = int(self.dg.pot_param_words)
        return p

    def _opt_struct(self):
        d = self._dev
        o = _abi.ViOptStruct()
        for field, name in (('w_tau', 'w_tau'), ('w', 'w'), ('eta_c', 'eta_c'), ('tau_d', 'tau_d'), ('eta_d', 'eta_d'),
                            ('m_w', 'm_w_tau'), ('s_w', 's_w_tau'), ('m_c', 'm_eta_c'), ('s_c', 's_eta_c'), ('m_d', 'm_tau_d'),
                            ('s_d', 's_tau_d'), ('g_w', 'g_w'), ('g_c', 'g_c'), ('g_d', 'g_d'), ('fe', 'fe')):
            setattr(o, field, _abi.ptr(d[name]))
        o.lr, o.b1, o.b2, o.eps, o.var_min, o.t = float(self.alpha), self.b1, self.b2, self.eps, float(self.var_threshold), int(self.t)
        return o

    def _refresh(self):
        """w = softmax(w_tau); eta[drv] = softmax(eta_tau[drv], 1) (VI:211-213)"""
        d, l, st = self._dev, _abi.lib(), _abi.stream_ptr()
        _abi.check(l.lhvi_softmax_rows(_abi.ptr(d['w_tau']), _abi.ptr(d['w']), 1, self.K, self.K, st))
        flat = self.flat
        if self._has_disc:
            if self._uniform_states:
                D = self._uniform_states
                _abi.check(l.lhvi_softmax_rows(_abi.ptr(d['tau_d']), _abi.ptr(d['eta_d']), flat.V * self.K, D, self.Dmax, st))
            else:
                torch = _abi.require_gpu()
                e = torch.exp(d['tau_d']) * d['mask_d']
                d['eta_d'].copy_(e / e.sum(dim=2, keepdim=True).clamp_min(1e-300))
            d['eta_d'].mul_(d['mask_d'])
        self._cache = {}

    def _upload_params(self, w_tau, eta_c, tau_d):
        d = self._dev
        d['w_tau'].copy_(_abi.to_dev(np.asarray(w_tau, dtype=np.float64)))
        d['eta_c'].copy_(_abi.to_dev(np.nan_to_num(np.asarray(eta_c, dtype=np.float64), nan=1.0)))
        td = np.zeros((self.flat.V, self.K, self.Dmax))
        src = np.nan_to_num(np.asarray(tau_d, dtype=np.float64), nan=0.0)
        td[:, :, :min(self.Dmax, src.shape[2])] = src[:, :, :self.Dmax]
        d['tau_d'].copy_(_abi.to_dev(td))
        self._refresh()

    def init_param(self):
        """VI:197-213 -- same RNG stream as the reference (np.random.rand in g.rvs order)"""
        if self._dev is None:
            self._setup(self._graph_like())
        flat, K = self.flat, self.K
        eta_c = np.ones((flat.V, K, 2))
        tau_d = np.zeros((flat.V, K, self.Dmax))
        hidden, cont, nst = flat.var_hidden, flat.var_cont, flat.var_nstates      # (properties: evaluate once)
        if not (hidden & ~cont).any():
            # continuous variables only: one call draws the same stream as rand(K) per variable in order
            idx = np.flatnonzero(hidden)
            eta_c[idx, :, 0] = np.random.rand(idx.size, K) * 3 - 1.5
        else:
            for v in np.flatnonzero(hidden):
                if cont[v]:
                    eta_c[v, :, 0] = np.random.rand(K) * 3 - 1.5
                else:
                    D = int(nst[v])
                    tau_d[v, :, :D] = np.random.rand(K, D) * 10
        self._upload_params(np.zeros(K), eta_c, tau_d)

    # ---- gradient / free energy -------------------------------------------------------------------
    def _grad(self):
        d = self._dev
        _abi.check(_abi.lib().lhvi_vi_grad(self.dg.g, self.dg.p, self._struct(), _abi.ptr(d['g_w']), _abi.ptr(d['g_c']),
                                           _abi.ptr(d['g_d']), _abi.ptr(d['fe']), _abi.ptr(d['ws']), d['ws_bytes'],
                                           _abi.stream_ptr()))

    def free_energy(self):
        self._grad()
        return float(self._dev['fe'].item())

    def gradient_w_tau(self):
        self._grad()
        return self._dev['g_w'].cpu().numpy()

    def gradient_mu_var(self, rv):
        self._grad()
        return self._dev['g_c'][self._var_index(rv)].cpu().numpy()

    def gradient_category_tau(self, rv):
        self._grad()
        v = self._var_index(rv)
        return self._dev['g_d'][v, :, :int(self.flat.var_nstates[v])].cpu().numpy()

    # ---- optimisation -----------------------------------------------------------------------------
    def run(self, iteration=100, lr=0.1, is_log=True, log_fe=True):
        self.is_log, self.log_fe = is_log, log_fe
        self.init_param()
        self.alpha, self.b1, self.b2, self.eps = lr, 0.9, 0.999, 1e-8
        d = self._dev
        for name in ('w_tau', 'eta_c', 'tau_d'):
            d['m_' + name].zero_()
            d['s_' + name].zero_()
        self.t = 0
        if is_log:
            self.time_log, self.total_time = [], 0
        self.ADAM_update(iteration)

    def ADAM_update(self, iteration):
        """VI:249-300: all gradients from the pre-update parameters, then one ADAM step per array.  The loop never waits
        for the device: the free energy the reference logs after an update is the one the NEXT iteration's gradient pass
        computes anyway (same parameters), so it is copied into a device buffer there (one extra pass after the last
        update) and read back once at the end; ``time_log`` times are the loop's CPU time spread evenly over its updates."""
        d, l = self._dev, _abi.lib()
        torch = _abi.require_gpu()
        log_dev = bool(self.is_log and self.log_fe)
        fe_buf = torch.empty(max(iteration, 1), dtype=torch.float64, device=self.dg.device) if log_dev else None
        start = time.process_time()
        if self.fused_loop and (log_dev or not self.is_log):
            # the whole loop enqueued by one call: per update one gradient pass and one launch for the three ADAM steps, the
            # variance clip and the softmaxes (lhvi_vi_adam_run) -- same arithmetic as the per-array calls below
            _abi.check(l.lhvi_vi_adam_run(self.dg.g, self.dg.p, self._struct(), self._opt_struct(), int(iteration),
                                          _abi.ptr(fe_buf) if log_dev else None, _abi.ptr(d['ws']), d['ws_bytes'], _abi.stream_ptr()))
            self.t += iteration
            self._cache = {}
            if log_dev and iteration > 0:
                fes = fe_buf.cpu().numpy()
                elapsed = time.process_time() - start
                for i in range(iteration):
                    self.total_time += elapsed / iteration
                    if self.verbose:
                        print(float(fes[i]), self.total_time)
                    self.time_log.append([self.total_time, float(fes[i])])
            return
        for i in range(iteration):
            self.t += 1
            self._grad()
            if log_dev and i > 0:
                fe_buf[i - 1:i].copy_(d['fe'])
            st = _abi.stream_ptr()
            d['g_c'].mul_(d['mask_c'])
            d['g_d'].mul_(d['mask_d'])
            for name, grad, stride in (('w_tau', 'g_w', 0), ('eta_c', 'g_c', 2), ('tau_d', 'g_d', 0)):
                _abi.check(l.lhvi_adam_step(_abi.ptr(d[name]), _abi.ptr(d['m_' + name]), _abi.ptr(d['s_' + name]),
                                            _abi.ptr(d[grad]), d[name].numel(), self.t, float(self.alpha), self.b1, self.b2,
                                            self.eps, stride, float(self.var_threshold), st))
            self._refresh()
            if self.is_log and not self.log_fe:          # the reference's other log: -log phi at the current MAP (host queries)
                self.total_time += time.process_time() - start
                from .utils import log_likelihood
                fe = log_likelihood(self._ground_graph(), {rv: self.map(rv) for rv in self._ground_graph().rvs})
                if self.verbose:
                    print(fe, self.total_time)
                self.time_log.append([self.total_time, fe])
                start = time.process_time()
        if log_dev and iteration > 0:
            self._grad()
            fe_buf[iteration - 1:iteration].copy_(d['fe'])
            fes = fe_buf.cpu().numpy()                   # the only synchronisation of the loop
            elapsed = time.process_time() - start
            for i in range(iteration):
                self.total_time += elapsed / iteration
                if self.verbose:
                    print(float(fes[i]), self.total_time)
                self.time_log.append([self.total_time, float(fes[i])])

    def GD_update(self, iteration, lr):
        """VI:302-331: plain gradient descent on the same gradients"""
        d = self._dev
        for _ in range(iteration):
            self._grad()
            d['w_tau'].sub_(d['g_w'] * lr)
            d['eta_c'].sub_(d['g_c'] * d['mask_c'] * lr)
            d['eta_c'][:, :, 1].clamp_(min=self.var_threshold)
            d['tau_d'].sub_(d['g_d'] * d['mask_d'] * lr)
            self._refresh()

    # ---- reference-style views --------------------------------------------------------------------
    def _host(self, name):
        if name not in self._cache:
            self._cache[name] = self._dev[name].cpu().numpy()
        return self._cache[name]

    @property
    def eta(self):
        flat = self.flat
        out = {}
        for v, rv in enumerate(flat.rvs):
            if self._cont[v]:
                out[rv] = self._host('eta_c')[v]
            elif self._disc[v]:
                out[rv] = self._host('eta_d')[v, :, :int(flat.var_nstates[v])]
        return out

    @property
    def eta_tau(self):
        """category logits of the discrete hidden variables, {rv: array [K, #states]} like the reference's ``eta_tau``"""
        flat = self.flat
        return {rv: self._host('tau_d')[v, :, :int(flat.var_nstates[v])] for v, rv in enumerate(flat.rvs) if self._disc[v]}

    def _w_host(self):
        return self._host('w')

    @property
    def w(self):
        return self._host('w') if self._dev is not None else np.zeros(self.K)

    @property
    def w_tau(self):
        return self._host('w_tau') if self._dev is not None else np.zeros(self.K)

    @staticmethod
    def norm_pdf(x, eta):
        u = x - eta[0]
        return np.e ** (-u * u * 0.5 / eta[1]) / (2.506628274631 * eta[1])

    def rvs_belief(self, x, rvs):
        """VI:336-353 on the host (a K-term sum; the heavy expectation lives on the device)"""
        b = np.copy(self._w_host())
        for i, rv in enumerate(rvs):
            if rv.value is not None:
                if x[i] != rv.value:
                    return 0
                continue
            v = self._var_index(rv)
            if rv.domain.continuous:
                eta = self._host('eta_c')[v]
                xi = x[i] if np.ndim(x[i]) == 0 else float(np.ravel(x[i])[0])     # scipy's optimisers pass 1-element arrays
                for k in range(self.K):
                    b[k] *= self.norm_pdf(xi, eta[k])
            else:
                d = rv.domain.values.index(x[i])
                b *= self._host('eta_d')[v, :, d]
        return np.sum(b)

    def belief(self, x, rv):
        return self.rvs_belief((x,), (rv,))

    def map(self, rv):
        """VI:355-376"""
        if rv.value is not None:
            return rv.value
        if rv.domain.continuous:
            from scipy.optimize import minimize
            mus = self._host('eta_c')[self._var_index(rv)][:, 0]
            p = {x: self.belief(x, rv) for x in mus}
            x0 = max(p.keys(), key=lambda k: p[k])
            return minimize(lambda val: -self.belief(val, rv), x0=np.array([x0]), options={'disp': False})['x'][0]
        p = {x: self.belief(x, rv) for x in rv.domain.values}
        return max(p.keys(), key=lambda k: p[k])

    def _row_belief(self, v, x):
        """``belief(x, rv)`` of row `v` of the solver's graph (VI:333-353 for a single hidden variable)"""
        b = np.copy(self._w_host())
        if self.flat.var_cont[v]:
            eta = self._host('eta_c')[v]
            xi = x if np.ndim(x) == 0 else float(np.ravel(x)[0])
            for k in range(self.K):
                b[k] *= self.norm_pdf(xi, eta[k])
        else:
            b *= self._host('eta_d')[v, :, int(x)]
        return np.sum(b)

    def map_rows(self):
        """``map`` (VI:355-376, C2FVI:439-461) of every hidden ROW of the solver's graph -- one ``scipy.optimize.minimize`` per
        row instead of one per ground variable (the members of a cluster share its parameters, hence its answer).  Returns an
        array [V]: the maximiser for a continuous row, the state VALUE for a discrete one, NaN for an observed one."""
        from scipy.optimize import minimize
        flat = self.flat
        out = np.full(flat.V, np.nan)
        for v in np.flatnonzero(flat.var_hidden):
            if flat.var_cont[v]:
                p = {x: self._row_belief(v, x) for x in self._host('eta_c')[v][:, 0]}
                x0 = max(p.keys(), key=lambda k: p[k])
                out[v] = minimize(lambda val: -self._row_belief(v, val), x0=np.array([x0]), options={'disp': False})['x'][0]
            else:
                d = flat.var_dom[v]
                vals = flat.dom_val[flat.dom_ptr[d]:flat.dom_ptr[d + 1]]
                p = [self._row_belief(v, i) for i in range(vals.size)]
                out[v] = vals[int(np.argmax(p))]
        return out

    def rvs_map(self, rvs):
        """VI:378-456: coordinate ascent on the joint mixture belief"""
        from scipy.optimize import minimize
        res = {}
        for rv in rvs:
            if rv.value is not None:
                res[rv] = rv.value
                continue
            cand = self._host('eta_c')[self._var_index(rv)][:, 0] if rv.domain.continuous else rv.domain.values
            b = {v: self.belief(v, rv) for v in cand}
            res[rv] = max(b.keys(), key=lambda x: b[x])
        K = self.K

        def comp(rv, x):
            v = self._var_index(rv)
            if rv.domain.continuous:
                eta = self._host('eta_c')[v]
                return np.array([self.norm_pdf(x, eta[k]) for k in range(K)])
            return self._host('eta_d')[v, :, rv.domain.values.index(x)].copy()

        b = np.copy(self._w_host())
        for rv in rvs:
            if rv.value is None:
                b *= comp(rv, res[rv])
        for _ in range(10):
            for rv in rvs:
                if rv.value is not None:
                    continue
                b /= comp(rv, res[rv])
                if rv.domain.continuous:
                    new_x = minimize(lambda x: -float(np.sum(b * comp(rv, x))), x0=np.array([res[rv]]),
                                     options={'disp': False})['x']
                    res[rv] = new_x
                else:
                    scores = {x: float(np.sum(b * comp(rv, x))) for x in rv.domain.values}
                    res[rv] = max(scores.keys(), key=lambda x: scores[x])
                b *= comp(rv, res[rv])
        return res


class VarInference(_Variational):
    """Ground solver (``VarInference.py``)."""

    def __init__(self, g, num_mixtures=5, num_quadrature_points=3):
        self.g = g
        self._init_common(num_mixtures, num_quadrature_points)

    def _graph_like(self):
        return self.g

    def _ground_graph(self):
        return self.g

    def _var_index(self, rv):
        return self.flat.var_index[rv]


class LiftedVarInference(_Variational):
    """Lifted solver (``LiftedVarInference.py``): colour passing once, then the same step with cluster multiplicities.
    Queries take ground rvs and go through ``rv.cluster`` like the reference."""

    def __init__(self, g, num_mixtures=5, num_quadrature_points=3):
        from .lifting import CompressedGraph
        self._ground = g
        self.g = CompressedGraph(g)
        self.g.run()
        self._init_common(num_mixtures, num_quadrature_points)

    def _graph_like(self):
        return self.g

    def _ground_graph(self):
        return self._ground

    def _var_index(self, rv):
        c = getattr(rv, 'cluster', None)
        return self.flat.var_index[c if c in self.flat.var_index else rv]
