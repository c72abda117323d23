"""Python face of the CPU oracle.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module; the product package never does (tests/test_no_oracle_in_product.py
greps for it).  The arithmetic lives in ``oracle/c/*.c`` (plain C, flat arrays, each function citing the
reference lines it restates) and, for the integer colour refinement, in the pure-Python functions below.
Pinned against the golden vectors in ``tests/golden`` captured from the reference by
``oracle/capture_golden.py``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, 'liboracle.so')


def build():
    subprocess.check_call(['make', '-s', '-C', _HERE, 'liboracle.so'])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
    return _lib


class OGraph(C.Structure):
    _fields_ = [('V', C.c_int32), ('F', C.c_int32), ('E', C.c_int32), ('nnz', C.c_int32),
                ('fac_ptr', C.c_void_p), ('edge_var', C.c_void_p), ('edge_fac', C.c_void_p),
                ('edge_canon', C.c_void_p), ('var_ptr', C.c_void_p), ('var_edge', C.c_void_p),
                ('edge_count', C.c_void_p), ('fac_pot', C.c_void_p), ('var_value', C.c_void_p),
                ('pot_kind', C.c_void_p), ('pot_off', C.c_void_p), ('pot_param', C.c_void_p),
                ('var_dom', C.c_void_p), ('var_mult', C.c_void_p), ('fac_mult', C.c_void_p),
                ('dom_cont', C.c_void_p), ('dom_lo', C.c_void_p), ('dom_hi', C.c_void_p),
                ('dom_ptr', C.c_void_p), ('dom_val', C.c_void_p)]


class OPbp(C.Structure):
    _fields_ = [('n', C.c_int32), ('T', C.c_int32), ('flags', C.c_uint32),
                ('var_threshold', C.c_double), ('max_log_value', C.c_double),
                ('particles', C.c_void_p), ('old_particles', C.c_void_p), ('np', C.c_void_p),
                ('uniq', C.c_void_p), ('q', C.c_void_p)]


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class HostGraph:
    """keeps the numpy arrays of a FlatGraph alive next to the C struct that points at them"""

    def __init__(self, flat):
        self.flat = flat
        self.arrs = {k: np.ascontiguousarray(getattr(flat, k)) for k in
                     ('fac_ptr', 'edge_var', 'edge_fac', 'edge_canon', 'var_ptr', 'var_edge', 'edge_count', 'fac_pot',
                      'var_value', 'pot_kind', 'pot_off', 'pot_param', 'var_dom', 'var_mult', 'fac_mult',
                      'dom_cont', 'dom_lo', 'dom_hi', 'dom_ptr', 'dom_val')}
        g = OGraph()
        g.V, g.F, g.E, g.nnz = flat.V, flat.F, flat.E, int(flat.var_edge.size)
        for k, a in self.arrs.items():
            if k in ('edge_count', 'var_mult', 'fac_mult') and not flat.lifted:
                setattr(g, k, None)
            else:
                setattr(g, k, _p(a))
        self.g = g


def gabp_run(flat, iterations):
    """(f2v [E,2], v2f [E,2], mu_var [V,2]) after ``iterations`` flooding sweeps (GaBP.run / GaLBP.run)"""
    hg = HostGraph(flat)
    f2v = np.zeros((flat.E, 2))
    v2f = np.zeros((flat.E, 2))
    mv = np.zeros((flat.V, 2))
    l = lib()
    l.oracle_gabp_run(C.byref(hg.g), _p(f2v), _p(v2f), C.c_int(iterations))
    l.oracle_gabp_marginals(C.byref(hg.g), _p(f2v), _p(mv))
    return f2v, v2f, mv


def gabp_half_sweeps(flat, f2v, v2f):
    """one v2f half sweep then one f2v half sweep on copies; returns (v2f', f2v')"""
    hg = HostGraph(flat)
    f2v, v2f = f2v.copy(), v2f.copy()
    l = lib()
    l.oracle_gabp_v2f(C.byref(hg.g), _p(f2v), _p(v2f))
    l.oracle_gabp_f2v(C.byref(hg.g), _p(v2f), _p(f2v))
    return v2f, f2v


# ---- colour refinement (integer, exact) --------------------------------------------------------
def _dense(keys):
    table, out = {}, np.zeros(len(keys), dtype=np.int32)
    for i, k in enumerate(keys):
        out[i] = table.setdefault(k, len(table))
    return out, len(table)


def refine_factors(flat, symmetric, rv_color, f_color):
    """SuperF.split_by_structure for every cluster (CompressedGraphWithObs.py:152-175): clusters never merge, so
    the old colour is part of the key; the scope's colours are sorted iff the potential is symmetric."""
    keys = []
    for f in range(flat.F):
        nb = tuple(int(rv_color[flat.edge_var[e]]) for e in range(flat.fac_ptr[f], flat.fac_ptr[f + 1]))
        if symmetric[f]:
            nb = tuple(sorted(nb))
        keys.append((int(f_color[f]), nb))
    return _dense(keys)


def refine_rvs(flat, f_color, rv_color):
    """SuperRV.split_by_structure (CompressedGraphWithObs.py:47-76): key = sorted multiset of the colours of the
    incident factors (argument position ignored), within the old cluster."""
    keys = []
    for v in range(flat.V):
        nb = sorted(int(f_color[flat.edge_fac[flat.var_edge[k]]]) for k in range(flat.var_ptr[v], flat.var_ptr[v + 1]))
        keys.append((int(rv_color[v]), tuple(nb)))
    return _dense(keys)


def color_passing(flat, symmetric, rv_color, f_color):
    """CompressedGraph.run loop (CompressedGraphWithObs.py:264-271): stop when #rv clusters is unchanged"""
    rv_color, f_color = np.asarray(rv_color), np.asarray(f_color)
    n_rv = int(rv_color.max()) + 1 if rv_color.size else 0
    prev = -1
    while prev != n_rv:
        prev = n_rv
        f_color, _ = refine_factors(flat, symmetric, rv_color, f_color)
        rv_color, n_rv = refine_rvs(flat, f_color, rv_color)
    return rv_color, f_color


def canonical_labels(color):
    """label every item with the smallest index sharing its colour (the form the golden files use)"""
    color = np.asarray(color)
    first = {}
    for i, c in enumerate(color.tolist()):
        first.setdefault(c, i)
    return [first[c] for c in color.tolist()]


# ---- particle BP --------------------------------------------------------------------------------
PBP_EP = 1
PBP_EPBP_DISCRETE = 2


def pbp_counts(flat, n):
    """valid particle count per variable: n (continuous hidden), #states (discrete hidden), 0 (observed)"""
    nst = flat.var_nstates
    return np.where(flat.var_hidden, np.where(flat.var_cont, n, nst), 0).astype(np.int32)


def pbp_T(flat):
    sizes = np.diff(flat.dom_ptr)
    cont = flat.dom_cont.astype(bool)
    return int(sizes[cont].max()) if cont.any() else 0


class PbpOracle:
    """Flat-array twin of EPBP / HybridLBP state; every method is one reference function applied to all edges."""

    def __init__(self, flat, n, ep, epbp, var_threshold):
        self.flat, self.hg = flat, HostGraph(flat)
        self.n, self.T = n, pbp_T(flat)
        self.S = self.n + self.T
        self.flags = (PBP_EP if ep else 0) | (PBP_EPBP_DISCRETE if epbp else 0)
        self.var_threshold = float(var_threshold)
        self.np = pbp_counts(flat, n)
        self.particles = np.zeros((flat.V, n))
        self.old_particles = np.zeros((flat.V, n))
        self.uniq = np.zeros((flat.V, n), dtype=np.uint8)
        self.q = np.full((flat.V, 2), np.nan)
        self.eta = np.full((flat.E, 2), np.nan)
        self.f2v = np.zeros((flat.E, self.S))
        self.v2f = np.zeros((flat.E, n))

    def _s(self):
        s = OPbp()
        s.n, s.T, s.flags = self.n, self.T, self.flags
        s.var_threshold, s.max_log_value = self.var_threshold, 700.0
        s.particles, s.old_particles = _p(self.particles), _p(self.old_particles)
        s.np, s.uniq, s.q = _p(self.np), _p(self.uniq), _p(self.q)
        return s

    def set_particles(self, particles):
        """install a new sample (rows of observed variables are ignored); discrete rows = the domain states"""
        self.old_particles = self.particles
        p = np.array(particles, dtype=np.float64)
        flat = self.flat
        for v in np.flatnonzero(flat.var_hidden & ~flat.var_cont):
            d = flat.var_dom[v]
            vals = flat.dom_val[flat.dom_ptr[d]:flat.dom_ptr[d + 1]]
            p[v, :vals.size] = vals
        self.particles = np.ascontiguousarray(np.nan_to_num(p, nan=0.0))
        self.uniq = np.zeros((flat.V, self.n), dtype=np.uint8)
        lib().oracle_pbp_uniq(C.byref(self.hg.g), C.c_int(self.n), _p(self.particles), _p(self.np), _p(self.uniq))

    def init(self):
        lib().oracle_pbp_init(C.byref(self.hg.g), C.byref(self._s()), _p(self.eta), _p(self.q), _p(self.f2v), _p(self.v2f))

    def step_v2f(self):
        lib().oracle_pbp_v2f(C.byref(self.hg.g), C.byref(self._s()), _p(self.f2v), _p(self.v2f))

    def step_proposal(self):
        lib().oracle_pbp_proposal(C.byref(self.hg.g), C.byref(self._s()), _p(self.f2v), _p(self.eta), _p(self.q))

    def step_f2v(self):
        lib().oracle_pbp_f2v(C.byref(self.hg.g), C.byref(self._s()), _p(self.v2f), _p(self.f2v))

    def belief_points(self, qvar, x):
        qvar = np.ascontiguousarray(qvar, dtype=np.int32)
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.zeros_like(x)
        lib().oracle_pbp_belief_points(C.byref(self.hg.g), C.byref(self._s()), _p(self.v2f), C.c_int(qvar.size),
                                       _p(qvar), C.c_int(x.shape[1]), _p(x), _p(out))
        return out

    def edge_points(self, qedge, x):
        qedge = np.ascontiguousarray(qedge, dtype=np.int32)
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(qedge.size, -1)
        out = np.zeros_like(x)
        lib().oracle_pbp_edge_points(C.byref(self.hg.g), C.byref(self._s()), _p(self.v2f), C.c_int(qedge.size), _p(qedge),
                                     C.c_int(x.shape[1]), _p(x), _p(out))
        return out

    def run(self, iterations, samples, on_iteration=None):
        """EPBP.run / HybridLBP.run(c2f=-1) with injected samples: samples[0] is the initial draw, samples[i+1] the
        draw of iteration i (EPBP.py:225-289)."""
        self.init()
        self.set_particles(samples[0])
        for i in range(iterations):
            self.step_v2f()
            if i < iterations - 1:
                self.step_proposal()
                self.set_particles(samples[i + 1])
                if on_iteration:
                    on_iteration(i, self)     # v2f of iteration i, proposal updated, new particles installed
                self.step_f2v()
            elif on_iteration:
                on_iteration(i, self)


def interval_probability(log_belief, a, b, lo, hi, max_log_value=700.0):
    """EPBP.probability / HybridLBP.probability (EPBPLogVersion.py:356-375, HybridLBPLogVersion.py:384-403) on top of
    ``log_area`` (EPBPLogVersion.py:291-308): 20-point trapezoid of exp(belief_rv - shift) on the domain, shift =
    log_message_balance of the tabulated values (mean, or max - 700; all 20 keys are distinct indices), then the 5-point
    trapezoid on [a, b] with the same shift.  `log_belief(xs) -> log-beliefs at the points xs`."""
    def log_area(a, b, n, shift=None):
        x = np.linspace(a, b, n)
        d = x[1] - x[0]
        y = [float(t) for t in log_belief(x)]
        if shift is None:
            mean_m, max_m = sum(y) / len(y), max(y)
            shift = max_m - max_log_value if max_m - mean_m > max_log_value else mean_m
        y = [t - shift for t in y]
        res, prev = 0, np.e ** y[0]
        for i in range(1, n):
            cur = np.e ** y[i]
            res += (prev + cur) * d
            prev = cur
        return res * 0.5, shift
    z, shift = log_area(lo, hi, 20)
    num, _ = log_area(a, b, 5, shift)
    return num / z


# ---- mixture variational inference ----------------------------------------------------------------
class OVi(C.Structure):
    _fields_ = [('K', C.c_int32), ('T', C.c_int32), ('Dmax', C.c_int32), ('quirks', C.c_int32),
                ('gh_x', C.c_void_p), ('gh_w', C.c_void_p), ('w', C.c_void_p), ('eta_c', C.c_void_p),
                ('eta_d', C.c_void_p), ('obs_var', C.c_void_p)]


def softmax_rows(x):
    """VarInference.softmax(x, 1) (VI:32-38): e ** x / sum"""
    r = np.e ** x
    return r / np.sum(r, 1)[:, np.newaxis]


class ViOracle:
    """Flat-array twin of VarInference / LiftedVarInference (parameters for every variable, unused rows ignored)."""

    def __init__(self, flat, K, T, quirks=1, obs_var=None):
        """`obs_var` [V]: variance of Gaussian observation clusters (C2FVarInference), 0 elsewhere; None = none"""
        from numpy.polynomial.hermite import hermgauss
        self.flat, self.hg, self.K, self.T, self.quirks = flat, HostGraph(flat), K, T, quirks
        self.obs_var = None if obs_var is None else np.ascontiguousarray(obs_var, dtype=np.float64)
        self.gh_x, self.gh_w = hermgauss(T)
        self.gh_w = self.gh_w / np.sqrt(np.pi)
        disc = flat.var_hidden & ~flat.var_cont
        self.Dmax = int(flat.var_nstates[disc].max()) if disc.any() else 1
        self.cont = flat.var_hidden & flat.var_cont
        self.disc = disc
        self.nst = np.where(disc, flat.var_nstates, 0)
        self.w_tau = np.zeros(K)
        self.eta_c = np.zeros((flat.V, K, 2))
        self.tau_d = np.zeros((flat.V, K, self.Dmax))
        self.refresh()

    def set_params(self, w_tau, eta_c, tau_d):
        self.w_tau = np.array(w_tau, dtype=float)
        self.eta_c = np.ascontiguousarray(np.nan_to_num(np.array(eta_c, dtype=float), nan=1.0))
        td = np.zeros((self.flat.V, self.K, self.Dmax))
        src = np.nan_to_num(np.array(tau_d, dtype=float), nan=0.0)
        td[:, :, :min(self.Dmax, src.shape[2])] = src[:, :, :self.Dmax]
        self.tau_d = td
        self.refresh()

    def refresh(self):
        r = np.e ** self.w_tau
        self.w = r / np.sum(r, 0)
        self.eta_d = np.zeros_like(self.tau_d)
        for v in np.flatnonzero(self.disc):
            d = self.nst[v]
            self.eta_d[v, :, :d] = softmax_rows(self.tau_d[v, :, :d])

    def grad(self):
        p = OVi()
        p.K, p.T, p.Dmax, p.quirks = self.K, self.T, self.Dmax, self.quirks
        self._keep = [np.ascontiguousarray(a) for a in (self.gh_x, self.gh_w, self.w, self.eta_c, self.eta_d)]
        p.gh_x, p.gh_w, p.w, p.eta_c, p.eta_d = (_p(a) for a in self._keep)
        p.obs_var = _p(self.obs_var) if self.obs_var is not None else None
        g_w = np.zeros(self.K)
        g_c = np.zeros((self.flat.V, self.K, 2))
        g_d = np.zeros((self.flat.V, self.K, self.Dmax))
        fe = np.zeros(1)
        lib().oracle_vi_grad(C.byref(self.hg.g), C.byref(p), _p(g_w), _p(g_c), _p(g_d), _p(fe))
        g_c[~self.cont] = 0.0
        g_d[~self.disc] = 0.0
        return g_w, g_c, g_d, float(fe[0])

    def run(self, iterations, lr=0.1, moments=None, t0=0):
        """ADAM_update (VI:249-300); returns the free energy logged after every update.  `moments`: dict with the first /
        second moment arrays m_w_tau, s_w_tau, m_eta_c, s_eta_c, m_tau_d, s_tau_d to continue from (updated in place;
        C2FVarInference carries them across its rounds); `t0`: updates already done (bias correction)"""
        b1, b2, eps = 0.9, 0.999, 1e-8
        if moments is None:
            moments = {}
        mw, sw = moments.setdefault('m_w_tau', np.zeros(self.K)), moments.setdefault('s_w_tau', np.zeros(self.K))
        mc, sc = moments.setdefault('m_eta_c', np.zeros_like(self.eta_c)), moments.setdefault('s_eta_c', np.zeros_like(self.eta_c))
        md, sd = moments.setdefault('m_tau_d', np.zeros_like(self.tau_d)), moments.setdefault('s_tau_d', np.zeros_like(self.tau_d))
        log = []
        adam = lib().oracle_adam_step
        adam.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_double, C.c_double,
                         C.c_double, C.c_double, C.c_int, C.c_double]
        for t in range(t0 + 1, t0 + iterations + 1):
            g_w, g_c, g_d, _ = self.grad()
            for theta, m, s, g, stride in ((self.w_tau, mw, sw, g_w, 0), (self.eta_c, mc, sc, g_c, 2),
                                           (self.tau_d, md, sd, g_d, 0)):
                g = np.ascontiguousarray(g)
                adam(_p(theta), _p(m), _p(s), _p(g), theta.size, t, lr, b1, b2, eps, stride, 0.1)
            self.refresh()
            log.append(self.grad()[3])
        return log
