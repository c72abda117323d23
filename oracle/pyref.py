"""Pure-Python, dict-of-dicts restatement of the reference's particle sweep (``EPBP.run``).

TEST INFRASTRUCTURE / BASELINE ONLY.  Nothing under the product package imports this file; ``bench.py`` uses it only
for the ``cpu_baseline.python`` leg (after the timed region), ``tests/test_oracle_pbp.py`` pins it against the golden
vectors captured from the reference.

Why it exists: the reference's files cannot travel to the GPU box, so the
"reference-equivalent pure-Python path" timed there is this restatement: the same data structures (one Python ``dict``
per message keyed by the float value of the point, ``(rv, f)`` / ``(f, rv)`` tuple keys, ``itertools.product`` over the
neighbours' particles, one ``potential.get`` call per joint assignment, ``statistics.mean`` in the balance step) and
the same arithmetic in the same order as ``/root/reference/EPBPLogVersion.py``; single-threaded like the original.
Every method cites the lines it follows.  It walks any object model with the reference's attribute names
(``g.rvs``, ``g.factors``, ``rv.nb``, ``rv.value``, ``rv.domain.{values,continuous,integral_points}``, ``f.nb``,
``f.potential.get``).
"""
from itertools import product
from math import e, log, sqrt
from statistics import mean

import numpy as np

INF = float('inf')


def _pdf(x, mu, sd):
    """EPBPLogVersion.py:49-53 (``sd`` is a standard deviation here)"""
    u = (x - mu) / sd
    return np.exp(-u * u * 0.5) / (2.506628274631 * sd)


class DictEPBP:
    var_threshold = 3          # EPBPLogVersion.py:17-18
    max_log_value = 700

    def __init__(self, g, n, proposal_approximation='EP'):
        self.g, self.n, self.approx = g, n, proposal_approximation
        self.message, self.sample, self.q, self.eta_message = {}, {}, {}, {}
        self.old_sample = None

    # ---- proposals ------------------------------------------------------------------------------
    def initial_proposal(self):
        """EPBPLogVersion.py:72-81"""
        for rv in self.g.rvs:
            if rv.value is None:
                self.q[rv] = (0, 5)
                site = (0, 5 * len(rv.nb))
                for f in rv.nb:
                    self.eta_message[(f, rv)] = site

    @staticmethod
    def _product(sites):
        """gaussian_product, EPBPLogVersion.py:30-41"""
        mu, sig = 0, 0
        for mu_, sig_ in sites:
            sig += sig_ ** -1
            mu += sig_ ** -1 * mu_
        sig = sig ** -1
        return sig * mu, sig

    @staticmethod
    def _division(a, b):
        """gaussian_division, EPBPLogVersion.py:43-47"""
        sig = a[1] * b[1] / (b[1] - a[1])
        return (a[0] * (b[1] + sig) - b[0] * sig) / b[1], sig

    def _moments(self, f, rv, cavity=None):
        """eta_approximation_simple (:103-121) and the tilted moments of eta_approximation (:137-151)"""
        m = self.message[(f, rv)]
        pts = rv.domain.integral_points
        if cavity is None:
            weight = [e ** m[x] for x in pts]
        else:
            param = (cavity[0], sqrt(cavity[1]))
            weight = [e ** m[x] * _pdf(x, *param) for x in pts]
        z = sum(weight)
        mu = sig = 0
        for w, x in zip(weight, pts):
            mu += w * x
            sig += w * x ** 2
        mu = mu / z
        return mu, sig / z - mu ** 2

    def _site(self, f, rv):
        """EPBPLogVersion.py:123-154"""
        if self.approx != 'EP':
            return self._moments(f, rv)
        a, b = self.q[rv], self.eta_message[(f, rv)]
        if a[1] >= b[1]:
            return self._moments(f, rv)
        cavity = self._division(a, b)
        return self._division(self._moments(f, rv, cavity), cavity)

    def update_proposal(self):
        """EPBPLogVersion.py:83-101"""
        for rv in self.g.rvs:
            if rv.value is None and rv.domain.continuous:
                sites = []
                floor = len(rv.nb) * self.var_threshold
                for f in rv.nb:
                    mu, sig = self._site(f, rv)
                    if 0 < sig < INF:
                        sig = max(sig, floor)
                        self.eta_message[(f, rv)] = (mu, sig)
                    else:
                        mu, sig = self.eta_message[(f, rv)]
                    sites.append((mu, sig))
                self.q[rv] = self._product(sites)

    # ---- messages -------------------------------------------------------------------------------
    def _weight(self, x, rv):
        """important_weight, EPBPLogVersion.py:156-163"""
        if x == rv.domain.values[0] or x == rv.domain.values[1]:
            return 1e-200
        return 1 / max(_pdf(x, self.q[rv][0], sqrt(self.q[rv][1])), 1e-200)

    def _rv_to_f(self, x, rv, f):
        """message_rv_to_f, EPBPLogVersion.py:165-174"""
        res = 0
        for nb in rv.nb:
            if nb != f:
                res += self.message[(nb, rv)][x]
        return res + log(self._weight(x, rv))

    def _balance(self, m):
        """log_message_balance, EPBPLogVersion.py:204-215"""
        values = m.values()
        mean_m, max_m = mean(values), max(values)
        shift = max_m - self.max_log_value if max_m - mean_m > self.max_log_value else mean_m
        for k, v in m.items():
            m[k] = v - shift

    def _f_to_rv(self, x, f, rv, sample):
        """message_f_to_rv, EPBPLogVersion.py:176-194"""
        res = 0
        param = []
        for nb in f.nb:
            if nb == rv:
                param.append((x,))
            elif nb.value is None:
                param.append(sample[nb])
            else:
                param.append((nb.value,))
        for x_join in product(*param):
            m = 0
            for idx, nb in enumerate(f.nb):
                if nb != rv and nb.value is None:
                    m += self.message[(nb, f)][x_join[idx]]
            res += f.potential.get(x_join) * e ** m
        return log(res) if res > 0 else -700

    # ---- the sweep, split into the phases of EPBP.run (EPBPLogVersion.py:225-289) -------------------
    def start(self, first_sample):
        """:226-242: initial proposal, first sample, zero messages on sample and integral points"""
        self.initial_proposal()
        self.sample = first_sample
        for rv in self.g.rvs:
            if rv.value is None:
                for f in rv.nb:
                    m = {k: 0 for k in self.sample[rv]}
                    if rv.domain.continuous:
                        self.message[(f, rv)] = {**m, **{k: 0 for k in rv.domain.integral_points}}
                    else:
                        self.message[(f, rv)] = m
                    self.message[(rv, f)] = m

    def v2f_half(self):
        """:249-257"""
        for rv in self.g.rvs:
            if rv.value is None:
                for f in rv.nb:
                    m = {}
                    for point in self.sample[rv]:
                        m[point] = self._rv_to_f(point, rv, f)
                    self._balance(m)
                    self.message[(rv, f)] = m

    def install(self, new_sample):
        """:268-270"""
        self.old_sample = self.sample
        self.sample = new_sample

    def f2v_factor(self, f):
        """body of the f -> rv loop for one factor, :273-282"""
        for rv in f.nb:
            if rv.value is None:
                m = {}
                for point in self.sample[rv]:
                    m[point] = self._f_to_rv(point, f, rv, self.old_sample)
                if rv.domain.continuous:
                    for point in rv.domain.integral_points:
                        m[point] = self._f_to_rv(point, f, rv, self.old_sample)
                self.message[(f, rv)] = m

    def f2v_half(self):
        for f in self.g.factors:
            self.f2v_factor(f)

    def belief_rv(self, x, rv):
        """:196-202"""
        res = 0
        for f in rv.nb:
            res += self._f_to_rv(x, f, rv, self.sample)
        return res

    def run(self, iterations, samples, on_iteration=None):
        """``samples[k]``: dict rv -> points of the k-th ``generate_sample`` call (injected; the reference draws them)"""
        self.start(samples[0])
        for i in range(iterations):
            self.v2f_half()
            if i < iterations - 1:
                self.update_proposal()
                self.install(samples[i + 1])
                if on_iteration:
                    on_iteration(i, self)
                self.f2v_half()
            elif on_iteration:
                on_iteration(i, self)


def sample_dicts(rvs, array):
    """[V, n] array of a golden fixture -> the reference's ``sample`` dict (discrete rvs: their domain values)"""
    out = {}
    for i, rv in enumerate(rvs):
        if rv.value is None:
            out[rv] = array[i][~np.isnan(array[i])] if rv.domain.continuous else rv.domain.values
    return out


def objects_from_flat(flat, api, potentials):
    """object graph (``api.RV`` / ``api.F`` / ``api.Graph``, potentials from ``potentials``) of a ground FlatGraph whose
    potential rows are quadratic / hybrid-quadratic / table -- the benchmark generator's families (lhvi/synth.py)"""
    POT_TABLE, POT_QUADRATIC, POT_HYBRID = potentials.POT_TABLE, potentials.POT_QUADRATIC, potentials.POT_HYBRID_QUADRATIC
    pots = []
    for k in range(flat.pot_kind.size):
        p = flat.pot_param[flat.pot_off[k]:flat.pot_off[k + 1]]
        kind = int(flat.pot_kind[k])
        if kind == POT_QUADRATIC:
            n = int(p[0])
            pots.append(potentials.QuadraticPotential(p[1:1 + n * n].reshape(n, n), p[1 + n * n:1 + n * n + n], float(p[-1])))
        elif kind == POT_HYBRID:
            nd, nc = int(p[0]), int(p[1])
            dims = tuple(int(x) for x in p[2:2 + nd])
            cnt = int(np.prod(dims))
            o = 2 + nd
            A = p[o:o + cnt * nc * nc].reshape(dims + (nc, nc))
            b = p[o + cnt * nc * nc:o + cnt * nc * nc + cnt * nc].reshape(dims + (nc,))
            c = p[o + cnt * nc * nc + cnt * nc:].reshape(dims)
            pots.append(potentials.HybridQuadraticPotential(A, b, c))
        elif kind == POT_TABLE:
            nd = int(p[0])
            dims = tuple(int(x) for x in p[1:1 + nd])
            pots.append(potentials.TablePotential(p[1 + nd:].reshape(dims)))
        else:
            raise NotImplementedError('potential kind %d' % kind)
    rvs = []
    for v in range(flat.V):
        d = flat.domains[flat.var_dom[v]]
        val = flat.var_value[v]
        rvs.append(api.RV(d, None if np.isnan(val) else (float(val) if d.continuous else int(val))))
    fs = [api.F(pots[flat.fac_pot[f]], [rvs[v] for v in flat.edge_var[flat.fac_ptr[f]:flat.fac_ptr[f + 1]]])
          for f in range(flat.F)]
    g = api.Graph()
    g.rvs, g.factors = rvs, fs
    g.init_nb()
    return g


def time_sweep(flat, n, api, potentials, budget_s=12.0, seed=0):
    """one sweep of ``DictEPBP`` ('simple' proposal) on the object graph of `flat`, the f -> rv half bounded to
    `budget_s` seconds of factors.  Returns dict(edges, v2f_s, proposal_s, f2v_s, f2v_edges_done, edge_messages_per_sec)
    where the rate extrapolates the f -> rv half linearly to all hidden edges."""
    import time
    g = objects_from_flat(flat, api, potentials)
    rng = np.random.default_rng(seed)

    def draw():
        return {rv: (np.clip(rng.normal(0.0, sqrt(5.0), n), rv.domain.values[0], rv.domain.values[1])
                     if rv.domain.continuous else rv.domain.values) for rv in g.rvs if rv.value is None}
    bp = DictEPBP(g, n, 'simple')
    bp.start(draw())
    t0 = time.perf_counter()
    bp.v2f_half()
    t1 = time.perf_counter()
    bp.update_proposal()
    t2 = time.perf_counter()
    bp.install(draw())
    done = 0
    hidden_edges = sum(1 for f in g.factors for rv in f.nb if rv.value is None)
    t3 = time.perf_counter()
    for f in g.factors:
        bp.f2v_factor(f)
        done += sum(1 for rv in f.nb if rv.value is None)
        if time.perf_counter() - t3 > budget_s:
            break
    t4 = time.perf_counter()
    f2v_full = (t4 - t3) * hidden_edges / max(done, 1)
    sweep_s = (t1 - t0) + (t2 - t1) + f2v_full
    return dict(edges=int(flat.E), hidden_edges=hidden_edges, v2f_s=t1 - t0, proposal_s=t2 - t1, f2v_s=t4 - t3,
                f2v_edges_done=done, sweep_s_extrapolated=sweep_s, edge_messages_per_sec=2.0 * flat.E / sweep_s)
