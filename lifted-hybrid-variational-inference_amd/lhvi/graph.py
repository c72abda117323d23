"""Factor-graph object model (host side, plain Python).

Drop-in surface for the reference's ``Graph.py`` (``/root/reference/Graph.py:11-209``):
``Domain``, ``Potential``, ``RV``, ``F``, ``Graph`` keep the reference's constructor
signatures and attribute names, because the solvers' callers build models through them
(SURVEY.md section 8(b)).  Nothing here touches the GPU; solvers flatten these objects
into CSR arrays (``lhvi.flat``) before any kernel runs.
"""
from __future__ import annotations

import itertools
from abc import ABC, abstractmethod
from math import log

import numpy as np


class Domain:
    """Value set of a random variable (``Graph.py:11-19``).

    Discrete: ``values`` is the tuple of states.  Continuous: ``values`` is ``(lo, hi)`` and
    ``integral_points`` is the grid messages are tabulated on (default: 30 evenly spaced points).
    Domains are compared by identity, exactly like the reference (its ``__eq__`` is commented out),
    which colour passing relies on (``CompressedGraphWithObs.py:193-199``).
    """

    def __init__(self, values, continuous=False, integral_points=None):
        self.values = tuple(values)
        self.continuous = continuous
        if continuous:
            if integral_points is None:
                integral_points = np.linspace(values[0], values[1], 30)
            self.integral_points = integral_points


class Potential(ABC):
    """Non-negative factor function phi(x) (``Graph.py:32-51``)."""

    def __init__(self, symmetric=False):
        self.symmetric = symmetric
        self.alpha = 0.001  # finite-difference step of gradient()/log_gradient()

    @abstractmethod
    def get(self, parameters):
        ...

    def _shifted(self, parameters, wrt):
        p = np.array(parameters)
        return p, p + np.array(wrt) * self.alpha

    def gradient(self, parameters, wrt):
        p, q = self._shifted(parameters, wrt)
        return (self.get(q) - self.get(p)) / self.alpha

    def log_gradient(self, parameters, wrt):
        p, q = self._shifted(parameters, wrt)
        return (log(self.get(q)) - log(self.get(p))) / self.alpha


class _Node:
    """Shared id / ordering / printing behaviour of RV and F."""

    def __lt__(self, other):
        return self.id < other.id

    def __repr__(self):
        return str(self)


class RV(_Node):
    """Random variable (``Graph.py:54-90``): ``value is None`` means hidden, otherwise evidence."""

    id_counter = itertools.count()

    def __init__(self, domain, value=None):
        self.domain = domain
        self.value = value
        self.id = next(RV.id_counter)
        self.nb = []          # incident factors, filled by Graph.init_nb()
        self.N = 0            # degree
        self.cluster = None   # set by colour passing
        self.belief_params_ = {}
        self.belief_params = {}
        self.sharing_count = 1

    @property
    def dstates(self):
        return None if self.domain.continuous else len(self.domain.values)

    @property
    def domain_type(self):
        return 'c-g' if self.domain.continuous else 'd-' + str(self.dstates)

    @property
    def values(self):
        return np.array(self.domain.values)

    def __str__(self):
        return '{} rv #{}'.format(self.domain_type, self.id)


class F(_Node):
    """Factor (``Graph.py:93-129``): ``potential`` over the ordered scope ``nb``."""

    id_counter = itertools.count()

    def __init__(self, potential=None, nb=None, potential_fun=None, log_potential=None, log_potential_fun=None):
        self.potential = potential
        self.potential_fun = potential_fun
        self.log_potential = log_potential
        self.log_potential_fun = log_potential_fun
        self.nb = [] if nb is None else nb
        self.id = next(F.id_counter)
        self.cluster = None
        self.sharing_count = 1

    @property
    def nb_domain_types(self):
        return tuple(rv.domain_type for rv in self.nb)

    @property
    def domain_type(self):
        kinds = {t[0] for t in self.nb_domain_types}
        if not kinds:
            return None
        if kinds == {'d'}:
            return 'd'
        if kinds == {'c'}:
            return 'c'
        return 'h'

    def __str__(self):
        return '{}factor #{}'.format(self.domain_type, self.id)


class Graph:
    """Ground factor graph (``Graph.py:137-209``).  ``rvs`` / ``factors`` may be sets or lists."""

    def __init__(self):
        self.rvs = set()
        self.factors = set()
        self._sorted_cache = {}

    def _sorted(self, name):
        # the reference memoises with lru_cache on a property; cache on identity+length instead so that
        # re-assigning g.rvs / g.factors after construction is picked up
        items = getattr(self, name)
        key = (id(items), len(items))
        hit = self._sorted_cache.get(name)
        if hit is None or hit[0] != key:
            hit = (key, sorted(items))
            self._sorted_cache[name] = hit
        return hit[1]

    @property
    def rvs_list(self):
        return self._sorted('rvs')

    @property
    def factors_list(self):
        return self._sorted('factors')

    def init_nb(self):
        """Fill ``rv.nb`` (factors in id order) and ``rv.N`` (``Graph.py:142-154``)."""
        for rv in self.rvs:
            rv.nb = []
        for f in self.factors_list:
            for rv in f.nb:
                rv.nb.append(f)
        for rv in self.rvs:
            rv.N = len(rv.nb)

    def init_rv_indices(self):
        """Discrete / continuous index tables used by the osi stack (``Graph.py:174-209``)."""
        self.Vd = [rv for rv in self.rvs_list if not rv.domain.continuous]
        self.Vc = [rv for rv in self.rvs_list if rv.domain.continuous]
        self.Nd, self.Nc = len(self.Vd), len(self.Vc)
        self.Vd_idx = {rv: i for i, rv in enumerate(self.Vd)}
        self.Vc_idx = {rv: i for i, rv in enumerate(self.Vc)}
        for f in self.factors_list:
            f.disc_nb_idx = tuple(self.Vd_idx[rv] for rv in f.nb if not rv.domain.continuous)
            f.cont_nb_idx = tuple(self.Vc_idx[rv] for rv in f.nb if rv.domain.continuous)
        self.dstates = [rv.dstates for rv in self.Vd]
