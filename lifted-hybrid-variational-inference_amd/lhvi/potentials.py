"""Potential classes (host objects) and their device encodings.

API surface of the reference's ``Potential.py`` (``/root/reference/Potential.py:15-424``): same class
names, constructor arguments and public attributes (``mu``/``sig``/``prec``, ``coeff``/``sig``,
``A``/``b``/``c``, ``table``), same ``get(parameters)`` values and the same ``__eq__``/``__hash__``
policy (value equality only for Linear/X2/XY; identity for everything else), because colour passing
groups factors by ``potential`` hash/eq (``CompressedGraphWithObs.py:227-234``).

New here: every class exposes ``device_spec(dom_sizes)`` -> ``(kind, params)``, the flat encoding the
HIP evaluators in ``csrc/potential.hpp`` switch on.  ``kind`` values are the ``LHVI_POT_*`` constants
of ``include/lhvi.h``.
"""
from __future__ import annotations

from math import e, exp, pi, pow, sqrt  # noqa: F401  (re-exported like the reference module)

import warnings

import numpy as np

from .graph import Potential

# potential kinds -- keep in sync with include/lhvi.h
POT_GENERIC = 0
POT_TABLE = 1
POT_GAUSSIAN = 2
POT_QUADRATIC = 3
POT_HYBRID_QUADRATIC = 4
POT_LINEAR_GAUSSIAN = 5
POT_X2 = 6
POT_XY = 7
POT_MLN = 8
POT_MLN_HARD = 9
POT_IMAGE_NODE = 10
POT_IMAGE_EDGE = 11

MAX_ARITY = 6


def mu_prec_to_quad_params(mu, prec):
    """(A, b, c) with x'Ax + b'x + c == -0.5 (x-mu)' prec (x-mu)  (``Potential.py:6-12``)."""
    mu, prec = np.asarray(mu), np.asarray(prec)
    b = prec @ mu
    return -0.5 * prec, b, -0.5 * np.dot(mu, b)


class _ValueEq:
    """Value-based equality for the three single-coefficient quadratic potentials."""

    def __hash__(self):
        return hash((self.coeff, self.sig))

    def __eq__(self, other):
        return self.__class__ == other.__class__ and self.coeff == other.coeff and self.sig == other.sig


class TablePotential(Potential):
    """phi(x) = table[x]; ``table`` is an ndarray indexed by state, or a dict keyed by value tuples."""

    def __init__(self, table, symmetric=False):
        Potential.__init__(self, symmetric=symmetric)
        self.table = table

    def get(self, parameters):
        return self.table[parameters]

    def to_log_potential(self):
        return LogTable(np.log(self.table))

    def device_spec(self, domains):
        dims = [len(d.values) for d in domains]
        if isinstance(self.table, dict):
            dense = np.zeros(dims)
            for idx in np.ndindex(*dims):
                key = tuple(d.values[i] for d, i in zip(domains, idx))
                dense[idx] = self.table[key]
        else:
            dense = np.asarray(self.table, dtype=np.float64)
            if list(dense.shape) != dims:
                raise ValueError('TablePotential shape %s does not match the scope domains %s' % (dense.shape, dims))
            for d in domains:
                if tuple(d.values) != tuple(range(len(d.values))):
                    # an ndarray table is indexed by the *values*; they must be 0..d-1 to be valid indices
                    raise ValueError('ndarray TablePotential needs domain values 0..d-1, got %r' % (d.values,))
        return POT_TABLE, [float(len(dims))] + [float(x) for x in dims] + dense.ravel().tolist()


class LogTable:
    def __init__(self, table):
        self.table = table

    def __call__(self, args):
        return self.table[tuple(args)]


class GaussianPotential(Potential):
    """phi(x) = w_coef * exp(-0.5 (x-mu)' sig^-1 (x-mu))  (``Potential.py:35-58``); ``get`` omits the
    normaliser unless ``use_coef``."""

    def __init__(self, mu, sig, w=1):
        Potential.__init__(self, symmetric=False)
        self.mu = np.array(mu)
        with warnings.catch_warnings():     # np.matrix on purpose: GaBP inverts it as `sig ** -1` (GaBP.py:44)
            warnings.simplefilter('ignore', PendingDeprecationWarning)
            self.sig = np.matrix(sig)
            self.prec = self.sig.I
        det = np.linalg.det(self.sig)
        if det == 0:
            raise NameError("The covariance matrix can't be singular")
        p = float(len(mu))
        self.coefficient = w / (pow(2 * pi, p * 0.5) * pow(det, 0.5))

    def get(self, parameters, use_coef=False):
        d = np.asarray(parameters, dtype=np.float64) - self.mu
        q = float(d @ np.asarray(self.prec) @ d)
        return (self.coefficient if use_coef else 1.) * pow(e, -0.5 * q)

    def get_quadratic_params(self):
        return mu_prec_to_quad_params(self.mu, self.prec)

    def to_log_potential(self):
        return LogQuadratic(*self.get_quadratic_params())

    def device_spec(self, domains):
        n = len(self.mu)
        # GaBP inverts the covariance itself as ``sig ** -1`` (GaBP.py:44); ship that matrix too so the
        # closed forms see the reference's bits rather than ``sig.I``'s
        with warnings.catch_warnings():
            warnings.simplefilter('ignore', PendingDeprecationWarning)
            inv = np.asarray(self.sig ** -1)
        return POT_GAUSSIAN, [float(n)] + self.mu.astype(np.float64).tolist() + \
            np.asarray(self.prec).ravel().tolist() + inv.ravel().tolist()


class QuadraticPotential(Potential):
    """phi(x) = exp(x'Ax + b'x + c)  (``Potential.py:69-110``)."""

    def __init__(self, A, b, c):
        Potential.__init__(self, symmetric=False)
        self.A = np.array(A)
        self.b = np.array(b)
        self.c = c
        self.log_potential = LogQuadratic(A, b, c)

    def to_log_potential(self):
        return self.log_potential

    def get(self, args, ignore_const=False):
        x = np.array(args)
        res = np.dot(x, self.A @ x) + np.dot(self.b, x)
        if not ignore_const:
            res += self.c
        return e ** res

    __call__ = get

    def get_quadratic_params(self):
        return self.A, self.b, self.c

    def dim(self):
        return self.b.size

    def device_spec(self, domains):
        n = int(self.b.size)
        return POT_QUADRATIC, [float(n)] + np.asarray(self.A, dtype=np.float64).ravel().tolist() + \
            np.asarray(self.b, dtype=np.float64).ravel().tolist() + [float(self.c)]


class LogQuadratic:
    """x'Ax + b'x + c on scalars or broadcastable NumPy arrays (``Potential.py:113-166``; the
    TensorFlow branch of the reference is out of scope)."""

    def __init__(self, A, b, c=0):
        self.A, self.b, self.c = A, b, c

    def __call__(self, args, ignore_const=False):
        xs = np.broadcast_arrays(*[np.asarray(a, dtype=np.float64) for a in args])
        A, b = np.asarray(self.A), np.asarray(self.b)
        res = 0.
        for i, xi in enumerate(xs):
            res = res + b[i] * xi
            for j, xj in enumerate(xs):
                res = res + A[i, j] * xi * xj
        if not ignore_const and self.c != 0:
            res = res + self.c
        return res


class LogGaussian:
    """-0.5 (x-mu)' prec (x-mu) (``Potential.py:169-217``)."""

    def __init__(self, mu, prec):
        self.mu = np.array(mu)
        self.prec = np.array(prec)

    def __call__(self, args):
        xs = np.broadcast_arrays(*[np.asarray(a, dtype=np.float64) for a in args])
        res = 0.
        for i, xi in enumerate(xs):
            for j, xj in enumerate(xs):
                res = res + self.prec[i, j] * (xi - self.mu[i]) * (xj - self.mu[j])
        return -0.5 * res


class LogHybridQuadratic:
    """exp-quadratic in the continuous arguments, one (A,b,c) per discrete configuration
    (``Potential.py:220-270``); arguments ordered [x_d, x_c]."""

    def __init__(self, A, b, c):
        self.A, self.b, self.c = A, b, c
        self.Nd = len(c.shape)
        self.Nc = b.shape[-1]

    def get_quadratic_params_given_x_d(self, x_d):
        x_d = tuple(x_d)
        return self.A[x_d], self.b[x_d], self.c[x_d]

    def get_table_params_given_x_c(self, x_c):
        outer = np.outer(x_c, x_c)
        return np.sum(self.A * outer, axis=(-1, -2)) + np.sum(self.b * x_c, axis=-1) + self.c

    def __call__(self, args, **kwargs):
        return LogQuadratic(*self.get_quadratic_params_given_x_d(args[:self.Nd]))(args[self.Nd:], **kwargs)


class HybridQuadraticPotential(Potential):
    """``Potential.py:273-308``; ``get`` takes [x_d..., x_c...] with x_d integer states."""

    def __init__(self, A, b, c):
        Potential.__init__(self, symmetric=False)
        self.A, self.b, self.c = A, b, c
        self.Nd = len(c.shape)
        self.Nc = int(b.shape[-1])
        self.log_potential = LogHybridQuadratic(A, b, c)

    def get(self, args, **kwargs):
        A, b, c = self.log_potential.get_quadratic_params_given_x_d(args[:self.Nd])
        return QuadraticPotential(A, b, c).get(args[self.Nd:], **kwargs)

    def to_log_potential(self):
        return self.log_potential

    def device_spec(self, domains):
        dims = list(np.asarray(self.c).shape)
        return POT_HYBRID_QUADRATIC, [float(self.Nd), float(self.Nc)] + [float(x) for x in dims] + \
            np.asarray(self.A, dtype=np.float64).ravel().tolist() + \
            np.asarray(self.b, dtype=np.float64).ravel().tolist() + \
            np.asarray(self.c, dtype=np.float64).ravel().tolist()


class LinearGaussianPotential(_ValueEq, Potential):
    """phi(x0, x1) = exp(-(x1 - coeff*x0)^2 / (2 sig))  (``Potential.py:311-338``)."""

    def __init__(self, coeff, sig):
        Potential.__init__(self, symmetric=False)
        self.coeff = coeff
        self.sig = sig

    def get(self, parameters):
        return np.exp(-(parameters[1] - self.coeff * parameters[0]) ** 2 * 0.5 / self.sig)

    def get_quadratic_params(self):
        a = self.coeff
        return mu_prec_to_quad_params(np.zeros(2), np.array([[a ** 2, -a], [-a, 1.]]) / self.sig)

    def to_log_potential(self):
        return LogQuadratic(*self.get_quadratic_params())

    def device_spec(self, domains):
        return POT_LINEAR_GAUSSIAN, [float(self.coeff), float(self.sig)]


class X2Potential(_ValueEq, Potential):
    """phi(x0) = exp(-coeff x0^2 / (2 sig))  (``Potential.py:341-368``)."""

    def __init__(self, coeff, sig):
        Potential.__init__(self, symmetric=False)
        self.coeff = coeff
        self.sig = sig

    def get(self, parameters):
        return np.exp(-self.coeff * parameters[0] ** 2 * 0.5 / self.sig)

    def get_quadratic_params(self):
        return mu_prec_to_quad_params(np.zeros(1), np.array([[self.coeff / self.sig]]))

    def to_log_potential(self):
        return LogQuadratic(*self.get_quadratic_params())

    def device_spec(self, domains):
        return POT_X2, [float(self.coeff), float(self.sig)]


class XYPotential(_ValueEq, Potential):
    """phi(x0, x1) = exp(-coeff x0 x1 / (2 sig)), symmetric  (``Potential.py:371-397``)."""

    def __init__(self, coeff, sig):
        Potential.__init__(self, symmetric=True)
        self.coeff = coeff
        self.sig = sig

    def get(self, parameters):
        return np.exp(-self.coeff * parameters[0] * parameters[1] * 0.5 / self.sig)

    def get_quadratic_params(self):
        return mu_prec_to_quad_params(np.zeros(2), np.array([[0., 0.5], [0.5, 0.]]) * self.coeff / self.sig)

    def to_log_potential(self):
        return LogQuadratic(*self.get_quadratic_params())

    def device_spec(self, domains):
        return POT_XY, [float(self.coeff), float(self.sig)]


class ImageNodePotential(Potential):
    """N(x0 - x1; mu, sig) with sig a standard deviation (``Potential.py:400-408``)."""

    def __init__(self, mu, sig):
        Potential.__init__(self, symmetric=True)
        self.mu, self.sig = mu, sig

    def get(self, parameters):
        u = (parameters[0] - parameters[1] - self.mu) / self.sig
        return exp(-u * u * 0.5) / (2.506628274631 * self.sig)

    def device_spec(self, domains):
        return POT_IMAGE_NODE, [float(self.mu), float(self.sig)]


class ImageEdgePotential(Potential):
    """Truncated-exponential smoothness prior (``Potential.py:411-424``)."""

    def __init__(self, distant_cof, scaling_cof, max_threshold):
        Potential.__init__(self, symmetric=True)
        self.distant_cof = distant_cof
        self.scaling_cof = scaling_cof
        self.max_threshold = max_threshold
        self.v = pow(e, -self.max_threshold / self.scaling_cof)

    def get(self, parameters):
        d = abs(parameters[0] - parameters[1])
        tail = self.v if d > self.max_threshold else pow(e, -d / self.scaling_cof)
        return d * self.distant_cof + tail

    def device_spec(self, domains):
        return POT_IMAGE_EDGE, [float(self.distant_cof), float(self.scaling_cof), float(self.max_threshold), float(self.v)]
