"""Soft-logic operators and MLN potentials (``/root/reference/MLNPotential.py:6-58``).

Same names and values as the reference.  ``MLNPotential.get`` is ``e ** (formula(x) * w)``.  For the
device the formula is traced once into postfix bytecode (``lhvi.expr``); the kernels then evaluate
``w * formula(x)`` in log space.
"""
from __future__ import annotations

from math import e

from . import expr
from .graph import Potential
from .potentials import POT_MLN, POT_MLN_HARD


def and_op(x, y):
    return x * y


def or_op(x, y):
    return x + y - x * y


def neg_op(x):
    return 1 - x


def imp_op(x, y):
    return or_op(1 - x, y)


def bic_op(x, y):
    return imp_op(x, y) * imp_op(y, x)


def eq_op(x, y):
    return -(x - y) ** 2


class _Traced:
    """Caches the device bytecode per arity."""

    def _program(self, arity):
        cache = self.__dict__.setdefault('_prog_cache', {})
        if arity not in cache:
            cache[arity] = expr.trace(self.formula, arity)
        return cache[arity]


class MLNPotential(_Traced, Potential):
    def __init__(self, formula, w=1):
        Potential.__init__(self, symmetric=False)
        self.formula = formula
        self.w = w

    def get(self, parameters):
        return e ** (self.formula(parameters) * self.w)

    def to_log_potential(self):
        return MLNLogPotential(self.formula, self.w)

    def device_spec(self, domains):
        prog = self._program(len(domains))
        return POT_MLN, [float(self.w), float(len(prog) // 2)] + prog


class MLNHardPotential(_Traced, Potential):
    def __init__(self, formula):
        Potential.__init__(self, symmetric=False)
        self.formula = formula

    def get(self, parameters):
        return 1 if self.formula(parameters) > 0 else 0

    def device_spec(self, domains):
        prog = self._program(len(domains))
        return POT_MLN_HARD, [0.0, float(len(prog) // 2)] + prog


class MLNLogPotential:
    def __init__(self, formula, w=1):
        self.formula = formula
        self.w = w

    def __call__(self, args):
        return self.formula(args) * self.w
