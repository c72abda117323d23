"""Soft-logic connectives and Markov-logic potentials (API of ``/root/reference/MLNPotential.py:6-58``).

Truth values live in [0, 1]; the connectives are the product / probabilistic-sum family, and a weighted formula turns
into the potential ``exp(w * formula(x))`` (hard formulas: the indicator ``formula(x) > 0``).  On the device a formula is
traced once per arity into postfix bytecode (``lhvi.expr``), which the generic f2v kernel interprets in log space.
"""
from __future__ import annotations

from math import e

from . import expr
from .graph import Potential
from .potentials import POT_MLN, POT_MLN_HARD


class SoftLogic:
    """the connectives, as arithmetic on truth values (plain numbers, NumPy arrays or ``expr`` tracers alike)"""

    @staticmethod
    def conj(p, q):
        return p * q

    @staticmethod
    def disj(p, q):
        return p + q - p * q

    @staticmethod
    def neg(p):
        return 1 - p

    @staticmethod
    def implies(p, q):
        return SoftLogic.disj(1 - p, q)

    @staticmethod
    def iff(p, q):
        return SoftLogic.implies(p, q) * SoftLogic.implies(q, p)

    @staticmethod
    def near(p, q):
        """soft equality of two real values: a negated squared distance (0 when equal)"""
        return -(p - q) ** 2


# the reference's names
and_op, or_op, neg_op = SoftLogic.conj, SoftLogic.disj, SoftLogic.neg
imp_op, bic_op, eq_op = SoftLogic.implies, SoftLogic.iff, SoftLogic.near


class _FormulaPotential(Potential):
    """a potential defined by a Python formula over the factor's arguments; ``kind`` selects the device interpretation"""

    kind = None

    def __init__(self, formula, w):
        Potential.__init__(self, symmetric=False)
        self.formula, self.w = formula, w
        self._programs = {}

    def _program(self, arity):
        if arity not in self._programs:
            self._programs[arity] = expr.trace(self.formula, arity)
        return self._programs[arity]

    def get(self, parameters):
        return self._value(self.formula(parameters))

    cq_max_states = 4096        # joint discrete states a conditional-quadratic view may tabulate

    def _program_for(self, domains):
        """the formula's program over these domains: traced once per arity; a formula that branches on its discrete arguments,
        once per joint discrete state of THESE domains (``expr.trace_by_state``)"""
        try:
            return self._program(len(domains))
        except expr.FormulaNotTraceable:
            roles = tuple(None if d.continuous else tuple(d.values) for d in domains)
            key = ('by state', roles)
            if key not in self._programs:
                self._programs[key] = expr.trace_by_state(self.formula, roles)
            return self._programs[key]

    def device_spec(self, domains):
        """parameter row ``[w, ncode, cq_off, (op, val) * ncode, conditional-quadratic block]`` (csrc/potential.hpp): ``cq_off`` =
        offset of the block from the start of the row, 0 when the formula has none -- with it the device evaluates the formula
        as a table lookup and at most six multiply-adds, never through the bytecode (which the CPU oracle keeps interpreting)"""
        program = self._program_for(domains)
        tail = self._cq_tail(program, domains)
        return self.kind, [float(self.w), float(len(program) // 2), float(3 + len(program) if tail else 0)] + program + tail

    def _cq_tail(self, program, domains):
        return []


class MLNPotential(_FormulaPotential):
    kind = POT_MLN

    def __init__(self, formula, w=1):
        _FormulaPotential.__init__(self, formula, w)

    def _value(self, truth):
        return e ** (truth * self.w)

    def _cq_tail(self, program, domains):
        """conditional-quadratic view of the formula (``expr.cq_block``), appended behind the bytecode: the evaluators
        ignore it, the f -> v work-list builder routes such factors to the quadratic-family kernels (csrc/pbp.hip)"""
        roles = [None if d.continuous else tuple(d.values) for d in domains]
        states = 1
        for r in roles:
            states *= 1 if r is None else len(r)
        if states > self.cq_max_states:
            return []
        try:
            return expr.cq_block(program, roles, self.w)
        except expr.NotConditionallyQuadratic:
            return []

    def to_log_potential(self):
        return MLNLogPotential(self.formula, self.w)


class MLNHardPotential(_FormulaPotential):
    kind = POT_MLN_HARD

    def __init__(self, formula):
        _FormulaPotential.__init__(self, formula, 0.0)

    def _value(self, truth):
        return 1 if truth > 0 else 0


class MLNLogPotential:
    """the log of an ``MLNPotential`` as a callable: ``w * formula(args)``"""

    def __init__(self, formula, w=1):
        self.formula, self.w = formula, w

    def __call__(self, args):
        return self.formula(args) * self.w
