"""Trace a Python formula ``lambda x: ...`` into postfix bytecode for the device interpreter.

The reference's MLN potentials hold arbitrary Python callables (``MLNPotential.py:30-49``); the
formulas it actually ships (``Demo/Data/HMLN/Generator*.py``) are arithmetic over ``x[i]`` with
``+ - * ** ==`` and constants.  We run the callable once on tracer objects that overload those
operators and record a postfix program; ``csrc/potential.hpp::mln_eval`` executes it per joint
assignment.  Formulas that branch on values (``1 if ... else 0``) cannot be traced and raise
``FormulaNotTraceable`` -- the caller then fails loudly instead of silently evaluating on the CPU.
"""
from __future__ import annotations

import numpy as np

# opcodes -- keep in sync with csrc/potential.hpp
OP_ARG, OP_CONST, OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_POW, OP_NEG, OP_SQR, \
    OP_EQ, OP_NE, OP_LT, OP_LE, OP_GT, OP_GE, OP_ABS = range(16)

MAX_STACK = 12


class FormulaNotTraceable(Exception):
    pass


class Sym:
    """One node of the traced expression; ``prog`` is its postfix program as (op, operand) pairs."""

    __slots__ = ('prog',)
    __hash__ = None

    def __init__(self, prog):
        self.prog = prog

    @staticmethod
    def lift(v):
        if isinstance(v, Sym):
            return v
        if isinstance(v, (bool, int, float, np.integer, np.floating, np.bool_)):
            return Sym([(OP_CONST, float(v))])
        raise FormulaNotTraceable('unsupported operand %r' % (v,))

    def _bin(self, other, op, swap=False):
        a, b = Sym.lift(self), Sym.lift(other)
        if swap:
            a, b = b, a
        return Sym(a.prog + b.prog + [(op, 0.0)])

    def __add__(self, o): return self._bin(o, OP_ADD)
    def __radd__(self, o): return self._bin(o, OP_ADD, True)
    def __sub__(self, o): return self._bin(o, OP_SUB)
    def __rsub__(self, o): return self._bin(o, OP_SUB, True)
    def __mul__(self, o): return self._bin(o, OP_MUL)
    def __rmul__(self, o): return self._bin(o, OP_MUL, True)
    def __truediv__(self, o): return self._bin(o, OP_DIV)
    def __rtruediv__(self, o): return self._bin(o, OP_DIV, True)
    def __neg__(self): return Sym(self.prog + [(OP_NEG, 0.0)])
    def __pos__(self): return self
    def __abs__(self): return Sym(self.prog + [(OP_ABS, 0.0)])

    def __pow__(self, o):
        if isinstance(o, (int, float)) and o == 2:
            return Sym(self.prog + [(OP_SQR, 0.0)])   # float ** 2 is an exact multiply in CPython/libm
        return self._bin(o, OP_POW)

    def __rpow__(self, o): return self._bin(o, OP_POW, True)

    # comparisons produce 0/1 indicators, like bool arithmetic in the reference's formulas
    def __eq__(self, o): return self._bin(o, OP_EQ)
    def __ne__(self, o): return self._bin(o, OP_NE)
    def __lt__(self, o): return self._bin(o, OP_LT)
    def __le__(self, o): return self._bin(o, OP_LE)
    def __gt__(self, o): return self._bin(o, OP_GT)
    def __ge__(self, o): return self._bin(o, OP_GE)

    def __bool__(self):
        raise FormulaNotTraceable('formula branches on a value (if/and/or); cannot be compiled for the device')

    def __float__(self):
        raise FormulaNotTraceable('formula converts a symbolic value to float')

    __int__ = __index__ = __float__


class _Args:
    def __init__(self, n):
        self.n = n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(self.n))]
        if i < 0:
            i += self.n
        if not 0 <= i < self.n:
            raise IndexError(i)
        return Sym([(OP_ARG, float(i))])

    def __len__(self):
        return self.n

    def __iter__(self):
        return (self[i] for i in range(self.n))


def stack_depth(prog):
    depth = peak = 0
    for op, _ in prog:
        if op in (OP_ARG, OP_CONST):
            depth += 1
        elif op in (OP_NEG, OP_SQR, OP_ABS):
            pass
        else:
            depth -= 1
        peak = max(peak, depth)
    return peak


def trace(formula, arity):
    """Return the postfix program of ``formula`` over ``arity`` arguments as a flat float list
    ``[op0, val0, op1, val1, ...]``."""
    try:
        out = Sym.lift(formula(_Args(arity)))
    except FormulaNotTraceable:
        raise
    except Exception as exc:  # numpy ufuncs on Sym, math.exp(Sym), ...
        raise FormulaNotTraceable('formula is not traceable: %s: %s' % (type(exc).__name__, exc))
    if stack_depth(out.prog) > MAX_STACK:
        raise FormulaNotTraceable('formula needs an evaluation stack deeper than %d' % MAX_STACK)
    flat = []
    for op, val in out.prog:
        flat += [float(op), float(val)]
    return flat


def run(flat, x):
    """Host interpreter of a traced program (used by tests to check tracing against the lambda)."""
    st = []
    for i in range(0, len(flat), 2):
        op, val = int(flat[i]), flat[i + 1]
        if op == OP_ARG:
            st.append(float(x[int(val)]))
        elif op == OP_CONST:
            st.append(val)
        elif op == OP_NEG:
            st[-1] = -st[-1]
        elif op == OP_SQR:
            st[-1] = st[-1] * st[-1]
        elif op == OP_ABS:
            st[-1] = abs(st[-1])
        else:
            b = st.pop()
            a = st.pop()
            st.append({OP_ADD: lambda: a + b, OP_SUB: lambda: a - b, OP_MUL: lambda: a * b,
                       OP_DIV: lambda: a / b, OP_POW: lambda: a ** b,
                       OP_EQ: lambda: float(a == b), OP_NE: lambda: float(a != b),
                       OP_LT: lambda: float(a < b), OP_LE: lambda: float(a <= b),
                       OP_GT: lambda: float(a > b), OP_GE: lambda: float(a >= b)}[op]())
    return st[-1]
