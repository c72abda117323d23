"""Trace a Python formula ``lambda x: ...`` into postfix bytecode for the device interpreter.

The reference's MLN potentials hold arbitrary Python callables (``MLNPotential.py:30-49``); the
formulas it actually ships (``Demo/Data/HMLN/Generator*.py``) are arithmetic over ``x[i]`` with
``+ - * ** ==`` and constants.  We run the callable once on tracer objects that overload those
operators and record a postfix program; ``csrc/potential.hpp::mln_eval`` executes it per joint
assignment.  A formula that branches on the values of its DISCRETE arguments (``1 if ... else 0``) is traced once per joint
discrete state (``trace_by_state``); one that branches on a continuous value cannot be traced and raises
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


class _MixedArgs(_Args):
    """arguments of a formula with its DISCRETE ones fixed to plain values (one joint state) and tracers for the rest"""

    def __init__(self, n, fixed):
        _Args.__init__(self, n)
        self.fixed = fixed

    def __getitem__(self, i):
        if not isinstance(i, slice):
            j = i + self.n if i < 0 else i
            if j in self.fixed:
                return self.fixed[j]
        return _Args.__getitem__(self, i)


MAX_PROGRAM_OPS = 1024      # longest program trace_by_state may emit (the device interpreter walks it per joint assignment)


def trace_by_state(formula, roles):
    """Program of a formula that BRANCHES on the values of its discrete arguments (``1 if x[0] + x[1] + x[2] > 0 else 0``: the
    reference keeps such variants beside the arithmetic ones, Demo/Data/HMLN/GeneratorRobotMapping.py:37,42; MLNPotential accepts
    any callable, MLNPotential.py:36-37).  ``roles[a]`` = the state values of a discrete argument, ``None`` for a continuous one.
    The formula is run once per joint discrete state with those arguments as the plain values they take -- every branch on them
    resolves -- and tracers for the continuous ones; the program is sum_state [x_disc == state] * program_state (states whose
    program is the constant 0 are left out).  A branch on a CONTINUOUS value still raises ``FormulaNotTraceable``."""
    import itertools
    n = len(roles)
    disc = [a for a, r in enumerate(roles) if r is not None]
    if not disc:
        raise FormulaNotTraceable('formula branches on a continuous value; cannot be compiled for the device')
    prog = []
    for states in itertools.product(*[roles[a] for a in disc]):
        fixed = dict(zip(disc, states))
        try:
            out = Sym.lift(formula(_MixedArgs(n, fixed)))
        except FormulaNotTraceable:
            raise
        except Exception as exc:
            raise FormulaNotTraceable('formula is not traceable: %s: %s' % (type(exc).__name__, exc))
        sub = out.prog
        if all(op != OP_ARG for op, _ in sub):             # a constant for this state: fold it
            flat = []
            for op, val in sub:
                flat += [float(op), float(val)]
            c = run(flat, [])
            if c == 0.0:
                continue
            sub = [(OP_CONST, float(c))]
        term = []
        for k, (a, v) in enumerate(fixed.items()):
            term += [(OP_ARG, float(a)), (OP_CONST, float(v)), (OP_EQ, 0.0)] + ([(OP_MUL, 0.0)] if k else [])
        term += sub + [(OP_MUL, 0.0)]
        prog += term + ([(OP_ADD, 0.0)] if prog else [])
    if not prog:
        prog = [(OP_CONST, 0.0)]
    if stack_depth(prog) > MAX_STACK:
        raise FormulaNotTraceable('formula needs an evaluation stack deeper than %d' % MAX_STACK)
    if len(prog) > MAX_PROGRAM_OPS:
        raise FormulaNotTraceable('formula resolved per discrete state needs %d operations (limit %d)' % (len(prog), MAX_PROGRAM_OPS))
    flat = []
    for op, val in prog:
        flat += [float(op), float(val)]
    return flat


class NotConditionallyQuadratic(Exception):
    pass


CQ_MAGIC = 17233.0          # first word of the conditional-quadratic block behind a formula's bytecode (csrc/potential.hpp)


class _Poly:
    """polynomial of total degree <= 2 in (at most) two continuous arguments: {(i, j): coefficient}"""

    __slots__ = ('t',)

    def __init__(self, t):
        self.t = {k: v for k, v in t.items() if v != 0.0 or k == (0, 0)}

    @property
    def const(self):
        return all(k == (0, 0) for k in self.t)

    def value(self):
        return self.t.get((0, 0), 0.0)

    def add(self, o, sign=1.0):
        t = dict(self.t)
        for k, v in o.t.items():
            t[k] = t.get(k, 0.0) + sign * v
        return _Poly(t)

    def mul(self, o):
        t = {}
        for (i, j), a in self.t.items():
            for (k, l), b in o.t.items():
                if a == 0.0 or b == 0.0:
                    continue
                if i + j + k + l > 2:
                    raise NotConditionallyQuadratic('degree above 2')
                t[(i + k, j + l)] = t.get((i + k, j + l), 0.0) + a * b
        return _Poly(t)


def conditional_quadratic(flat, roles):
    """Evaluate a traced program symbolically for every joint state of its discrete arguments.

    ``roles[a]`` = tuple of the state values of a discrete argument, or ``None`` for a continuous one.  Returns
    ``(dims, coef)``: ``coef[cfg]`` = ``(a00, axy, a11, b0, b1, c)`` (no continuous argument: only ``c``, a table in log space) with
    ``formula(x) = a00 u^2 + axy u v + a11 v^2 + b0 u + b1 v + c`` for ``u, v`` = the continuous arguments in argument
    order and ``cfg`` the mixed-radix index of the discrete states (first discrete argument most significant).
    Raises ``NotConditionallyQuadratic`` when some state's restriction is not such a polynomial (or there are no / more
    than two continuous arguments).  The reference's hybrid MLN formulas are all of this family:
    ``x[0] * eq_op(x[1], x[2])`` with ``eq_op(a, b) = -(a - b) ** 2`` (MLNPotential.py:26-27,
    Demo/Data/HMLN/GeneratorPaperPopularity.py:28-40, GeneratorRobotMapping.py:60-75)."""
    import itertools
    cont = [a for a, r in enumerate(roles) if r is None]
    disc = [a for a, r in enumerate(roles) if r is not None]
    if len(cont) > 2:
        raise NotConditionallyQuadratic('more than two continuous arguments')
    cpos = {a: i for i, a in enumerate(cont)}
    dims = [len(roles[a]) for a in disc]
    coef = []
    for states in itertools.product(*[range(d) for d in dims]):
        val = {a: float(roles[a][k]) for a, k in zip(disc, states)}
        st = []
        for i in range(0, len(flat), 2):
            op, v = int(flat[i]), flat[i + 1]
            if op == OP_ARG:
                a = int(v)
                st.append(_Poly({(0, 0): val[a]}) if a in val else _Poly({((1, 0) if cpos[a] == 0 else (0, 1)): 1.0}))
            elif op == OP_CONST:
                st.append(_Poly({(0, 0): v}))
            elif op == OP_NEG:
                st[-1] = _Poly({k: -c for k, c in st[-1].t.items()})
            elif op == OP_SQR:
                st[-1] = st[-1].mul(st[-1])
            elif op == OP_ABS:
                if not st[-1].const:
                    raise NotConditionallyQuadratic('abs of a non-constant')
                st[-1] = _Poly({(0, 0): abs(st[-1].value())})
            else:
                b = st.pop()
                a = st.pop()
                if op == OP_ADD:
                    st.append(a.add(b))
                elif op == OP_SUB:
                    st.append(a.add(b, -1.0))
                elif op == OP_MUL:
                    st.append(a.mul(b))
                elif op == OP_DIV:
                    if not b.const or b.value() == 0.0:
                        raise NotConditionallyQuadratic('division by a non-constant')
                    st.append(a.mul(_Poly({(0, 0): 1.0 / b.value()})))
                elif op == OP_POW:
                    if a.const and b.const:
                        st.append(_Poly({(0, 0): a.value() ** b.value()}))
                    elif b.const and b.value() in (0.0, 1.0, 2.0):
                        st.append({0.0: _Poly({(0, 0): 1.0}), 1.0: a, 2.0: a.mul(a)}[b.value()])
                    else:
                        raise NotConditionallyQuadratic('power of a non-constant')
                else:
                    if not (a.const and b.const):
                        raise NotConditionallyQuadratic('comparison of non-constants')
                    x, y = a.value(), b.value()
                    st.append(_Poly({(0, 0): float({OP_EQ: x == y, OP_NE: x != y, OP_LT: x < y, OP_LE: x <= y,
                                                    OP_GT: x > y, OP_GE: x >= y}[op])}))
        p = st[-1].t
        coef.append([p.get((2, 0), 0.0), p.get((1, 1), 0.0), p.get((0, 2), 0.0), p.get((1, 0), 0.0), p.get((0, 1), 0.0),
                     p.get((0, 0), 0.0)])
    return dims, coef


def cq_block(flat, roles, w):
    """the parameter block appended behind an MLN potential's bytecode when its formula is conditionally quadratic:
    ``[CQ_MAGIC, arity, Nd, Nc, role[arity], dims[Nd], coef[ncfg][6]]`` (Nc = 0, 1 or 2) with ``role[a]`` = index among the discrete
    arguments, or ``-1 - index`` among the continuous ones; coefficients already multiplied by the weight (log phi)"""
    dims, coef = conditional_quadratic(flat, roles)
    role, nd, nc = [], 0, 0
    for r in roles:
        if r is None:
            role.append(float(-1 - nc))
            nc += 1
        else:
            role.append(float(nd))
            nd += 1
    out = [CQ_MAGIC, float(len(roles)), float(nd), float(nc)] + role + [float(d) for d in dims]
    for row in coef:
        out += [float(w) * c for c in row]
    return out


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
