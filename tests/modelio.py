"""Serialise a factor-graph object model to plain JSON-able data and back.

Works on any object model with the reference's attribute names (``rv.domain/value``, ``f.potential/nb``,
``domain.values/continuous/integral_points``), so the same code dumps a model built with the
reference's classes (in ``oracle/capture_golden.py``) and rebuilds it with this repository's classes
(in the tests).  MLN formulas are referred to by name through ``FORMULAS``.
"""
import numpy as np


def eq_op(x, y):
    return -(x - y) ** 2


FORMULAS = {
    'eq1': lambda x: eq_op(x[0], 1),
    'x0_eq12': lambda x: x[0] * eq_op(x[1], x[2]),
    'x0': lambda x: x[0],
    'x0_eq1c': lambda x: x[0] * eq_op(x[1], 0.341),
    'nand': lambda x: (1 - x[0]) + (1 - x[1]) - (1 - x[0]) * (1 - x[1]),
    'any3': lambda x: 1 - (x[0] == 0) * (x[1] == 0) * (x[2] == 0),
    # the robot-mapping template (Demo/Data/HMLN/GeneratorRobotMapping.py:41-75)
    'rm_aligned': lambda x: 1 - (x[0] == 1) * (x[1] == 1) * (x[2] == 0) * (x[3] == 1) * (1 - x[4]),
    'x0_eq_0.1': lambda x: x[0] * eq_op(x[1], 0.1),
    'x0_eq_0.02': lambda x: x[0] * eq_op(x[1], 0.02),
    'x0_eq_0.001': lambda x: x[0] * eq_op(x[1], 0.001),
}
_FORMULA_NAME = {id(f): k for k, f in FORMULAS.items()}


def _pot_to_dict(p):
    name = type(p).__name__
    if name == 'TablePotential':
        t = p.table
        if isinstance(t, dict):
            return {'cls': name, 'dict': [[list(k), float(v)] for k, v in t.items()], 'symmetric': bool(p.symmetric)}
        return {'cls': name, 'table': np.asarray(t, dtype=float).tolist(), 'symmetric': bool(p.symmetric)}
    if name == 'GaussianPotential':
        return {'cls': name, 'mu': np.asarray(p.mu, dtype=float).tolist(), 'sig': np.asarray(p.sig, dtype=float).tolist()}
    if name in ('LinearGaussianPotential', 'X2Potential', 'XYPotential'):
        return {'cls': name, 'coeff': float(p.coeff), 'sig': float(p.sig)}
    if name in ('QuadraticPotential', 'HybridQuadraticPotential'):
        return {'cls': name, 'A': np.asarray(p.A, dtype=float).tolist(), 'b': np.asarray(p.b, dtype=float).tolist(),
                'c': np.asarray(p.c, dtype=float).tolist()}
    if name in ('MLNPotential', 'MLNHardPotential'):
        return {'cls': name, 'formula': _FORMULA_NAME[id(p.formula)], 'w': float(getattr(p, 'w', 1))}
    raise TypeError('cannot serialise potential %r' % name)


def _pot_from_dict(d, api):
    cls = getattr(api, d['cls'])
    name = d['cls']
    if name == 'TablePotential':
        if 'dict' in d:
            return cls({tuple(k): v for k, v in d['dict']}, symmetric=d['symmetric'])
        return cls(np.array(d['table']), symmetric=d['symmetric'])
    if name == 'GaussianPotential':
        return cls(d['mu'], d['sig'])
    if name in ('LinearGaussianPotential', 'X2Potential', 'XYPotential'):
        return cls(d['coeff'], d['sig'])
    if name in ('QuadraticPotential', 'HybridQuadraticPotential'):
        return cls(np.array(d['A']), np.array(d['b']), np.array(d['c']) if name[0] == 'H' else float(d['c']))
    if name == 'MLNPotential':
        return cls(FORMULAS[d['formula']], d['w'])
    if name == 'MLNHardPotential':
        return cls(FORMULAS[d['formula']])
    raise TypeError(name)


def dump_model(g):
    """g.rvs and g.factors must be *lists* (iteration order is part of the fixture)."""
    rvs, factors = list(g.rvs), list(g.factors)
    dom_ids, doms = {}, []
    pot_ids, pots = {}, []
    rv_index = {id(rv): i for i, rv in enumerate(rvs)}
    out_rvs = []
    for rv in rvs:
        d = rv.domain
        if id(d) not in dom_ids:
            dom_ids[id(d)] = len(doms)
            doms.append({'values': [float(x) for x in d.values], 'continuous': bool(d.continuous),
                         'integral_points': (np.asarray(d.integral_points, dtype=float).tolist()
                                             if d.continuous else None)})
        out_rvs.append([dom_ids[id(d)], None if rv.value is None else float(rv.value)])
    out_fs = []
    for f in factors:
        p = f.potential
        if id(p) not in pot_ids:
            pot_ids[id(p)] = len(pots)
            pots.append(_pot_to_dict(p))
        out_fs.append([pot_ids[id(p)], [rv_index[id(rv)] for rv in f.nb]])
    return {'domains': doms, 'potentials': pots, 'rvs': out_rvs, 'factors': out_fs}


def load_model(d, api):
    """Rebuild with ``api``'s classes (Domain, RV, F, Graph + potential classes).  Returns (g, rvs, factors)."""
    doms = []
    for x in d['domains']:
        vals = x['values']
        if not x['continuous'] and all(float(v).is_integer() for v in vals):
            vals = [int(v) for v in vals]
        doms.append(api.Domain(tuple(vals), continuous=x['continuous'],
                               integral_points=None if x['integral_points'] is None else np.array(x['integral_points'])))
    pots = [_pot_from_dict(p, api) for p in d['potentials']]
    rvs = []
    for di, val in d['rvs']:
        if val is not None and not doms[di].continuous and float(val).is_integer():
            val = int(val)
        rvs.append(api.RV(doms[di], val))
    factors = [api.F(pots[pi], [rvs[i] for i in nb]) for pi, nb in d['factors']]
    g = api.Graph()
    g.rvs = rvs
    g.factors = factors
    g.init_nb()
    return g, rvs, factors
