"""First-order templates that ground to a factor graph (host side).

API of the reference's ``RelationalGraph.py:7-132``: ``LV``, ``Atom``, ``ParamF``, ``RelationalGraph(atoms,
parametric_factors)`` with ``ground_graph()`` and ``add_evidence(data)``.  Atom expressions are strings such as
``'PartOf(s,l)'`` or ``'SegType(s,$W)'`` (``$`` marks a constant).  Grounding is model construction, not the sweep:
it runs on the host.  ``ground_graph`` produces ordinary ``Graph`` objects like the reference; ``ground_flat`` is the
flat-array fast path: the same grounding written straight into a ``FlatGraph`` with NumPy
index arithmetic, no Python object per ground atom or factor.
"""
from __future__ import annotations

import re
from itertools import product

import numpy as np

from .graph import F, Graph, RV

_TOKEN = re.compile(r'\$?\w+')


class LV:
    """logical variable: a named set of instances"""

    def __init__(self, instances):
        self.instances = instances


class Atom:
    """relational atom: a predicate over logical variables with a value domain"""

    def __init__(self, domain, logical_variables, name=None):
        self.domain = domain
        self.lvs = logical_variables
        self.name = name


class ParamF:
    """parametric factor: a potential over atom expressions, optionally constrained"""

    def __init__(self, potential, nb=None, constrain=None):
        self.potential = potential
        self.constrain = constrain
        self.nb = [] if nb is None else nb


class RelationalGraph:
    def __init__(self, atoms, parametric_factors):
        self.atoms = atoms
        self.param_factors = parametric_factors
        self.atoms_dict = {atom.name: atom for atom in atoms}
        self.rvs_dict = dict()
        self.grounding = None

    @staticmethod
    def _parse(atom_expression):
        return _TOKEN.findall(atom_expression) if isinstance(atom_expression, str) else list(atom_expression)

    def atom_substitution(self, atom_expression, substitution):
        """ground one atom expression under `substitution`; creates the RV on first use (``RelationalGraph.py:42-65``)"""
        parts = self._parse(atom_expression)
        atom = self.atoms_dict[parts[0]]
        key = tuple([parts[0]] + [s[1:] if s[0] == '$' else substitution[s] for s in parts[1:]])
        rv = self.rvs_dict.get(key)
        if rv is None:
            rv = RV(atom.domain)
            self.rvs_dict[key] = rv
        return key, rv

    def extract_lvs(self, atom_expression, lvs=None):
        """logical variables (token -> instances) mentioned by an atom expression"""
        if type(lvs) is not dict:
            lvs = dict()
        parts = self._parse(atom_expression)
        atom = self.atoms_dict[parts[0]]
        for i in range(len(atom.lvs)):
            s = parts[i + 1]
            if s[0] != '$':
                lvs[s] = atom.lvs[i].instances
        return lvs

    @staticmethod
    def lvs_iter(lvs):
        tokens = list(lvs)
        for combo in product(*[lvs[t] for t in tokens]):
            yield dict(zip(tokens, combo))

    def add_evidence(self, data):
        """data: {(AtomName, instance, ...): value}; every other rv becomes hidden (``RelationalGraph.py:94-102``)"""
        for key, rv in self.rvs_dict.items():
            rv.value = data[key] if key in data else None
        return self.grounding, self.rvs_dict

    def ground_graph(self):
        """``RelationalGraph.py:104-132``: one ground factor per admissible substitution of every parametric factor"""
        factors = []
        for pf in self.param_factors:
            lvs = dict()
            for expr in pf.nb:
                self.extract_lvs(expr, lvs)
            parsed = [self._parse(expr) for expr in pf.nb]
            for sub in self.lvs_iter(lvs):
                if pf.constrain is None or pf.constrain(sub):
                    factors.append(F(potential=pf.potential, nb=[self.atom_substitution(p, sub)[1] for p in parsed]))
        g = Graph()
        g.rvs = set(self.rvs_dict.values())
        g.factors = set(factors)
        g.init_nb()
        self.grounding = g
        return g, self.rvs_dict

    # ---- flat-array grounding --------------------------------------------------------------------------------------
    def ground_flat(self, data=None):
        """The graph ``ground_graph()`` + ``add_evidence(data)`` + ``flatten`` would give, built without per-node objects.

        Same substitution order as ``ground_graph`` (parametric factors in order, ``itertools.product`` over the logical
        variables in first-mention order), so factor ids, variable ids (first use, like ``rvs_dict``) and the ``rv.nb``
        order of the variable CSR are those of the object path.  ``ParamF.constrain`` is called once per substitution
        unless it carries a true ``vectorized`` attribute, in which case it receives a dict of instance *arrays* and
        returns a boolean mask.  Returns ``(flat, keys)`` where ``keys.var_id(key)`` maps an atom key such as
        ``('loss', 'c1', 'b2')`` to its variable index (-1 if no factor mentions it) and ``keys.key_of(v)`` inverts it."""
        from .flat import build_flat
        atom_ids = {atom.name: i for i, atom in enumerate(self.atoms)}
        shapes = [tuple(len(lv.instances) for lv in atom.lvs) for atom in self.atoms]
        sizes = np.array([int(np.prod(sh)) if sh else 1 for sh in shapes], dtype=np.int64)
        base = np.concatenate([[0], np.cumsum(sizes)])
        inst_index = {}

        def index_of(lv):
            if id(lv) not in inst_index:
                inst_index[id(lv)] = {x: i for i, x in enumerate(lv.instances)}
            return inst_index[id(lv)]

        scopes, fac_pot_parts, pots, pot_ids, specs = [], [], [], {}, []
        for pf in self.param_factors:
            lvs = dict()
            for expr in pf.nb:
                self.extract_lvs(expr, lvs)
            tokens = list(lvs)
            dims = [len(lvs[t]) for t in tokens]
            nsub = int(np.prod(dims)) if dims else 1
            # index of every token's instance for every substitution, itertools.product (row-major) order
            idx = {}
            rep = nsub
            for t, dsz in zip(tokens, dims):
                rep //= dsz
                idx[t] = np.tile(np.repeat(np.arange(dsz, dtype=np.int64), rep), nsub // (dsz * rep))
            keep = None
            if pf.constrain is not None:
                if getattr(pf.constrain, 'vectorized', False):
                    keep = np.asarray(pf.constrain({t: np.asarray(lvs[t], dtype=object)[idx[t]] for t in tokens}), dtype=bool)
                else:
                    keep = np.fromiter((bool(pf.constrain({t: lvs[t][idx[t][i]] for t in tokens})) for i in range(nsub)),
                                       dtype=bool, count=nsub)
            cols = []
            for expr in pf.nb:
                parts = self._parse(expr)
                a = atom_ids[parts[0]]
                atom = self.atoms[a]
                lin = np.zeros(nsub, dtype=np.int64)
                for k, tok in enumerate(parts[1:]):
                    stride = int(np.prod(shapes[a][k + 1:])) if k + 1 < len(shapes[a]) else 1
                    if tok[0] == '$':
                        lin += index_of(atom.lvs[k])[tok[1:]] * stride
                    else:
                        if lvs[tok] is atom.lvs[k].instances:
                            lin += idx[tok] * stride
                        else:       # same token bound through another atom's LV object with equal instances
                            m = index_of(atom.lvs[k])
                            lin += np.array([m[x] for x in lvs[tok]], dtype=np.int64)[idx[tok]] * stride
                cols.append(base[a] + lin)
            sc = np.stack(cols, axis=1) if cols else np.zeros((nsub, 0), dtype=np.int64)
            if keep is not None:
                sc = sc[keep]
            doms = tuple(self.atoms_dict[self._parse(expr)[0]].domain for expr in pf.nb)
            pkey = (id(pf.potential), tuple(id(d) for d in doms))       # same table key as flatten()
            if pkey not in pot_ids:
                spec = getattr(pf.potential, 'device_spec', None)
                if spec is None:
                    raise NotImplementedError('potential %r has no device encoding (device_spec)' % type(pf.potential).__name__)
                pot_ids[pkey] = len(pots)
                pots.append(pf.potential)
                specs.append(spec(doms))
            scopes.append(sc)
            fac_pot_parts.append(np.full(sc.shape[0], pot_ids[pkey], dtype=np.int32))
        arity = np.concatenate([np.full(sc.shape[0], sc.shape[1], dtype=np.int64) for sc in scopes]) if scopes else np.zeros(0, np.int64)
        fac_ptr = np.concatenate([[0], np.cumsum(arity)]).astype(np.int32)
        dense = np.concatenate([sc.ravel() for sc in scopes]) if scopes else np.zeros(0, np.int64)
        # variable ids in first-use order (the insertion order of rvs_dict)
        used, first = np.unique(dense, return_index=True)
        order = np.argsort(first, kind='stable')
        rank = np.empty(used.size, dtype=np.int64)
        rank[order] = np.arange(used.size)
        edge_var = rank[np.searchsorted(used, dense)].astype(np.int32)
        var_dense = used[order]                                   # dense atom-instance id of every variable
        var_atom = (np.searchsorted(base, var_dense, side='right') - 1).astype(np.int32)
        domains, dom_ids = [], {}
        atom_dom = np.zeros(len(self.atoms), dtype=np.int32)
        for i, atom in enumerate(self.atoms):
            if id(atom.domain) not in dom_ids:
                dom_ids[id(atom.domain)] = len(domains)
                domains.append(atom.domain)
            atom_dom[i] = dom_ids[id(atom.domain)]
        keys = GroundKeys(self.atoms, atom_ids, shapes, base, used, rank, var_dense, var_atom, index_of)
        var_value = np.full(var_dense.size, np.nan)
        if data:
            for key, val in data.items():
                v = keys.var_id(key)
                if v >= 0:
                    var_value[v] = float(val)
        flat = build_flat(fac_ptr, edge_var, np.concatenate(fac_pot_parts) if fac_pot_parts else np.zeros(0, np.int32),
                          specs, var_value, atom_dom[var_atom], domains)
        flat.potentials = pots
        return flat, keys


class GroundKeys:
    """atom key <-> variable index of a ``ground_flat`` result, by index arithmetic (no dict of all ground atoms)"""

    def __init__(self, atoms, atom_ids, shapes, base, used, rank, var_dense, var_atom, index_of):
        self.atoms, self.atom_ids, self.shapes, self.base = atoms, atom_ids, shapes, base
        self.used, self.rank, self.var_dense, self.var_atom, self._index_of = used, rank, var_dense, var_atom, index_of

    def var_id(self, key):
        a = self.atom_ids.get(key[0])
        if a is None or len(key) - 1 != len(self.shapes[a]):
            return -1                   # (evidence about something the model does not ground is ignored, RelationalGraph.py:94-102)
        atom, sh = self.atoms[a], self.shapes[a]
        lin = 0
        for k, inst in enumerate(key[1:]):
            i = self._index_of(atom.lvs[k]).get(inst)
            if i is None:
                return -1
            lin = lin * sh[k] + i
        d = self.base[a] + lin
        i = int(np.searchsorted(self.used, d))
        return int(self.rank[i]) if i < self.used.size and self.used[i] == d else -1

    def var_ids(self, name, lin):
        """vectorised ``var_id``: variable indices of the instances of atom `name` with row-major instance numbers `lin`
        (over the atom's logical variables); -1 where no factor mentions the instance"""
        d = self.base[self.atom_ids[name]] + np.asarray(lin, dtype=np.int64)
        i = np.minimum(np.searchsorted(self.used, d), self.used.size - 1)
        return np.where(self.used[i] == d, self.rank[i], -1)

    def key_of(self, v):
        a = int(self.var_atom[v])
        atom, sh = self.atoms[a], self.shapes[a]
        lin = int(self.var_dense[v] - self.base[a])
        out = []
        for k in reversed(range(len(sh))):
            out.append(atom.lvs[k].instances[lin % sh[k]])
            lin //= sh[k]
        return (atom.name,) + tuple(reversed(out))
