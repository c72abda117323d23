"""First-order templates that ground to a factor graph (host side).

API of the reference's ``RelationalGraph.py:7-132``: ``LV``, ``Atom``, ``ParamF``, ``RelationalGraph(atoms,
parametric_factors)`` with ``ground_graph()`` and ``add_evidence(data)``.  Atom expressions are strings such as
``'PartOf(s,l)'`` or ``'SegType(s,$W)'`` (``$`` marks a constant).  Grounding is model construction, not the sweep:
it runs on the host and produces ordinary ``Graph`` objects (SURVEY.md section 8(f) row 1 is the flat-array fast path).
"""
from __future__ import annotations

import re
from itertools import product

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
