"""The reference's three relational model templates and their evidence readers, as builders on this package's API.

``rgm(C, B)``                     the recession / market / loss / revenue Gaussian model (Demo/Data/RGM/Generator.py:8-38)
``paper_popularity(P, T, ...)``   the paper-popularity hybrid MLN (Demo/Data/HMLN/GeneratorPaperPopularity.py:7-47)
``robot_mapping()``               the robot-mapping hybrid MLN (Demo/Data/HMLN/GeneratorRobotMapping.py:7-82): 37 wall
                                  segments, three segment types, two lines; ten weighted formulas up to arity five, ``$W /
                                  $D / $O`` constants, and a continuous domain whose integral points reach beyond it
``load_raw_data(path)``           the Alchemy-style evidence parser (GeneratorRobotMapping.py:85-106)
``load_data(path)``               the JSON evidence reader of all three generators (keys are ``str(tuple)``)
``closed_world(rvs_dict, data, query)``   the demos' closed-world fill (Demo/HMLN/DemoRobotMapping.py:14-23)

Each returns a ``RelationalGraph``: ``ground_graph()`` / ``add_evidence()`` give the object model, ``ground_flat()`` the
array form the GPU solvers take directly.
"""
from __future__ import annotations

import ast
import json
import re

import numpy as np

from .graph import Domain
from .mln import MLNPotential, eq_op, neg_op, or_op
from .potentials import GaussianPotential
from .relational import LV, Atom, ParamF, RelationalGraph


def rgm(C=100, B=10):
    d = Domain((-50, 50), continuous=True, integral_points=np.linspace(-50, 50, 100))
    p1 = GaussianPotential([0., 0.], [[10., -7.], [-7., 10.]])
    p2 = GaussianPotential([0., 0.], [[10., 5.], [5., 10.]])
    p3 = GaussianPotential([0., 0.], [[10., 7.], [7., 10.]])
    lv_r, lv_c, lv_b = LV(('all',)), LV(['c%d' % i for i in range(C)]), LV(['b%d' % i for i in range(B)])
    atoms = (Atom(d, (lv_r,), 'recession'), Atom(d, (lv_b,), 'revenue'), Atom(d, (lv_c, lv_b), 'loss'), Atom(d, (lv_c,), 'market'))
    pfs = (ParamF(p1, nb=('recession($all)', 'market(c)')), ParamF(p2, nb=('market(c)', 'loss(c,b)')),
           ParamF(p3, nb=('loss(c,b)', 'revenue(b)')))
    return RelationalGraph(atoms, pfs)


def paper_popularity(P=300, T=10, points=20):
    db = Domain((0, 1))
    dr = Domain((-15, 15), continuous=True, integral_points=np.linspace(0, 10, points))
    lv_p, lv_t = LV(['p%d' % i for i in range(P)]), LV(['t%d' % i for i in range(T)])
    atoms = (Atom(db, (lv_t, lv_t), 'SameSession'), Atom(db, (lv_p, lv_t), 'PaperIn'),
             Atom(dr, (lv_t,), 'TopicPopularity'), Atom(dr, (lv_p,), 'PaperPopularity'))
    differ = lambda s: s['t1'] != s['t2']
    differ.vectorized = True
    pfs = (ParamF(MLNPotential(lambda x: eq_op(x[0], 1), w=0.3), nb=['PaperPopularity(p)']),
           ParamF(MLNPotential(lambda x: x[0] * eq_op(x[1], x[2]), w=0.5),
                  nb=['SameSession(t1,t2)', 'TopicPopularity(t1)', 'TopicPopularity(t2)'], constrain=differ),
           ParamF(MLNPotential(lambda x: x[0] * eq_op(x[1], x[2]), w=1),
                  nb=['PaperIn(p,t)', 'PaperPopularity(p)', 'TopicPopularity(t)']))
    return RelationalGraph(atoms, pfs)


def robot_mapping(segments=37, lines=2):
    seg = ['A1_%d' % i for i in range(1, segments + 1)]
    line = ['LA%d' % i for i in range(1, lines + 1)]
    db = Domain((0, 1))
    d_length = Domain((0, 1), continuous=True, integral_points=np.linspace(0, 1, 20))
    d_depth = Domain((0, 0.5), continuous=True, integral_points=np.linspace(0, 1, 20))     # (sic: points up to 1 on [0, 0.5])
    lv_seg, lv_type, lv_line = LV(seg), LV(['W', 'D', 'O']), LV(line)
    atoms = (Atom(db, (lv_seg, lv_line), 'PartOf'), Atom(db, (lv_seg, lv_type), 'SegType'), Atom(db, (lv_seg, lv_seg), 'Aligned'),
             Atom(d_length, (lv_seg,), 'Length'), Atom(d_depth, (lv_seg,), 'Depth'))
    t_differ = lambda s: s['t1'] != s['t2']
    s_differ = lambda s: s['s1'] != s['s2']
    t_differ.vectorized = s_differ.vectorized = True
    pfs = (
        ParamF(MLNPotential(lambda x: or_op(neg_op(x[0]), neg_op(x[1])), w=3), nb=['SegType(s,t1)', 'SegType(s,t2)'], constrain=t_differ),
        ParamF(MLNPotential(lambda x: 1 - (x[0] == 0) * (x[1] == 0) * (x[2] == 0), w=3),
               nb=['SegType(s,$W)', 'SegType(s,$D)', 'SegType(s,$O)']),
        ParamF(MLNPotential(lambda x: 1 - (x[0] == 1) * (x[1] == 1) * (x[2] == 0) * (x[3] == 1) * (1 - x[4]), w=1.591),
               nb=['SegType(s1,$W)', 'SegType(s2,$W)', 'PartOf(s1,l)', 'PartOf(s2,l)', 'Aligned(s2,s1)'], constrain=s_differ),
        ParamF(MLNPotential(lambda x: x[0], w=0.3), nb=['SegType(s,$W)']),
        ParamF(MLNPotential(lambda x: x[0], w=-0.737), nb=['SegType(s,$D)']),
        ParamF(MLNPotential(lambda x: x[0], w=-0.077), nb=['SegType(s,$O)']),
        ParamF(MLNPotential(lambda x: x[0] * eq_op(x[1], 0.1), w=3.228), nb=['SegType(s,$D)', 'Length(s)']),
        ParamF(MLNPotential(lambda x: x[0] * eq_op(x[1], 0.02), w=2.668), nb=['SegType(s,$D)', 'Depth(s)']),
        ParamF(MLNPotential(lambda x: x[0] * eq_op(x[1], 0.341), w=3.754), nb=['SegType(s,$W)', 'Length(s)']),
        ParamF(MLNPotential(lambda x: x[0] * eq_op(x[1], 0.001), w=2.532), nb=['SegType(s,$W)', 'Depth(s)']),
    )
    return RelationalGraph(atoms, pfs)


def load_raw_data(path):
    """``Atom(arg, ...)`` lines, optionally followed by a number (the atom's value; 1 otherwise); ``/* ... */`` blocks are
    skipped -- the line that opens and the line that closes a comment are dropped whole, as in the reference"""
    data = dict()
    in_comment = False
    with open(path, 'r') as fh:
        for text in fh:
            if re.search(r'/\*', text):
                in_comment = True
            elif re.search(r'\*/', text):
                in_comment = False
            elif not in_comment:
                parts = re.findall(r'[\w.]+', text)
                if not parts:
                    continue
                if re.search(r'\s\d', text):
                    data[tuple(parts[:-1])] = float(parts[-1])
                else:
                    data[tuple(parts)] = 1
    return data


def load_data(path):
    """JSON evidence whose keys are ``str(tuple)`` (the generators' ``load_data``; the reference ``eval``s the keys)"""
    with open(path, 'r') as fh:
        return {ast.literal_eval(k): v for k, v in json.load(fh).items()}


def closed_world(rvs_dict, data, query=()):
    """every discrete atom that is neither observed nor queried is false; ``query``: atom names or keys to leave open"""
    names = {q for q in query if isinstance(q, str)}
    keys = {q for q in query if not isinstance(q, str)}
    out = dict(data)
    for key, rv in rvs_dict.items():
        if key not in out and key not in keys and key[0] not in names and not rv.domain.continuous:
            out[key] = 0
    return out
