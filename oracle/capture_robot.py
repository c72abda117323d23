"""Golden vectors for the robot-mapping hybrid MLN (the reference's second HMLN demo), captured from the reference.

TEST INFRASTRUCTURE, build container only (see capture_golden.py).  The model is the reference's own: template from
Demo/Data/HMLN/GeneratorRobotMapping.py:11-82 (``generate_rel_graph``: arity-5 boolean formula, ``$W/$D/$O`` constants,
or_op / neg_op, a continuous domain whose integral points lie outside it), raw evidence parsed by its ``load_raw_data``
(:85-106) from Demo/Data/HMLN/robot-map, closed-world fill and query set of Demo/HMLN/DemoRobotMapping.py:11-27.  Only data
is written: the grounding (tests/golden/grounding.json.gz, key ``robot_mapping``), and solver runs as ``.npz`` fixtures in the
formats of capture_vi.py / capture_pbp.py.  usage: python oracle/capture_golden.py robot
"""
import gzip
import importlib.util
import json
import os

import numpy as np

# names under which tests/modelio.py knows the generator's formulas (same expressions, restated there)
FORMULA_OF_FACTOR = ['nand', 'any3', 'rm_aligned', 'x0', 'x0', 'x0', 'x0_eq_0.1', 'x0_eq_0.02', 'x0_eq1c', 'x0_eq_0.001']


def load_generator(cg):
    spec = importlib.util.spec_from_file_location('ref_robot_generator', os.path.join(cg.REF, 'Demo/Data/HMLN/GeneratorRobotMapping.py'))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    return gen


def model_robot(cg, return_parts=False):
    """the demo's graph, with deterministic iteration order for the fixture (rvs in rvs_dict order, factors by template
    and scope)"""
    gen = load_generator(cg)
    rel_g = gen.generate_rel_graph()
    g, rvs_dict = rel_g.ground_graph()
    query = {key for key in rvs_dict if key[0] in ('SegType', 'PartOf', 'Length', 'Depth')}
    data = gen.load_raw_data(os.path.join(cg.REF, 'Demo/Data/HMLN/robot-map'))
    for key, rv in rvs_dict.items():
        if key not in data and key not in query and not rv.domain.continuous:
            data[key] = 0                                   # closed world assumption (DemoRobotMapping.py:21-23)
    g, rvs_dict = rel_g.add_evidence(data)
    rvs = list(rvs_dict.values())
    idx = {id(rv): i for i, rv in enumerate(rvs)}
    tmpl = {id(pf.potential): i for i, pf in enumerate(rel_g.param_factors)}
    g.rvs = rvs
    g.factors = sorted(g.factors, key=lambda f: (tmpl[id(f.potential)], [idx[id(r)] for r in f.nb]))
    g.init_nb()
    for i, pf in enumerate(rel_g.param_factors):
        cg.modelio._FORMULA_NAME[id(pf.potential.formula)] = FORMULA_OF_FACTOR[i]
        # the restated lambda must agree with the generator's on every joint state / a few real values
        mine = cg.modelio.FORMULAS[FORMULA_OF_FACTOR[i]]
        for x in np.ndindex(*([2] * len(pf.nb))):
            for r in (0.0, 0.37):
                xs = [float(v) for v in x]
                if pf.nb[-1].startswith(('Length', 'Depth')):
                    xs[-1] = r
                assert abs(mine(xs) - pf.potential.formula(xs)) < 1e-15, (i, xs)
    if return_parts:
        return g, rel_g, rvs_dict, data
    return g


def capture_grounding(cg):
    gen = load_generator(cg)
    rel_g = gen.generate_rel_graph()
    g, rvs_dict = rel_g.ground_graph()
    key_of = {id(rv): list(k) for k, rv in rvs_dict.items()}
    pf_of = {id(pf.potential): i for i, pf in enumerate(rel_g.param_factors)}
    entry = {'rvs': sorted(key_of.values()),
             'factors': sorted([pf_of[id(f.potential)], [key_of[id(rv)] for rv in f.nb]] for f in g.factors)}
    # evidence as the demo assembles it (raw data + closed world), for the keys that name a ground atom
    _, _, rvs_dict, data = model_robot(cg, True)
    entry['evidence'] = sorted([list(k), float(v)] for k, v in data.items() if k in rvs_dict)
    entry['raw_keys_without_atom'] = sorted(list(k) for k in data if k not in rvs_dict)
    path = os.path.join(cg.ROOT, 'tests', 'golden', 'grounding.json.gz')
    with gzip.open(path, 'rt') as fh:
        out = json.load(fh)
    out['robot_mapping'] = entry
    with gzip.open(path, 'wt') as fh:
        json.dump(out, fh, separators=(',', ':'))
    print('wrote', path, os.path.getsize(path), 'bytes; robot_mapping:', len(entry['rvs']), 'rvs', len(entry['factors']), 'factors',
          len(entry['evidence']), 'evidence atoms', len(entry['raw_keys_without_atom']), 'raw keys naming no atom')


def capture_robot(cg, what=('grounding', 'lvi', 'c2fvi', 'epbp', 'hlbp')):
    from capture_pbp import capture_epbp, capture_hlbp
    from capture_vi import capture_c2fvi, capture_one
    if 'grounding' in what:
        capture_grounding(cg)
    if 'lvi' in what:
        capture_one(cg, 'lifted_robot_k2', model_robot(cg), True, 2, 3, 51, 3, lr=0.2)
    if 'c2fvi' in what:
        capture_c2fvi(cg, 'c2f_robot_k2', model_robot(cg), 2, 3, 52, 30, 0.2)
    if 'epbp' in what:
        capture_epbp(cg, 'epbp_robot', model_robot(cg), 10, 3, 'simple', 53)
    if 'hlbp' in what:
        capture_hlbp(cg, 'hlbp_robot', model_robot(cg), 10, 3, 'simple', 54)
