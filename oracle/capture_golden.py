#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by importing the reference implementation.

TEST INFRASTRUCTURE.  Runs only in the build container (needs /root/reference, read-only); the GPU box
never sees the reference.  Nothing under the product package imports this file.

The reference is pure Python; it imports under Python 3.10 / NumPy 2 once three aliases exist
(SURVEY.md section 8(c)): numpy.Inf, collections.MutableSet, time.clock.  Reference files are not
modified or copied: models are built with the reference's own classes, its solvers are run, and only
*data* (serialised models, messages, marginals, partitions) is written out.

usage: python oracle/capture_golden.py [gauss] [color] [pbp] [vi] [c2fvi] [c2fvi_loglik] [robot | robot:grounding,lvi,c2fvi,epbp,hlbp]
"""
import collections
import collections.abc
import contextlib
import io
import json
import os
import sys
import time

import numpy as np

REF = '/root/reference'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'tests', 'golden')

np.Inf = np.inf
collections.MutableSet = collections.abc.MutableSet
time.clock = time.perf_counter
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import modelio  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import Graph as RG  # noqa: E402  (reference modules)
import Potential as RP  # noqa: E402
import MLNPotential as RM  # noqa: E402


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()):
        yield


def save(name, obj):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + '.json')
    with open(path, 'w') as fh:
        json.dump(obj, fh)
    print('wrote', path, os.path.getsize(path), 'bytes')


def nan_pair(m):
    if m is None:
        return [float('nan'), float('nan')]
    return [float(m[0]), float('nan') if m[1] is None else float(m[1])]


def jsonable(x):
    """NaN/Inf-safe nested lists (json.dump writes NaN/Infinity literals, which json.load reads back)."""
    return np.asarray(x, dtype=float).tolist()


# ---------------------------------------------------------------------------------------------
# models
# ---------------------------------------------------------------------------------------------
def model_chain():
    """cfg 1 / fixture G1: 10-node Gaussian chain, pairwise LinearGaussian(0.9,1) + unary X2(1,4), rv0 observed"""
    d = RG.Domain((-10, 10), continuous=True, integral_points=np.linspace(-10, 10, 32))
    rvs = [RG.RV(d, 1.5 if i == 0 else None) for i in range(10)]
    lin = RP.LinearGaussianPotential(0.9, 1.0)
    x2 = RP.X2Potential(1.0, 4.0)
    fs = [RG.F(lin, [rvs[i], rvs[i + 1]]) for i in range(9)]
    fs += [RG.F(x2, [rvs[i]]) for i in range(1, 10)]
    g = RG.Graph()
    g.rvs, g.factors = rvs, fs
    g.init_nb()
    return g


def model_kalman(n=4, T=5, seed=0, missing=True):
    """fixture G2: dense-transition Kalman graph from the reference's own KalmanFilter builder"""
    import KalmanFilter as RK
    rng = np.random.RandomState(seed)
    A = rng.uniform(-0.5, 0.5, size=(n, n)) + np.eye(n) * 0.5
    data = rng.uniform(-2, 2, size=(n, T))
    if missing:
        data[rng.rand(n, T) < 0.3] = 5000
        data[:, 0] = rng.uniform(-2, 2, size=n)
    d = RG.Domain((-8, 8), continuous=True, integral_points=np.linspace(-8, 8, 32))
    kf = RK.KalmanFilter(d, A, 1.5, np.eye(n), 0.7)
    g, _ = kf.grounded_graph(T, data)
    return g


def model_rgm(idx=0):
    """fixture G3: the reference's RGM template + its shipped JSON evidence Demo/Data/RGM/<idx>"""
    import importlib.util
    spec = importlib.util.spec_from_file_location('rgm_generator', os.path.join(REF, 'Demo/Data/RGM/Generator.py'))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    rel_g = gen.generate_rel_graph()
    rel_g.ground_graph()
    data = gen.load_data(os.path.join(REF, 'Demo/Data/RGM/%d' % idx))
    g, _ = rel_g.add_evidence(data)
    g.rvs = sorted(g.rvs)
    g.factors = sorted(g.factors)
    g.init_nb()
    return g


def model_color_demo():
    """Demo/old/ColorPassingDemo.py graph: 3 boolean rvs in a chain with one symmetric table potential"""
    p1 = RP.TablePotential({(True, True): 4, (True, False): 1, (False, True): 1, (False, False): 3}, symmetric=True)
    d = RG.Domain([True, False])
    rvs = [RG.RV(d) for _ in range(3)]
    fs = [RG.F(p1, (rvs[i], rvs[i + 1])) for i in range(2)]
    g = RG.Graph()
    g.rvs, g.factors = rvs, fs
    g.init_nb()
    return g


def model_symmetric_ring():
    """ring of 8 continuous rvs: XY (symmetric) on even links, LinearGaussian (ordered) on odd links,
    X2 priors, two equal and one distinct evidence values: exercises sorted vs ordered signatures"""
    d = RG.Domain((-5, 5), continuous=True, integral_points=np.linspace(-5, 5, 16))
    vals = {0: 0.5, 4: 0.5, 6: -1.0}
    rvs = [RG.RV(d, vals.get(i)) for i in range(8)]
    xy = RP.XYPotential(0.4, 2.0)
    lin = RP.LinearGaussianPotential(0.7, 1.3)
    lin2 = RP.LinearGaussianPotential(0.7, 1.3)      # equal by value -> same colour as lin
    x2 = RP.X2Potential(1.0, 3.0)
    fs = []
    for i in range(8):
        j = (i + 1) % 8
        if i % 2 == 0:
            fs.append(RG.F(xy, [rvs[i], rvs[j]] if i % 4 == 0 else [rvs[j], rvs[i]]))
        else:
            fs.append(RG.F(lin if i % 4 == 1 else lin2, [rvs[i], rvs[j]]))
    fs += [RG.F(x2, [rvs[i]]) for i in range(8) if i not in vals]
    g = RG.Graph()
    g.rvs, g.factors = rvs, fs
    g.init_nb()
    return g


# ---------------------------------------------------------------------------------------------
# helpers shared by the captures
# ---------------------------------------------------------------------------------------------
def edge_list(g):
    """factor-major (factor, position) incidences; the layout lhvi/flat.py uses"""
    return [(f, pos, rv) for f in g.factors for pos, rv in enumerate(f.nb)]


def partition_labels(items, clusters, member_attr):
    """canonical labelling: every ground item gets the smallest ground index in its cluster"""
    index = {id(x): i for i, x in enumerate(items)}
    lab = [-1] * len(items)
    for c in clusters:
        members = [index[id(x)] for x in getattr(c, member_attr)]
        m = min(members)
        for i in members:
            lab[i] = m
    return lab


def capture_gauss():
    import GaBP as RGaBP
    import GaLBP as RGaLBP
    for name, builder in (('g1_chain', model_chain), ('g2_kalman', model_kalman), ('g3_rgm0', model_rgm)):
        g = builder()
        rec = {'model': modelio.dump_model(g), 'sweeps': {}}
        edges = edge_list(g)
        its = (1, 2, 5, 20) if name != 'g3_rgm0' else (2, 10)
        for k in its:
            bp = RGaBP.GaBP(g)
            with quiet():
                bp.run(k)
            rec['sweeps'][str(k)] = {
                'f2v': [nan_pair(bp.message[(f, rv)]) if rv.value is None else [float('nan')] * 2 for f, _, rv in edges],
                'v2f': [nan_pair(bp.message[(rv, f)]) for f, _, rv in edges],
                'mu_var': [list(map(float, bp.get_belief_params(rv))) if rv.value is None else [float(rv.value), 0.0]
                           for rv in g.rvs],
            }
        lbp = RGaLBP.GaLBP(g)
        with quiet():
            lbp.run(its[-1])
        rec['galbp'] = {
            'iterations': its[-1],
            'rv_label': partition_labels(g.rvs, lbp.g.rvs, 'rvs'),
            'f_label': partition_labels(g.factors, lbp.g.factors, 'factors'),
            'map': [float(lbp.map(rv)) for rv in g.rvs],
        }
        save('gauss_' + name, rec)


def capture_color():
    import CompressedGraphWithObs as CGWO
    import CompressedGraphSorted as CGS
    rec = {}
    for name, builder in (('color_demo', model_color_demo), ('sym_ring', model_symmetric_ring),
                          ('kalman', model_kalman), ('kalman_full', lambda: model_kalman(3, 6, 1, False)),
                          ('chain', model_chain)):
        g = builder()
        cg = CGWO.CompressedGraph(g)
        cg.run()
        entry = {
            'model': modelio.dump_model(g),
            'rv_label': partition_labels(g.rvs, cg.rvs, 'rvs'),
            'f_label': partition_labels(g.factors, cg.factors, 'factors'),
            'n_rv': len(cg.rvs), 'n_f': len(cg.factors),
        }
        # per-cluster N and count multiset (keyed by canonical labels)
        flab = {id(c): min(g.factors.index(f) for f in c.factors) for c in cg.factors}
        rlab = {id(c): min(g.rvs.index(r) for r in c.rvs) for c in cg.rvs}
        entry['counts'] = {str(rlab[id(c)]): sorted([flab[id(f)], int(n)] for f, n in c.count.items()) for c in cg.rvs}
        entry['N'] = {str(rlab[id(c)]): int(c.N) for c in cg.rvs}
        entry['value'] = {str(rlab[id(c)]): (None if c.value is None else float(c.value)) for c in cg.rvs}
        # coarse initial clustering used by c2f (continuous evidence merged regardless of value)
        cg2 = CGWO.CompressedGraph(g)
        cg2.init_cluster(False)
        entry['coarse_init_rv_label'] = partition_labels(g.rvs, cg2.rvs, 'rvs')
        if all(rv.value is None for rv in g.rvs):
            cs = CGS.CompressedGraphSorted(g)
            cs.run()
            entry['sorted_rv_label'] = partition_labels(g.rvs, cs.rvs, 'rvs')
            entry['sorted_f_label'] = partition_labels(g.factors, cs.factors, 'factors')
        rec[name] = entry
    save('color_partitions', rec)


if __name__ == '__main__':
    what = sys.argv[1:] or ['gauss', 'color', 'pbp', 'vi']
    os.chdir(REF)
    if 'gauss' in what:
        capture_gauss()
    if 'color' in what:
        capture_color()
    if 'pbp' in what:
        from capture_pbp import capture_pbp
        capture_pbp(sys.modules[__name__])
    if 'vi' in what:
        from capture_vi import capture_vi
        capture_vi(sys.modules[__name__])
    if 'c2fvi' in what:
        from capture_vi import capture_c2f
        capture_c2f(sys.modules[__name__])
    if 'c2fvi_loglik' in what:
        from capture_vi import capture_c2f_loglik
        capture_c2f_loglik(sys.modules[__name__])
    for w in what:
        if w.split(':')[0] == 'robot':
            from capture_robot import capture_robot
            capture_robot(sys.modules[__name__], *([tuple(w.split(':')[1].split(','))] if ':' in w else []))
