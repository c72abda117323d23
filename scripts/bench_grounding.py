"""Host-side grounding: RelationalGraph.ground_flat (arrays) against ground_graph + flatten (objects), RGM template.
usage: python scripts/bench_grounding.py [out.json]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'lifted-hybrid-variational-inference_amd'), os.path.join(ROOT, 'tests')]
import numpy as np
from lhvi import graph as G, potentials as P, relational as R
from lhvi.flat import flatten


def rgm(C, B):
    d = G.Domain((-50, 50), continuous=True, integral_points=np.linspace(-50, 50, 30))
    p1, p2, p3 = (P.GaussianPotential([0., 0.], s) for s in ([[10., -7.], [-7., 10.]], [[10., 5.], [5., 10.]], [[10., 7.], [7., 10.]]))
    lv_r, lv_c, lv_b = R.LV(('all',)), R.LV([f'c{i}' for i in range(C)]), R.LV([f'b{i}' for i in range(B)])
    atoms = (R.Atom(d, (lv_r,), 'recession'), R.Atom(d, (lv_b,), 'revenue'), R.Atom(d, (lv_c, lv_b), 'loss'), R.Atom(d, (lv_c,), 'market'))
    pfs = (R.ParamF(p1, nb=('recession($all)', 'market(c)')), R.ParamF(p2, nb=('market(c)', 'loss(c,b)')),
           R.ParamF(p3, nb=('loss(c,b)', 'revenue(b)')))
    return R.RelationalGraph(atoms, pfs)


rows = []
for C, B, objects in ((100, 50, True), (400, 250, True), (2000, 1250, False)):
    t0 = time.perf_counter()
    flat, keys = rgm(C, B).ground_flat()
    t_flat = time.perf_counter() - t0
    row = {'C': C, 'B': B, 'factors': int(flat.F), 'edges': int(flat.E), 'ground_flat_s': round(t_flat, 4),
           'flat_edges_per_s': round(flat.E / t_flat)}
    if objects:
        t0 = time.perf_counter()
        g, _ = rgm(C, B).ground_graph()
        fo = flatten(g)
        t_obj = time.perf_counter() - t0
        row.update({'ground_graph_plus_flatten_s': round(t_obj, 3), 'speedup': round(t_obj / t_flat, 1)})
    rows.append(row)
    print(json.dumps(row), flush=True)
if len(sys.argv) > 1:
    json.dump({'host': 'build container, 1 core', 'rows': rows}, open(sys.argv[1], 'w'), indent=1)
