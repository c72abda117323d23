import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'lifted-hybrid-variational-inference_amd')]
import numpy as np, torch
from lhvi import generators
from lhvi.pbp import EPBP, HybridLBP
rel = generators.rgm(100, 10); rel.ground_graph()
g, table = rel.add_evidence({('recession', 'all'): 25.0})
for cls in (EPBP, HybridLBP):
    bp = cls(g, n=10, proposal_approximation='simple', sampler='device', seed=1)
    bp.run(5)
    rv = [r for r in g.rvs if r.value is None][5]
    bp.belief(1.0, rv)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(500):
        bp.belief(0.01 * i, rv)
    torch.cuda.synchronize(); print(cls.__name__, 'belief(x, rv) per call: %.3f ms' % ((time.perf_counter() - t0) / 500 * 1e3))
