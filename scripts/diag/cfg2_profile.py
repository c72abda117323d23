"""host profile of the reference-size cfg 2 call: GaLBP(g).run(20) on the RGM template (C = 100, B = 50) through the objects"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'lifted-hybrid-variational-inference_amd')]
import numpy as np, torch
from lhvi import generators
from lhvi.gabp import GaLBP
rel = generators.rgm(100, 50)
rel.ground_graph()
rng = np.random.default_rng(0)
keys = list(rel.rvs_dict)
data = {k: float(rng.uniform(-30, 30)) for k in keys if rng.random() < 0.2}
g, _ = rel.add_evidence(data)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    b = GaLBP(g); b.run(20)
    torch.cuda.synchronize(); print('call', rep, round(1e3 * (time.perf_counter() - t0), 1), 'ms')
pr = cProfile.Profile(); pr.enable()
b = GaLBP(g); b.run(20)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
