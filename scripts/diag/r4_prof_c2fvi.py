"""host profile (cProfile) of a coarse-to-fine variational run on a reference-size model"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'lifted-hybrid-variational-inference_amd')]
import numpy as np, torch
from lhvi import synth, c2fvi
which = sys.argv[1] if len(sys.argv) > 1 else 'pp'
flat = synth.paper_popularity_flat(300, 10, seed=0, points=20)[0] if which == 'pp' else synth.rgm_flat(C=100, B=5, n_values=0, evidence_ratio=0.2, seed=0)[0]
K = 2 if which == 'pp' else 1
owner = c2fvi.VarInference.__new__(c2fvi.VarInference)
owner._init_common(K, 3)
opts = dict(k_mean_k=2, k_mean_its=10, update_obs_its=10, output_its=0, min_obs_var=0, gaussian_obs=True)
def run():
    np.random.seed(0)
    return c2fvi.run_c2fvi_flat(flat, c2fvi._DeviceEngine(owner), K, 100, 0.2, opts)
for rep in range(2):
    t0 = time.perf_counter(); res = run(); torch.cuda.synchronize(); print('wall', time.perf_counter() - t0, [round(1e3 * x, 2) for x in res['relift_s']])
pr = cProfile.Profile()
pr.enable(); run(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
pstats.Stats(pr).sort_stats('tottime').print_stats(30)
