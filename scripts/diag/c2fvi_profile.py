"""where a re-lift of run_c2fvi_flat spends its time on the 10 M-edge RGM"""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'lifted-hybrid-variational-inference_amd')]
import numpy as np, torch
from lhvi import _abi, synth, c2fvi
flat, sym, rv0, f0 = synth.rgm_structured_flat()
dg = _abi.DeviceGraph(flat)
owner = c2fvi.VarInference.__new__(c2fvi.VarInference)
owner._init_common(2, 3)
opts = dict(k_mean_k=2, k_mean_its=10, update_obs_its=10, output_its=0, min_obs_var=0, gaussian_obs=True)
np.random.seed(0)
c2fvi.run_c2fvi_flat(flat, c2fvi._DeviceEngine(owner), 2, 30, 0.2, opts, dg=dg)
pr = cProfile.Profile()
pr.enable()
np.random.seed(0)
res = c2fvi.run_c2fvi_flat(flat, c2fvi._DeviceEngine(owner), 2, 30, 0.2, opts, dg=dg)
pr.disable()
print('relift', res['relift_s'])
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
