"""host profile (cProfile) of one particle coarse-to-fine run on the 10 M-edge RGM"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'lifted-hybrid-variational-inference_amd')]
import numpy as np, torch
from lhvi import synth
from lhvi.pbp import HybridLBP
flat = synth.rgm_structured_flat()[0] if len(sys.argv) < 2 else synth.rgm_flat(C=100, B=10, evidence_ratio=0.07)[0]
for rep in range(2):
    bp = HybridLBP.on_flat(flat, n=10, proposal_approximation='simple', sampler='device', seed=1)
    t0 = time.perf_counter(); bp.run_flat(10, c2f=0); torch.cuda.synchronize(); print('wall', time.perf_counter() - t0)
bp = HybridLBP.on_flat(flat, n=10, proposal_approximation='simple', sampler='device', seed=1)
pr = cProfile.Profile()
pr.enable()
bp.run_flat(10, c2f=0)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(60)
