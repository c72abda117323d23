"""diagnostic: time of the device sampler's listed form on the headline graph"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'lifted-hybrid-variational-inference_amd')]
import numpy as np, torch
from lhvi import _abi, synth, dist
from lhvi.pbp import EPBP
flat = synth.hybrid_mrf_flat(V=2_500_000, deg=4, seed=0)
bp = EPBP(None, n=64, proposal_approximation='simple', sampler='device', seed=1)
bp._setup(None, flat=flat)
run = dist.SingleRunner(bp)
run.init()
run.sweep()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(8)]
for a, b in ev:
    a.record(); bp._generate_sample(); b.record()
torch.cuda.synchronize()
print('resample (listed): %.3f ms' % float(np.median([a.elapsed_time(b) for a, b in ev])), os.environ.get('LHVI_LIB', 'default'))
