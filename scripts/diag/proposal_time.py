"""diagnostic: time of lhvi_pbp_proposal (listed form) on the headline graph"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'lifted-hybrid-variational-inference_amd')]
import numpy as np, torch
from lhvi import _abi, synth, dist
from lhvi.pbp import EPBP
flat = synth.hybrid_mrf_flat(V=2_500_000, deg=4, seed=0)
for mode in ('simple', 'EP'):
    bp = EPBP(None, n=64, proposal_approximation=mode, sampler='device', seed=1)
    bp._setup(None, flat=flat)
    run = dist.SingleRunner(bp)
    run.init()
    run.sweep(); run.sweep()
    l, st = _abi.lib(), _abi.stream_ptr()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(8)]
    for a, b in ev:
        a.record(); _abi.check(l.lhvi_pbp_proposal(bp.dg.g, bp._struct(), _abi.ptr(bp.f2v), _abi.ptr(bp.eta), _abi.ptr(bp.q_dev), st)); b.record()
    torch.cuda.synchronize()
    print('proposal %s: %.3f ms' % (mode, float(np.median([a.elapsed_time(b) for a, b in ev]))), os.environ.get('LHVI_LIB', 'default'), flush=True)
    del bp, run
    torch.cuda.empty_cache()
