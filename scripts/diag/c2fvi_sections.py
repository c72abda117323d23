"""diagnostic: seconds per section of a re-lift round of run_c2fvi_flat on the 10 M-edge RGM (synchronised timers)"""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'lifted-hybrid-variational-inference_amd')]
import numpy as np, torch
from lhvi import _abi, synth, c2fvi, lifting
acc = collections.defaultdict(list)


def timed(mod, name):
    fn = getattr(mod, name)

    def wrap(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); acc[name].append(time.perf_counter() - t0)
        return r
    setattr(mod, name, wrap)


timed(c2fvi, 'split_evidence_observed'); timed(c2fvi, 'cp_run_device'); timed(lifting, 'lift_flat'); timed(lifting, 'refine_flat')
flat, sym, rv0, f0 = synth.rgm_structured_flat()
dg = _abi.DeviceGraph(flat)
owner = c2fvi.VarInference.__new__(c2fvi.VarInference)
owner._init_common(2, 3)
opts = dict(k_mean_k=2, k_mean_its=10, update_obs_its=10, output_its=0, min_obs_var=0, gaussian_obs=True)
for rep in range(2):
    acc.clear()
    np.random.seed(0)
    res = c2fvi.run_c2fvi_flat(flat, c2fvi._DeviceEngine(owner), 2, 30, 0.2, opts, dg=dg)
print('relift ms', [round(1e3 * x, 2) for x in res['relift_s']])
for k, v in acc.items():
    print('%-26s' % k, [round(1e3 * x, 2) for x in v])
