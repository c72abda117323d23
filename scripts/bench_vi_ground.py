import sys, time, json
sys.path[:0] = ['/root/repo', '/root/repo/lifted-hybrid-variational-inference_amd']
import numpy as np, torch
from lhvi import _abi, synth
from lhvi.vi import VarInference
C, B = int(sys.argv[1]), int(sys.argv[2])
flat, sym, rv0, f0 = synth.rgm_flat(C=C, B=B, n_values=0, evidence_ratio=0.1, seed=0)
vi = VarInference(None, 2, 3)
vi._setup_flat(flat)
np.random.seed(0)
vi.init_param()
vi._grad(); torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
for a, b in ev:
    a.record(); vi._grad(); b.record()
torch.cuda.synchronize()
t = float(np.median([a.elapsed_time(b) for a, b in ev]))
print(json.dumps({'config': 'ground VI gradient K=2 T=3, RGM', 'factors': int(flat.F), 'edges': int(flat.E), 'grad_ms': t,
                  'factors_per_s': flat.F / (t * 1e-3)}))
