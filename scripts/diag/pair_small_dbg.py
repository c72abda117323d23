"""where the packed pair kernel and the one-entry-per-wave kernel differ after ONE f -> v half sweep on the same state"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'lifted-hybrid-variational-inference_amd')]
import numpy as np, torch
from lhvi import _abi, synth
from lhvi.pbp import EPBP
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
flat = synth.hybrid_mrf_flat(V=6001, deg=4, seed=17, frac_discrete=0.4, evidence_ratio=0.25)
outs = []
for wide in (False, True):
    bp = EPBP(None, n=n, proposal_approximation='simple', sampler='device', seed=3)
    bp._setup(None, flat=flat)
    if wide:
        bp.flags |= _abi.PBP_WIDE_PAIRS


    bp._run_sweeps(2)
    torch.cuda.synchronize()
    outs.append((bp.f2v.cpu().numpy().copy(), bp))
a, b = outs[0][0], outs[1][0]
bp = outs[0][1]
diff = np.argwhere(a != b)
print('entries that differ', len(diff), 'of', a.size)
rows = np.unique(diff[:, 0])
ev = flat.edge_var[rows]
print('rows', len(rows), 'continuous targets', int(flat.var_cont[ev].sum()), 'discrete targets', int((~flat.var_cont[ev]).sum()))
for r, c in diff[:12]:
    print(r, c, a[r, c], b[r, c], 'target cont' if flat.var_cont[flat.edge_var[r]] else 'target disc', 'np', bp.np_host[flat.edge_var[r]])
print('columns', np.unique(diff[:, 1])[:40])
