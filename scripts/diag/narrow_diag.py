"""where do the 10-lane groups differ from the 16-lane groups on the wide domain?  (diagnosis aid; run on the GPU box)"""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, 'lifted-hybrid-variational-inference_amd'), os.path.join(R, 'tests')]
import torch
from lhvi import synth, _abi as api
from lhvi.graph import Domain
from lhvi.pbp import EPBP
from test_gpu_pbp import _with_domain, _init
n = 10
lo, hi = -40.0, 40.0
pts = np.linspace(lo, hi, 32)
flat = _with_domain(synth.hybrid_mrf_flat(V=3001, deg=4, seed=14, frac_discrete=0.1), Domain((lo, hi), continuous=True, integral_points=pts))
bp = EPBP(None, n=n, proposal_approximation='simple', sampler='device', seed=6)
bp._setup(None, flat=flat)
_init(api, bp)
for _ in range(2):
    bp.sweep(last=False)
l, st = api.lib(), api.stream_ptr()
outs = {}
for name, fl in (('narrow', 0), ('pow2', api.PBP_POW2_GROUPS), ('direct', api.PBP_NO_GRID)):
    s = bp._struct()
    s.flags |= fl
    bp.f2v.zero_()
    api.check(l.lhvi_pbp_f2v(bp.dg.g, bp.dg.p, s, api.ptr(bp.v2f), api.ptr(bp.f2v), st))
    outs[name] = bp.f2v.cpu().numpy().copy()
words = bp.small16_desc.view(torch.int32).view(-1, 32).cpu().numpy()
e = words[:, 0]
a, b, c = outs['narrow'][e], outs['pow2'][e], outs['direct'][e]
bad = np.argwhere(~np.isclose(a, b, rtol=1e-12, atol=1e-12))
print(len(bad), 'mismatches; rows', np.unique(bad[:, 0])[:20])
for i, j in bad[:40]:
    print('item', i, 'step6', i // 6, i % 6, 'step4', i // 4, i % 4, 'col', j, 'nj', words[i, 7], 'np', words[i, 8], 'narrow %.12g pow2 %.12g direct %.12g' % (a[i, j], b[i, j], c[i, j]))
