"""diagnostic: how much of the heavy f->v kernel's time is memory stall?  Time the launch as it is, then with every descriptor
replaced by the first one (all loads hit the caches, all stores go to one row: compute only) -- same number of edges, same terms."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'lifted-hybrid-variational-inference_amd')]
import numpy as np, torch
from lhvi import _abi, synth, dist
from lhvi.pbp import EPBP

E = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
flat = synth.hybrid_mrf_flat(V=E // 4, deg=4, seed=0)
bp = EPBP(None, n=int(sys.argv[2]) if len(sys.argv) > 2 else 64, proposal_approximation='simple', sampler='device', seed=1)
bp._setup(None, flat=flat)
run = dist.SingleRunner(bp)
run.init()
for _ in range(2):
    run.sweep()
l, st, g, p = _abi.lib(), _abi.stream_ptr(), bp.dg.g, bp.dg.p


def timed(name):
    s = bp._struct(); s.flags |= _abi.PBP_SKIP_GENERIC | _abi.PBP_SKIP_LIGHT | _abi.PBP_SKIP_CQ
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for a, b in ev:
        a.record(); _abi.check(l.lhvi_pbp_f2v(g, p, s, _abi.ptr(bp.v2f), _abi.ptr(bp.f2v), st)); b.record()
    torch.cuda.synchronize()
    print('%-34s %8.3f ms' % (name, float(np.median([a.elapsed_time(b) for a, b in ev]))), flush=True)


timed('heavy kernel, as it is')
keep = bp.heavy_desc.clone()
f2v_keep = bp.f2v.clone()
for k in (1, 64, 4096):
    bp.heavy_desc.copy_(keep[:k].repeat((keep.shape[0] + k - 1) // k, 1)[:keep.shape[0]])
    timed('every descriptor one of the first %d' % k)
bp.heavy_desc.copy_(keep)
bp.f2v.copy_(f2v_keep)
timed('heavy kernel, as it is (again)')
