"""How many heavy edges of the headline graph take the uniform-grid recurrence (the guard passes), and what the two forms
of the integral-point part cost: lhvi_pbp_f2v timed with and without LHVI_PBP_NO_GRID on the same state.
usage: python scripts/grid_stats.py [edges] [out.json]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'lifted-hybrid-variational-inference_amd')]
import numpy as np, torch
from lhvi import _abi, synth, dist
from lhvi.pbp import EPBP

E = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
flat = synth.hybrid_mrf_flat(V=E // 4, deg=4, seed=0)
bp = EPBP(None, n=64, proposal_approximation='simple', sampler='device', seed=1)
bp._setup(None, flat=flat)
run = dist.SingleRunner(bp)
run.init()
out = {'edges': flat.E, 'heavy_edges': bp.n_heavy}
l, st = _abi.lib(), _abi.stream_ptr()
words = bp.heavy_desc.view(torch.int32).view(-1, 32)
live = (words[:, 7] >= 24) & (words[:, 15] == 1)
out['eligible_edges'] = int(live.sum().item())
for sweep in (1, 5, 20):
    while bp._draws - 1 < sweep:
        run.sweep()
    s = bp._struct()
    s.flags |= _abi.PBP_SKIP_GENERIC | _abi.PBP_SKIP_LIGHT
    ms = {}
    res = {}
    for name, extra in (('grid', 0), ('direct', _abi.PBP_NO_GRID)):
        s.flags = (s.flags & ~_abi.PBP_NO_GRID) | extra
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        _abi.check(l.lhvi_pbp_f2v(bp.dg.g, bp.dg.p, s, _abi.ptr(bp.v2f), _abi.ptr(bp.f2v), st))
        a.record()
        for _ in range(3):
            _abi.check(l.lhvi_pbp_f2v(bp.dg.g, bp.dg.p, s, _abi.ptr(bp.v2f), _abi.ptr(bp.f2v), st))
        b.record(); torch.cuda.synchronize()
        ms[name] = a.elapsed_time(b) / 3
        res[name] = bp.f2v[words[:, 0].long()][:, bp.n:].clone()
    same = (res['grid'] == res['direct']).all(dim=1)
    diff = (res['grid'] - res['direct']).abs().max().item()
    out['after_%d_sweeps' % sweep] = {'heavy_ms_grid': round(ms['grid'], 3), 'heavy_ms_direct': round(ms['direct'], 3),
                                      'eligible_edges_on_direct_form': int((same & live).sum().item()),
                                      'max_abs_difference_of_log_messages': diff}
print(json.dumps(out))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], 'w'), indent=1)
