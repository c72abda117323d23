"""per-kernel timing of the particle sweep with HIP events (tuning aid; bench.py is the contract benchmark)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'lifted-hybrid-variational-inference_amd')]
import numpy as np, torch
from lhvi import _abi, synth, dist
from lhvi.pbp import EPBP

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
reps = 5
fd = float(os.environ.get('FRAC_DISC', 0.2)); ev = float(os.environ.get('EVID', 0.1)); T = int(os.environ.get('GRID', 32))
flat = synth.hybrid_mrf_flat(V=E // 4, deg=4, seed=0, frac_discrete=fd, evidence_ratio=ev, T=T)
bp = EPBP(None, n=n, proposal_approximation='simple', sampler='device', seed=1)
bp._setup(None, flat=flat)
run = dist.SingleRunner(bp)
run.init()
for _ in range(2):
    run.sweep()
l, st, g, p = _abi.lib(), _abi.stream_ptr(), bp.dg.g, bp.dg.p


def timed(name, fn):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = np.median([a.elapsed_time(b) for a, b in ev])
    print('%-14s %8.3f ms   (x%.1f for 10M edges: %.2f ms)' % (name, t, 1e7 / flat.E, t * 1e7 / flat.E), flush=True)
    return t


s = bp._struct()
tot = 0
tot += timed('v2f', lambda: _abi.check(l.lhvi_pbp_v2f(g, s, _abi.ptr(bp.f2v), _abi.ptr(bp.v2f), st)))
tot += timed('proposal', lambda: _abi.check(l.lhvi_pbp_proposal(g, s, _abi.ptr(bp.f2v), _abi.ptr(bp.eta), _abi.ptr(bp.q_dev), st)))
tot += timed('resample+uniq', lambda: _abi.check(l.lhvi_pbp_resample_uniq(g, s, None, 1, 3, _abi.ptr(bp.particles), _abi.ptr(bp.uniq), st)))
sf = bp._struct(); sf.flags |= _abi.PBP_SKIP_GENERIC
tot += timed('f2v fast', lambda: _abi.check(l.lhvi_pbp_f2v(g, p, sf, _abi.ptr(bp.v2f), _abi.ptr(bp.f2v), st)))
sh = bp._struct(); sh.flags |= _abi.PBP_SKIP_GENERIC | _abi.PBP_SKIP_LIGHT
timed('  f2v heavy', lambda: _abi.check(l.lhvi_pbp_f2v(g, p, sh, _abi.ptr(bp.v2f), _abi.ptr(bp.f2v), st)))
sl = bp._struct(); sl.flags |= _abi.PBP_SKIP_GENERIC | _abi.PBP_SKIP_HEAVY
timed('  f2v light', lambda: _abi.check(l.lhvi_pbp_f2v(g, p, sl, _abi.ptr(bp.v2f), _abi.ptr(bp.f2v), st)))
sn = bp._struct(); sn.flags |= _abi.PBP_SKIP_GENERIC | _abi.PBP_SKIP_TERMS
timed('f2v fast (no term loop)', lambda: _abi.check(l.lhvi_pbp_f2v(g, p, sn, _abi.ptr(bp.v2f), _abi.ptr(bp.f2v), st)))
sg = bp._struct(); sg.flags |= _abi.PBP_SKIP_FAST
tot += timed('f2v generic', lambda: _abi.check(l.lhvi_pbp_f2v(g, p, sg, _abi.ptr(bp.v2f), _abi.ptr(bp.f2v), st)))
print('joint terms %.4g wave-terms %.4g' % (run.f2v_joint_terms(), run.f2v_joint_terms() / 64))
print('sum %.3f ms -> %.1f sweeps/s at 10M edges' % (tot, 1e3 / (tot * 1e7 / flat.E)))
