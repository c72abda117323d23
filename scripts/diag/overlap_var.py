"""does the per-variable half of a sweep (v -> f, proposal update, new particles) run beside the heavy f -> v kernel?  (diagnosis aid for
a two-phase pipeline of consecutive sweeps; the data dependencies are ignored here -- only the clock matters)"""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, 'lifted-hybrid-variational-inference_amd')]
import torch
from lhvi import _abi, synth
from lhvi.pbp import EPBP
E = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
flat = synth.hybrid_mrf_flat(V=E // 4, deg=4, seed=0, T=32)
bp = EPBP(None, n=64, proposal_approximation='simple', sampler='device', seed=1)
bp._setup(None, flat=flat)
l, st, g, p = _abi.lib(), _abi.stream_ptr(), bp.dg.g, bp.dg.p
_abi.check(l.lhvi_pbp_init(g, bp._struct(), _abi.ptr(bp.eta), _abi.ptr(bp.q_dev), _abi.ptr(bp.f2v), _abi.ptr(bp.v2f), st))
bp._generate_sample()
for _ in range(3):
    bp.sweep(last=False)
torch.cuda.synchronize()
main = torch.cuda.current_stream()
side = torch.cuda.Stream()
v2f2 = torch.zeros_like(bp.v2f)
p3 = torch.zeros_like(bp.particles)
gid = _abi.ptr(getattr(bp, 'var_gid', None))

def var_half(stream_ptr, frac=1.0):
    s = bp._struct()
    if frac < 1.0:              # a prefix of every per-variable list
        for name in ('v2f_wide', 'v2f_narrow', 'prop_desc'):
            setattr(s, 'n_' + name, int(getattr(s, 'n_' + name) * frac))
        s.resample_vars, s.n_resample_vars = _abi.ptr(bp.resample_vars), int(bp.resample_vars.shape[0] * frac)
    else:
        s.resample_vars, s.n_resample_vars = _abi.ptr(bp.resample_vars), int(bp.resample_vars.shape[0])
    _abi.check(l.lhvi_pbp_v2f(g, s, _abi.ptr(bp.f2v), _abi.ptr(v2f2), stream_ptr))
    _abi.check(l.lhvi_pbp_proposal(g, s, _abi.ptr(bp.f2v), _abi.ptr(bp.eta), _abi.ptr(bp.q_dev), stream_ptr))
    _abi.check(l.lhvi_pbp_resample_uniq(g, s, gid, 1, 7, _abi.ptr(p3), _abi.ptr(bp.uniq), stream_ptr))

def heavy(stream_ptr, share, frac=1.0, off=0.0):
    s = bp._struct()
    s.flags |= _abi.PBP_SKIP_GENERIC | _abi.PBP_SKIP_LIGHT | (_abi.PBP_SHARE_CUS if share else 0)
    n0 = int(bp.n_heavy * off)
    s.heavy_desc, s.n_heavy = bp.heavy_desc.data_ptr() + n0 * 128, int(bp.n_heavy * frac)
    _abi.check(l.lhvi_pbp_f2v(g, p, s, _abi.ptr(bp.v2f), _abi.ptr(bp.f2v), stream_ptr))

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

print('var half alone (all variables) %.3f ms' % timed(lambda: var_half(st)))
print('var half alone (half of the variables) %.3f ms' % timed(lambda: var_half(st, 0.5)))
print('heavy alone, full list %.3f ms; half list %.3f ms; half list sharing CUs %.3f ms' % (timed(lambda: heavy(st, False)), timed(lambda: heavy(st, False, 0.5)), timed(lambda: heavy(st, True, 0.5))))

def both(frac_var, share):
    ev = torch.cuda.Event(); ev.record(main)
    side.wait_event(ev)
    with torch.cuda.stream(side):
        var_half(_abi.stream_ptr(), frac_var)
        done = torch.cuda.Event(); done.record(side)
    heavy(st, share, 0.5, 0.5)
    main.wait_event(done)
for share in (True, False):
    print('half heavy list beside half of the variables (share CUs %s): %.3f ms' % (share, timed(lambda: both(0.5, share))))
