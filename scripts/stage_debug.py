"""bring-up aid: run the particle sweep stage by stage with a synchronize + print after each launch"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'lifted-hybrid-variational-inference_amd')]
import numpy as np, torch
from lhvi import _abi, synth
from lhvi.pbp import EPBP

def say(*a):
    print(*a, flush=True)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
if len(sys.argv) > 2:
    import json
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import modelio
    from test_oracle_golden import API
    from lhvi.flat import flatten
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'pbp_%s.npz' % sys.argv[2]))
    meta = json.loads(str(z['meta']))
    g, rvs, factors = modelio.load_model(meta['model'], API)
    flat = flatten(g, require_device_potentials=True)
    say('model', flat.V, flat.F, flat.E, 'maxdeg', int(np.diff(flat.var_ptr).max()))
else:
    flat = synth.hybrid_mrf_flat(V=2000, deg=4, seed=7)
bp = EPBP(None, n=n, proposal_approximation='simple', sampler='device', seed=5)
say('setup'); bp._setup(None, flat=flat); torch.cuda.synchronize(); say(' ok', bp.fast_edges.numel(), bp.generic_edges.numel())
l, st = _abi.lib(), _abi.stream_ptr()
say('init'); _abi.check(l.lhvi_pbp_init(bp.dg.g, bp._struct(), _abi.ptr(bp.eta), _abi.ptr(bp.q_dev), _abi.ptr(bp.f2v), _abi.ptr(bp.v2f), st)); torch.cuda.synchronize(); say(' ok')
say('resample'); bp.old_particles, bp.particles = bp.particles, bp.old_particles
_abi.check(l.lhvi_pbp_resample(bp.dg.g, bp._struct(), None, 5, 0, _abi.ptr(bp.particles), st)); torch.cuda.synchronize(); say(' ok')
say('uniq'); _abi.check(l.lhvi_pbp_uniq(bp.dg.g, n, _abi.ptr(bp.particles), _abi.ptr(bp.np_dev), _abi.ptr(bp.uniq), st)); torch.cuda.synchronize(); say(' ok')
say('v2f'); _abi.check(l.lhvi_pbp_v2f(bp.dg.g, bp._struct(), _abi.ptr(bp.f2v), _abi.ptr(bp.v2f), st)); torch.cuda.synchronize(); say(' ok')
say('proposal'); _abi.check(l.lhvi_pbp_proposal(bp.dg.g, bp._struct(), _abi.ptr(bp.f2v), _abi.ptr(bp.eta), _abi.ptr(bp.q_dev), st)); torch.cuda.synchronize(); say(' ok')
s = bp._struct(); s.flags |= _abi.PBP_SKIP_GENERIC
say('f2v fast'); _abi.check(l.lhvi_pbp_f2v(bp.dg.g, bp.dg.p, s, _abi.ptr(bp.v2f), _abi.ptr(bp.f2v), st)); torch.cuda.synchronize(); say(' ok')
s = bp._struct(); s.flags |= _abi.PBP_SKIP_FAST
say('f2v generic'); _abi.check(l.lhvi_pbp_f2v(bp.dg.g, bp.dg.p, s, _abi.ptr(bp.v2f), _abi.ptr(bp.f2v), st)); torch.cuda.synchronize(); say(' ok')
say('done')
