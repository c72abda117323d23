"""soak: many sweeps of the headline configuration, twice; everything must stay finite and the two runs bit-identical"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'lifted-hybrid-variational-inference_amd')]
import numpy as np, torch
from lhvi import synth, dist
from lhvi.pbp import EPBP
E = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 100
flat = synth.hybrid_mrf_flat(V=E // 4, deg=4, seed=0)
res = []
for rep in range(2):
    bp = EPBP(None, n=64, proposal_approximation='simple', sampler='device', seed=1)
    bp._setup(None, flat=flat)
    run = dist.SingleRunner(bp)
    run.init()
    t0 = time.perf_counter()
    for i in range(S):
        run.sweep()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    q = bp.q_dev.cpu().numpy()
    f2v = bp.f2v
    hid = flat.var_hidden & flat.var_cont
    ok = bool(torch.isfinite(f2v).all().item()) and bool(np.isfinite(q[hid]).all()) and bool((q[hid][:, 1] > 0).all())
    print('run %d: %d sweeps in %.2f s (%.1f sweeps/s), finite=%s, q var range [%.3g, %.3g]' % (rep, S, dt, S / dt, ok, q[hid][:, 1].min(), q[hid][:, 1].max()), flush=True)
    res.append((q.copy(), bp.particles.cpu().numpy().copy()))
print('bit-identical runs:', bool((res[0][0] == res[1][0]).all() and (res[0][1] == res[1][1]).all()))
