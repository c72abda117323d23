"""diagnostic: where do non-finite log messages first appear on the paper-popularity HMLN (device sampler)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'lifted-hybrid-variational-inference_amd')]
import numpy as np, torch
from lhvi import _abi, synth
from lhvi.pbp import EPBP

P, T, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
flat, keys = synth.paper_popularity_flat(P, T, seed=0)
for routed in (True, False):
    bp = EPBP(None, n=n, proposal_approximation='simple', sampler='device', seed=1)
    bp.cq_routing = routed
    bp._setup(None, flat=flat)
    _abi.check(_abi.lib().lhvi_pbp_init(bp.dg.g, bp._struct(), _abi.ptr(bp.eta), _abi.ptr(bp.q_dev), _abi.ptr(bp.f2v), _abi.ptr(bp.v2f), _abi.stream_ptr()))
    bp._generate_sample()
    cls = torch.zeros(flat.E, dtype=torch.uint8, device='cuda')
    _abi.check(_abi.lib().lhvi_pbp_classify(bp.dg.g, bp.dg.p, bp._struct(), _abi.ptr(cls), _abi.stream_ptr()))
    cls = cls.cpu().numpy()
    arity = np.diff(flat.fac_ptr)[flat.edge_fac]
    for it in range(4):
        bp.sweep()
        f2v, v2f, q = bp.f2v.cpu().numpy(), bp.v2f.cpu().numpy(), bp.q_dev.cpu().numpy()
        badf = ~np.isfinite(f2v).all(axis=1)
        badv = ~np.isfinite(v2f).all(axis=1)
        print('routed', routed, 'sweep', it, 'bad f2v rows', int(badf.sum()), 'by class', np.bincount(cls[badf], minlength=5).tolist(),
              'bad v2f rows', int(badv.sum()), 'bad q', int((~np.isfinite(q)).any(axis=1).sum()),
              'v2f max', float(np.nanmax(v2f)), 'f2v max', float(np.nanmax(np.where(np.isfinite(f2v), f2v, -1e300))), flush=True)
        if it == 3 and routed:
            keep = (f2v, q)
    if not routed:
        ok = np.isfinite(keep[0]) & np.isfinite(f2v)
        print('max |f2v routed - generic| over finite entries', float(np.abs(keep[0] - f2v)[ok].max()), 'q', float(np.nanmax(np.abs(keep[1] - q))))
