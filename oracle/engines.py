"""Test infrastructure: ``lhvi.c2f`` engines and refiners backed by the CPU oracle (``oracle.PbpOracle`` states, the exact CPU
colour refinement).  Used by ``tests/`` as the checker of the coarse-to-fine schedule and by the labelled CPU-baseline legs of the
measurement scripts; never by the product."""
import numpy as np

from . import oracle


class OracleEngine:
    """lhvi.c2f engine backed by the CPU oracle (PbpOracle states)"""

    def __init__(self, n, ep):
        self.n, self.ep = n, ep

    def make(self, flat, sides='vf'):
        return oracle.PbpOracle(flat, self.n, ep=self.ep, epbp=False, var_threshold=5)

    get = staticmethod(getattr)
    set = staticmethod(lambda st, name, value: setattr(st, name, np.ascontiguousarray(value)))
    host = staticmethod(lambda a: a)
    gather = staticmethod(lambda a, idx: np.ascontiguousarray(a[np.asarray(idx, dtype=np.int64)]))
    init = staticmethod(lambda st: st.init())
    v2f = staticmethod(lambda st: st.step_v2f())
    proposal = staticmethod(lambda st: st.step_proposal())
    f2v = staticmethod(lambda st: st.step_f2v())
    install = staticmethod(lambda st, p: st.set_particles(p))


class OracleTensorRefiner:
    """the half rounds of ``lhvi.c2f.run_c2f_flat`` on CPU tensors, backed by the exact CPU colour refinement"""

    def __init__(self, gflat, sym):
        self.gflat, self.sym = gflat, np.asarray(sym)

    def factors(self, rvc, fc):
        import torch
        return torch.from_numpy(np.asarray(oracle.refine_factors(self.gflat, self.sym, rvc.numpy(), fc.numpy())[0], dtype=np.int32))

    def rvs(self, fc, rvc):
        import torch
        return torch.from_numpy(np.asarray(oracle.refine_rvs(self.gflat, fc.numpy(), rvc.numpy())[0], dtype=np.int32))
