"""Evaluation helpers kept for API parity with the reference's ``utils.py`` (host-side, not on the hot path)."""
from math import log

import numpy as np


def log_likelihood(g, assignment):
    """negative log of the unnormalised joint at `assignment` (``utils.py:6-15``)"""
    res = 0
    for f in g.factors:
        value = f.potential.get([assignment[rv] for rv in f.nb])
        if value == 0:
            return -np.inf
        res += log(value)
    return -res


def kl_normal(p_mu, p_sig, q_mu, q_sig):
    """KL(N(p_mu, p_sig^2) || N(q_mu, q_sig^2))"""
    return np.log(q_sig / p_sig) + (p_sig ** 2 + (p_mu - q_mu) ** 2) / (2 * q_sig ** 2) - 0.5


def log_likelihood_flat(dg, x):
    """``log_likelihood`` of the assignment ``x`` (array of length V, variable-index order) on a device-resident graph
    (``_abi.DeviceGraph``): one thread per factor, two-stage reduction (``lhvi_log_likelihood``)."""
    from . import _abi
    torch = _abi.require_gpu()
    l = _abi.lib()
    xd = x if torch.is_tensor(x) else _abi.to_dev(np.ascontiguousarray(x, dtype=np.float64))
    nbytes = int(l.lhvi_log_likelihood_workspace_bytes(dg.g))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dg.device)
    out = torch.empty(1, dtype=torch.float64, device=dg.device)
    _abi.check(l.lhvi_log_likelihood(dg.g, dg.p, _abi.ptr(xd), _abi.ptr(out), _abi.ptr(ws), nbytes, _abi.stream_ptr()))
    return float(out.item())
