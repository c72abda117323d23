"""Evaluation helpers kept for API parity with the reference's ``utils.py`` (host-side, not on the hot path)."""
from math import log

import numpy as np


def log_likelihood(g, assignment):
    """negative log of the unnormalised joint at `assignment` (``utils.py:6-15``)"""
    from .flat import ground_order
    res = 0
    for f in ground_order(g.factors):        # (a set in the reference: creation order here, so the sum does not change from process to process)
        value = f.potential.get([assignment[rv] for rv in f.nb])
        if value == 0:
            return -np.inf
        res += log(value)
    return -res


def KL(q, p, domain):
    """``utils.KL`` (utils.py:18-31): sum over the domain's integral points (or states) of qx log(qx / px) with
    qx = q(x) w + 1e-8, px = p(x) w + 1e-8, w = the grid spacing (1 for a discrete domain)"""
    if domain.continuous:
        pts = domain.integral_points
        w = (domain.values[1] - domain.values[0]) / (len(pts) - 1)
    else:
        pts, w = domain.values, 1
    total = 0
    for x in pts:
        qx, px = q(x) * w + 1e-8, p(x) * w + 1e-8
        total += qx * np.log(qx / px)
    return total


def kl_discrete(p, q):
    """KL(p || q) of two probability tables of the same shape (utils.py:34-44)"""
    return np.sum(p * (np.log(p) - np.log(q)))


def _quad(integrand, a, b, args, kwargs):
    from scipy.integrate import quad
    res = quad(integrand, a, b, *args, **kwargs)
    return res if kwargs.get('full_result') else res[0]


def kl_continuous_no_add_const(p, q, a, b, *args, **kwargs):
    """KL(p || q) on [a, b] by adaptive quadrature, integrand log(p^p) - log(q^p) (utils.py:47-67)"""
    def integrand(x):
        px, qx = p(x), q(x)
        return log(px ** px) - log(qx ** px)
    return _quad(integrand, a, b, args, kwargs)


def kl_continuous(p, q, a, b, *args, **kwargs):
    """KL(p || q) on [a, b] by adaptive quadrature; both densities are offset by 1e-100 (utils.py:70-89)"""
    def integrand(x):
        px, qx = p(x) + 1e-100, q(x) + 1e-100
        return px * (log(px) - log(qx))
    return _quad(integrand, a, b, args, kwargs)


def kl_continuous_logpdf(log_p, log_q, a, b, *args, **kwargs):
    """KL(p || q) on [a, b] from the log densities (utils.py:92-113)"""
    from math import exp

    def integrand(x):
        lp = log_p(x)
        return exp(lp) * (lp - log_q(x))
    return _quad(integrand, a, b, args, kwargs)


def kl_normal(mu1, mu2, sig1, sig2):
    """KL(N(mu1, sig1^2) || N(mu2, sig2^2)) -- argument order of the reference (utils.py:116-128)"""
    return log(sig2) - log(sig1) + (sig1 ** 2 + (mu1 - mu2) ** 2) / (2 * sig2 ** 2) - 0.5


def kl_tables(p, q, a, b):
    """Batched ``kl_continuous`` on tabulated densities: ``p``, ``q`` are (V, m) device
    tensors (or arrays) of the two densities of every variable on its own uniform m-point grid over [a[v], b[v]] -- e.g.
    two solvers' ``belief_all`` on the same query grid -- and the result is the (V,) tensor of trapezoid sums of
    (p + 1e-100) (log(p + 1e-100) - log(q + 1e-100)), the integrand of ``kl_continuous``."""
    from . import _abi
    torch = _abi.require_gpu()

    def dev(t):
        return t if torch.is_tensor(t) else _abi.to_dev(np.ascontiguousarray(t, dtype=np.float64))
    p, q = dev(p), dev(q)
    a = dev(np.broadcast_to(np.asarray(a, dtype=np.float64), (p.shape[0],)).copy()) if not torch.is_tensor(a) else a
    b = dev(np.broadcast_to(np.asarray(b, dtype=np.float64), (p.shape[0],)).copy()) if not torch.is_tensor(b) else b
    pp, qq = p + 1e-100, q + 1e-100
    f = pp * (torch.log(pp) - torch.log(qq))
    h = (b - a) / (p.shape[1] - 1)
    return (f[:, 1:] + f[:, :-1]).sum(dim=1) * h * 0.5


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
