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
