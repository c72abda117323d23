"""Ground Gaussian MRF of a linear dynamical system (API of the reference's ``KalmanFilter.py:7-104``).

``KalmanFilter(domain, A, q, C, r).grounded_graph(T, data)`` -> ``(Graph, rv table [t][i])``.  The transition density
exp(-|x_{t+1} - A' x_t|^2 / 2q) is expanded into pairwise ``XYPotential`` and unary ``X2Potential`` factors, the
observation density into one ``LinearGaussianPotential`` per observed entry; ``data[i, t] == 5000`` marks "missing".
Model construction only (host side).
"""
from __future__ import annotations

import numpy as np

from .graph import F, Graph, RV
from .potentials import LinearGaussianPotential, X2Potential, XYPotential

MISSING = 5000


class KalmanFilter:
    def __init__(self, domain, transition_coeff, transition_variance, observation_coeff, observation_variance):
        self.domain = domain
        self.transition_coeff = transition_coeff
        self.transition_variance = transition_variance
        self.observation_coeff = observation_coeff
        self.observation_variance = observation_variance

    def grounded_graph(self, num_t_steps, data):
        A, q = self.transition_coeff, self.transition_variance
        n = A.shape[0]
        rvs, factors = [], []
        table = [[] for _ in range(num_t_steps)]
        # state variables; t = 0 is observed directly, later steps get an observation node + factor when data exists
        for t in range(num_t_steps):
            for i in range(n):
                if t == 0:
                    rv = RV(self.domain, data[i, 0])
                    rvs.append(rv)
                    table[t].append(rv)
                    continue
                rv = RV(self.domain, None)
                rvs.append(rv)
                table[t].append(rv)
                if data[i, t] != MISSING:
                    obs = RV(self.domain, data[i, t])
                    rvs.append(obs)
                    factors.append(F(LinearGaussianPotential(self.observation_coeff[i, i], self.observation_variance), [rv, obs]))
        # |x' - A^T x|^2 = x'^2 - 2 sum_ij A_ij x_i x'_j + sum_ij (A A^T)_ij x_i x_j
        gram = np.zeros((n, n))
        for j in range(n):
            gram += np.outer(A[:, j], A[:, j])
        for t in range(num_t_steps - 1):
            for i in range(n):
                if t > 0 and gram[i, i] != 0:
                    factors.append(F(X2Potential(gram[i, i], q), [table[t][i]]))
                for j in range(n):
                    if A[i, j] != 0:
                        factors.append(F(XYPotential(-2 * A[i, j], q), [table[t][i], table[t + 1][j]]))
                    if t > 0 and i < j and gram[i, j] != 0:
                        factors.append(F(XYPotential(2 * gram[i, j], q), [table[t][i], table[t][j]]))
        for t in range(1, num_t_steps):
            for i in range(n):
                factors.append(F(X2Potential(1, q), [table[t][i]]))
        g = Graph()
        g.rvs, g.factors = rvs, factors
        g.init_nb()
        return g, table
