"""Ground Gaussian MRF of a linear dynamical system (API of the reference's ``KalmanFilter.py:7-104``).

``KalmanFilter(domain, A, q, C, r).grounded_graph(T, data)`` -> ``(Graph, rv table [t][i])``.  The transition density
exp(-|x_{t+1} - A' x_t|^2 / 2q) is expanded into pairwise ``XYPotential`` and unary ``X2Potential`` factors, the
observation density into one ``LinearGaussianPotential`` per observed entry; ``data[i, t] == 5000`` marks "missing".
Model construction only (host side).  ``grounded_flat`` builds the same graph straight into a ``FlatGraph`` (no per-node
objects): same variable order, same factor order, potentials deduplicated by value.
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

    def grounded_flat(self, num_t_steps, data):
        """The graph of ``grounded_graph`` as a ``FlatGraph`` plus ``state_id[t, i]`` (variable index of state i at step t).
        One pass of NumPy index arithmetic per factor family; the per-step factor pattern is built once and tiled."""
        from .flat import build_flat
        from .potentials import POT_LINEAR_GAUSSIAN, POT_X2, POT_XY
        A, q = np.asarray(self.transition_coeff, dtype=np.float64), float(self.transition_variance)
        Cc, r = np.asarray(self.observation_coeff, dtype=np.float64), float(self.observation_variance)
        data = np.asarray(data, dtype=np.float64)
        n, T = A.shape[0], int(num_t_steps)
        has_obs = np.zeros((T, n), dtype=bool)
        has_obs[1:] = (data[:, 1:T] != MISSING).T
        # variable ids in creation order: state (t, i), immediately followed by its observation node when there is one
        before = np.concatenate([[0], np.cumsum(has_obs.ravel())[:-1]]).reshape(T, n)
        state_id = (np.arange(T * n).reshape(T, n) + before).astype(np.int64)
        V = T * n + int(has_obs.sum())
        value = np.full(V, np.nan)
        value[state_id[0]] = data[:, 0]
        tt, ii = np.nonzero(has_obs)
        value[state_id[tt, ii] + 1] = data[ii, tt]
        specs, spec_id = [], {}

        def pot(kind, a, b):
            key = (kind, float(a), float(b))
            if key not in spec_id:
                spec_id[key] = len(specs)
                specs.append((kind, [float(a), float(b)]))
            return spec_id[key]

        scopes, pots = [], []           # per family: (F, arity) int64 scope arrays / (F,) potential ids
        # observation factors, (t, i) order
        scopes.append(np.stack([state_id[tt, ii], state_id[tt, ii] + 1], axis=1))
        pots.append(np.array([pot(POT_LINEAR_GAUSSIAN, Cc[i, i], r) for i in range(n)], dtype=np.int64)[ii] if n else np.zeros(0, np.int64))
        # transition factors: pattern of one step as (potential, (dt, i) of each argument), second argument -1 = unary
        gram = np.zeros((n, n))
        for j in range(n):
            gram += np.outer(A[:, j], A[:, j])

        def pattern(first_step):
            rows = []
            for i in range(n):
                if not first_step and gram[i, i] != 0:
                    rows.append((pot(POT_X2, gram[i, i], q), 0, i, -1, -1))
                for j in range(n):
                    if A[i, j] != 0:
                        rows.append((pot(POT_XY, -2 * A[i, j], q), 0, i, 1, j))
                    if not first_step and i < j and gram[i, j] != 0:
                        rows.append((pot(POT_XY, 2 * gram[i, j], q), 0, i, 0, j))
            return np.array(rows, dtype=np.int64).reshape(-1, 5)

        step_scopes, step_pots, step_arity = [], [], []
        for first, steps in ((True, np.arange(0, min(1, T - 1))), (False, np.arange(1, max(T - 1, 1)))):
            pat = pattern(first)
            if steps.size == 0 or pat.shape[0] == 0:
                continue
            t = np.repeat(steps, pat.shape[0])
            pr = np.tile(pat, (steps.size, 1))
            a0 = state_id[t + pr[:, 1], pr[:, 2]]
            unary = pr[:, 3] < 0
            a1 = np.where(unary, -1, state_id[np.minimum(t + np.maximum(pr[:, 3], 0), T - 1), np.maximum(pr[:, 4], 0)])
            step_scopes.append(np.stack([a0, a1], axis=1)); step_pots.append(pr[:, 0]); step_arity.append(np.where(unary, 1, 2))
        # unary x'^2 terms, t >= 1
        x2 = pot(POT_X2, 1, q)
        tail = state_id[1:].ravel()
        edge_parts, arity_parts, pot_parts = [scopes[0].ravel()], [np.full(scopes[0].shape[0], 2)], [pots[0]]
        for sc, pp, ar in zip(step_scopes, step_pots, step_arity):
            edge_parts.append(sc.ravel()[sc.ravel() >= 0]); arity_parts.append(ar); pot_parts.append(pp)
        edge_parts.append(tail); arity_parts.append(np.ones(tail.size, dtype=np.int64)); pot_parts.append(np.full(tail.size, x2))
        arity = np.concatenate(arity_parts)
        fac_ptr = np.concatenate([[0], np.cumsum(arity)]).astype(np.int32)
        flat = build_flat(fac_ptr, np.concatenate(edge_parts).astype(np.int32), np.concatenate(pot_parts).astype(np.int32), specs, value,
                          np.zeros(V, dtype=np.int32), [self.domain])
        return flat, state_id
