"""Coarse-to-fine lifted particle BP: the schedule of ``HybridLBP.run(c2f >= 0)`` (``HybridLBPLogVersion.py:430-536``).

The reference starts from a coarse partition (continuous evidence merged regardless of value, HLBP:432), splits evidence
clusters by k-means while their variance exceeds a shrinking threshold (HLBP:250-266,475-477), refines the rv clusters once
at the start of every sweep (HLBP:268-290) and the factor clusters once after every proposal update (HLBP:292-308); a new
cluster inherits the messages / sites / proposal / particles of the cluster it was split from.

Between those two refinements the partition is NOT stable, so each sweep sees two different lifted graphs:

* phase 1 (v -> f messages, proposal update) uses the *variable side*: rv clusters P_rv(i) and, per cluster, the factor
  clusters of P_f(i-1) its representative touches with their counts (``SuperRV.update_nb``);
* phase 2 (f -> v messages) uses the *factor side*: factor clusters P_f(i) and the rv cluster at each argument position
  (``SuperF.update_nb``).

``rv_side_graph`` builds the first as a flat graph whose edges are (rv cluster, factor cluster) pairs; the second is the
ordinary flattening of the lifted objects.  State moves between them by gathering rows through pair maps.  The numerical
work is done by an *engine* (the HIP kernels in the product, the CPU oracle in the tests), so this file is pure host logic.
"""
from __future__ import annotations

import numpy as np

from .flat import FlatGraph, flatten
from .lifting import CompressedGraph, initial_colors, split_evidence_colors


def _first_member(colors, n):
    first = np.full(n, -1, dtype=np.int64)
    first[colors[::-1]] = np.arange(colors.size)[::-1]
    return first


def rv_side_graph(gflat, rvc, fc):
    """Flat graph of the variable side of an (unstable) partition: variable = rv cluster, edge = (rv cluster A, factor
    cluster phi) pair with ``count`` = number of the representative's factors that lie in phi, listed in first-seen order
    along the representative's ``rv.nb`` (``SuperRV.update_nb``, CGWO:41-45).  Every pair is its own unary pseudo-factor
    (v2f / proposal kernels never look at the factor side).  Returns (flat, pair_phi [E])."""
    nV = int(rvc.max()) + 1
    rep = _first_member(rvc, nV)
    var_ptr = np.zeros(nV + 1, dtype=np.int32)
    edge_var, pair_phi, counts = [], [], []
    for A in range(nV):
        r = rep[A]
        seen = {}
        for k in range(gflat.var_ptr[r], gflat.var_ptr[r + 1]):
            phi = int(fc[gflat.edge_fac[gflat.var_edge[k]]])
            seen[phi] = seen.get(phi, 0) + 1
        for phi, c in seen.items():
            edge_var.append(A)
            pair_phi.append(phi)
            counts.append(float(c))
        var_ptr[A + 1] = len(edge_var)
    E = len(edge_var)
    val = gflat.var_value
    value = np.full(nV, np.nan)
    obs = ~np.isnan(val)
    for A in np.unique(rvc[obs]):            # SuperRV.get_value: running sum of the members' values / size
        total = 0
        members = np.flatnonzero(rvc == A)
        for m in members:
            total += float(val[m])
        value[A] = total / members.size
    flat = FlatGraph(
        V=nV, F=E, E=E, fac_ptr=np.arange(E + 1, dtype=np.int32), edge_var=np.array(edge_var, dtype=np.int32).reshape(-1),
        edge_fac=np.arange(E, dtype=np.int32), edge_pos=np.zeros(E, dtype=np.int32), edge_canon=np.arange(E, dtype=np.int32),
        var_ptr=var_ptr, var_edge=np.arange(E, dtype=np.int32), edge_count=np.array(counts, dtype=np.float64).reshape(-1),
        lifted=True, fac_pot=np.zeros(E, dtype=np.int32), pot_kind=gflat.pot_kind, pot_off=gflat.pot_off,
        pot_param=gflat.pot_param, var_value=value, var_dom=gflat.var_dom[rep].astype(np.int32),
        var_mult=np.bincount(rvc, minlength=nV).astype(np.float64), fac_mult=np.ones(E),
        dom_cont=gflat.dom_cont, dom_lo=gflat.dom_lo, dom_hi=gflat.dom_hi, dom_ptr=gflat.dom_ptr, dom_val=gflat.dom_val,
        domains=gflat.domains)
    flat.rep_ground = rep
    return flat, np.array(pair_phi, dtype=np.int64)


class Refiner:
    """colour-refinement half rounds on the ground graph; the product uses the device (CompressedGraph), tests the oracle"""

    def factors(self, rvc, fc):
        raise NotImplementedError

    def rvs(self, fc, rvc):
        raise NotImplementedError


class DeviceRefiner(Refiner):
    def __init__(self, g):
        self.cg = CompressedGraph(g)

    def factors(self, rvc, fc):
        self.cg.set_colors(rvc, fc)
        self.cg.split_factors()
        return self.cg.colors()[1].copy()

    def rvs(self, fc, rvc):
        self.cg.set_colors(rvc, fc)
        self.cg.split_rvs()
        return self.cg.colors()[0].copy()


def run_c2f(g, engine, refiner, iteration, c2f, k_mean_k, k_mean_iteration, draw, observer=None):
    """Drive one coarse-to-fine run.  ``engine`` exposes ``make(flat) -> state``, ``init``, ``v2f``, ``proposal``,
    ``f2v``, ``install(state, host_particles)``, ``gather(array, index)``, ``get(state, name)`` / ``set(state, name, array)``
    for the arrays ``f2v v2f eta q particles old_particles uniq`` and ``host(array)``; ``draw(k, flat, q_host)`` returns the
    k-th sample as a [V, n] host array.  ``observer(k, rvc, old_fc, G1, pair_phi, st1)`` (optional) is called right before
    draw k >= 1 with the variable-side state, i.e. what the reference's ``message`` / ``eta_message`` hold at that
    ``generate_sample`` call: the edge of ground rv r and ground factor f is the pair (rvc[r], old_fc[f]).
    Returns (final state, final flat (factor side consistent), final CompressedGraph, rv colours, factor colours,
    history of (rv colours, factor colours) at every draw)."""
    gflat = flatten(g)
    values = gflat.var_value
    rvc, fc = initial_colors(g, is_split_cont_evidence=False)                       # HLBP:432
    rvc = split_evidence_colors(values, rvc, 2, 50, 0.0, use_sqrt=True)             # HLBP:440
    fc = refiner.factors(rvc, fc)
    rvc = refiner.rvs(fc, rvc)
    history = []

    def lift(rvc, fc):
        cg = CompressedGraph(g)
        cg.set_colors(rvc, fc)
        flat = flatten(cg, require_device_potentials=True)
        flat.rep_ground = _first_member(rvc, flat.V)
        return cg, flat

    def evidence_variance(rvc):
        obs = ~np.isnan(values)
        return [float(np.var(values[(rvc == c) & obs])) for c in np.unique(rvc[obs])]

    # ---- sweep 0, phase 1 state
    G1, pair_phi = rv_side_graph(gflat, rvc, fc)
    st1 = engine.make(G1)
    engine.init(st1)
    k = 0
    history.append((rvc.copy(), fc.copy()))
    engine.install(st1, draw(k, G1, engine.host(engine.get(st1, 'q'))))
    k += 1
    var = evidence_variance(rvc)
    epsilon = max(var) if var else 0                                                # HLBP:460-465
    d = (epsilon - c2f) / iteration
    epsilon -= d
    st2 = G2 = cg = None
    for i in range(iteration):
        if i > 0:
            # ---- split_evidence + split_rvs (HLBP:475-485): P_rv(i) from P_rv(i-1), P_f(i-1); inherit from phase 2 of i-1
            old_rvc = rvc
            rvc = split_evidence_colors(values, rvc, k_mean_k, k_mean_iteration, epsilon, use_sqrt=False)
            epsilon = max(epsilon - d, c2f)
            rvc = refiner.rvs(fc, rvc)
            G1, pair_phi = rv_side_graph(gflat, rvc, fc)
            parent = old_rvc[G1.rep_ground]                                        # parent cluster of every new rv cluster
            pair_to_edge2 = {}
            for e in range(G2.E):
                pair_to_edge2.setdefault((int(G2.edge_fac[e]), int(G2.edge_var[e])), int(G2.edge_canon[e]))
            pe = np.array([pair_to_edge2[(int(phi), int(parent[A]))] for A, phi in zip(G1.edge_var, pair_phi)], dtype=np.int64)
            new1 = engine.make(G1)
            for name in ('f2v', 'eta'):
                engine.set(new1, name, engine.gather(engine.get(st2, name), pe))
            for name in ('q', 'particles', 'old_particles', 'uniq'):
                engine.set(new1, name, engine.gather(engine.get(st2, name), parent))
            st1 = new1
        engine.v2f(st1)
        last = i == iteration - 1
        if not last:
            engine.proposal(st1)
        # ---- split_factors (HLBP:509-513; after the loop for the last sweep, HLBP:536): P_f(i) from P_rv(i), P_f(i-1)
        old_fc = fc
        fc = refiner.factors(rvc, fc)
        cg, G2 = lift(rvc, fc)
        parent_f = old_fc[_first_member(fc, G2.F)]
        pair_to_edge1 = {(int(A), int(phi)): e for e, (A, phi) in enumerate(zip(G1.edge_var, pair_phi))}
        pe = np.array([pair_to_edge1[(int(G2.edge_var[e]), int(parent_f[G2.edge_fac[e]]))] for e in range(G2.E)], dtype=np.int64)
        st2 = engine.make(G2)
        for name in ('v2f', 'eta'):
            engine.set(st2, name, engine.gather(engine.get(st1, name), pe))
        for name in ('q', 'particles', 'old_particles', 'uniq'):
            engine.set(st2, name, engine.get(st1, name))
        if not last:
            history.append((rvc.copy(), fc.copy()))
            if observer is not None:
                observer(k, rvc, old_fc, G1, pair_phi, st1)
            engine.install(st2, draw(k, G2, engine.host(engine.get(st2, 'q'))))
            k += 1
            engine.f2v(st2)
    return st2, G2, cg, rvc, fc, history
