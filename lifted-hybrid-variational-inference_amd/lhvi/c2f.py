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

from . import _abi
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
    """Drive one coarse-to-fine run.  ``engine`` exposes ``make(flat, sides) -> state`` (`sides`: 'v' for a state that only runs the v -> f half and the proposal update), ``init``, ``v2f``, ``proposal``,
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

    def refined(old, new):
        """a refinement only splits: the same number of colours means the same partition -- the old ids stay then (the
        refinement numbers its colours afresh every time; keeping them lets the array path reuse a sweep's graphs and states)"""
        return old if int(new.max()) == int(old.max()) else new
    fc = refined(fc, refiner.factors(rvc, fc))
    rvc = refined(rvc, refiner.rvs(fc, rvc))
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
    st1 = engine.make(G1, sides='v')
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
            rvc = refined(rvc, refiner.rvs(fc, rvc))
            G1, pair_phi = rv_side_graph(gflat, rvc, fc)
            parent = old_rvc[G1.rep_ground]                                        # parent cluster of every new rv cluster
            pair_to_edge2 = {}
            for e in range(G2.E):
                pair_to_edge2.setdefault((int(G2.edge_fac[e]), int(G2.edge_var[e])), int(G2.edge_canon[e]))
            pe = np.array([pair_to_edge2[(int(phi), int(parent[A]))] for A, phi in zip(G1.edge_var, pair_phi)], dtype=np.int64)
            new1 = engine.make(G1, sides='v')
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
        fc = refined(fc, refiner.factors(rvc, fc))
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


# ---- the same schedule on arrays (ground FlatGraph in, colours resident on the device) ------------------------------------------
def _rep_of(colors_t, n, size):
    """first (smallest-index) member of every colour, as a tensor on the colours' device"""
    from .lifting import first_members
    return first_members(colors_t, n, size)


def rv_side_graph_t(gflat, tg, rvc_t, fc_t):
    """``rv_side_graph`` from colour TENSORS (on the device in the product): the rows of the representatives are walked with
    tensor operations there and only lifted-size arrays reach the host.  `tg`: the ground arrays as tensors (``DeviceGraph`` or
    ``lifting.TensorGraph``).  Same numbering as the host function: edges ordered by (cluster, first appearance of the factor
    colour along the representative's row).  Returns (flat, pair_phi [E], rep [V] as a tensor)."""
    import torch
    from .lifting import segment_sums
    dev = rvc_t.device
    rl, fl = rvc_t.long(), fc_t.long()
    nV, nF = int(rl.max().item()) + 1, int(fl.max().item()) + 1
    rep = _rep_of(rvc_t, nV, gflat.V)
    var_ptr_g = tg.t['var_ptr'].long()
    deg = var_ptr_g[rep + 1] - var_ptr_g[rep]
    start = torch.zeros(nV + 1, dtype=torch.int64, device=dev)
    torch.cumsum(deg, 0, out=start[1:])
    total = int(start[-1].item())
    slots = torch.repeat_interleave(var_ptr_g[rep] - start[:-1], deg, output_size=total) + torch.arange(total, device=dev)
    fcol = fl[tg.t['edge_fac'].long()[tg.t['var_edge'].long()[slots]]]
    owner = torch.repeat_interleave(torch.arange(nV, device=dev), deg, output_size=total)
    uniq, inv, cnt = torch.unique(owner * nF + fcol, return_inverse=True, return_counts=True)
    first = _rep_of(inv, int(uniq.numel()), total)
    order = torch.sort(first, stable=True).indices                   # rows are contiguous per cluster: (cluster, first seen)
    uniq, cnt = uniq[order], cnt[order]
    pc, pf = (uniq // nF), (uniq % nF)
    # evidence value of a cluster: running sum of the members' values / size (SuperRV.get_value), members in ground order
    value = tg.t['var_value']
    val = value[rep].clone()
    mult = torch.bincount(rl, minlength=nV).to(torch.float64)
    obs_members = torch.nonzero(~torch.isnan(value)).flatten()
    if obs_members.numel():
        oc = rl[obs_members]
        o2 = torch.sort(oc, stable=True).indices
        sums = segment_sums(value[obs_members][o2], torch.bincount(oc, minlength=nV))
        ob = ~torch.isnan(val)
        val[ob] = sums[ob] / mult[ob]
    host = lambda t: t.cpu().numpy()
    edge_var = host(pc).astype(np.int32)
    E = int(edge_var.size)
    var_ptr = np.zeros(nV + 1, dtype=np.int32)
    np.cumsum(np.bincount(edge_var, minlength=nV), out=var_ptr[1:])
    rep_h = host(rep)
    flat = FlatGraph(
        V=nV, F=E, E=E, fac_ptr=np.arange(E + 1, dtype=np.int32), edge_var=edge_var,
        edge_fac=np.arange(E, dtype=np.int32), edge_pos=np.zeros(E, dtype=np.int32), edge_canon=np.arange(E, dtype=np.int32),
        var_ptr=var_ptr, var_edge=np.arange(E, dtype=np.int32), edge_count=host(cnt).astype(np.float64),
        lifted=True, fac_pot=np.zeros(E, dtype=np.int32), pot_kind=gflat.pot_kind, pot_off=gflat.pot_off,
        pot_param=gflat.pot_param, var_value=host(val), var_dom=gflat.var_dom[rep_h].astype(np.int32),
        var_mult=host(mult), fac_mult=np.ones(E),
        dom_cont=gflat.dom_cont, dom_lo=gflat.dom_lo, dom_hi=gflat.dom_hi, dom_ptr=gflat.dom_ptr, dom_val=gflat.dom_val,
        domains=gflat.domains)
    flat.rep_ground = rep_h
    return flat, host(pf).astype(np.int64), rep


def _lookup(keys_sorted, order, wanted, what):
    pos = np.searchsorted(keys_sorted, wanted)
    if wanted.size and ((pos >= keys_sorted.size).any() or (keys_sorted[np.minimum(pos, keys_sorted.size - 1)] != wanted).any()):
        raise KeyError('coarse-to-fine inheritance: a %s pair has no parent edge' % what)
    return order[pos]


def edges_from_factor_side(G1, pair_phi, parent, G2):
    """for every edge (A, phi) of the variable-side graph G1: the canonical edge of the previous factor-side graph G2 that
    carries (factor cluster phi, rv cluster parent[A]) -- the first such edge in edge order (HLBP:268-290: a new rv cluster
    inherits the messages of the cluster it was split from)"""
    key2 = G2.edge_fac.astype(np.int64) * G2.V + G2.edge_var
    order = np.argsort(key2, kind='stable')
    ks = key2[order]
    first = np.ones(ks.size, dtype=bool)
    first[1:] = ks[1:] != ks[:-1]
    e = _lookup(ks[first], order[first], pair_phi * G2.V + parent[G1.edge_var], '(factor cluster, rv cluster)')
    return G2.edge_canon[e].astype(np.int64)


def edges_from_variable_side(G2, parent_f, G1, pair_phi, n_phi):
    """for every edge of the factor-side graph G2: the edge of the variable-side graph G1 that carries (its rv cluster, the
    parent of its factor cluster) (HLBP:292-308: a new factor cluster inherits from the one it was split from)"""
    key1 = G1.edge_var.astype(np.int64) * n_phi + pair_phi
    order = np.argsort(key1, kind='stable')
    return _lookup(key1[order], order, G2.edge_var.astype(np.int64) * n_phi + parent_f[G2.edge_fac], '(rv cluster, factor cluster)').astype(np.int64)


def split_evidence_observed(ovals, oc, nc, k, iteration, epsilon, use_sqrt):
    """``lifting.split_evidence_colors`` on the OBSERVED members only (`ovals` their values and `oc` their colours, in ground
    order; `nc` colours in all): same pieces, same numbering (clusters in ascending colour, piece 0 keeps the colour, the others
    are numbered from `nc` up).  Returns (new oc, new nc)."""
    from .c2fvi import _kmeans_vec
    oc = np.asarray(oc, dtype=np.int64)
    n = np.bincount(oc, minlength=nc).astype(np.float64)
    with np.errstate(invalid='ignore', divide='ignore'):
        mean = np.bincount(oc, weights=ovals, minlength=nc) / n
        var = np.bincount(oc, weights=(ovals - mean[oc]) ** 2, minlength=nc) / n
    spread = np.sqrt(np.nan_to_num(var)) if use_sqrt else np.nan_to_num(var)
    todo = (n > 1) & (spread > epsilon)
    if not todo.any():
        return oc, nc
    sel = np.flatnonzero(todo[oc])
    order = sel[np.argsort(oc[sel], kind='stable')]
    cols, start = np.unique(oc[order], return_index=True)
    bounds = np.append(start, order.size)
    out = oc.copy()
    for gi in range(cols.size):
        loc = order[bounds[gi]:bounds[gi + 1]]
        assign = _kmeans_vec(ovals[loc], k, iteration)
        if assign is None:
            continue
        for piece in range(1, int(assign.max()) + 1):
            part = loc[assign == piece]
            if part.size:
                out[part] = nc
                nc += 1
    return out, nc


def colour_mean_var(oc_t, vals_t, nc):
    """(count, mean, variance) per colour of the observed members' values, every sum a running sum over the colour's members in
    ground order -- ``numpy.bincount``'s order in the objects path -- whatever the device (a weighted ``torch.bincount`` adds with
    atomics on the GPU: its last bit, and with it a ``spread > epsilon`` decision at a tie, would change from run to run)"""
    import torch
    from .lifting import segment_sums
    cnt = torch.bincount(oc_t, minlength=nc)
    order = torch.sort(oc_t, stable=True).indices
    n = cnt.to(torch.float64)
    mean = segment_sums(vals_t[order], cnt) / n
    var = segment_sums(((vals_t - mean[oc_t]) ** 2)[order], cnt) / n
    return n, mean, var


def split_evidence_tensors(ovals_t, vcode_t, distinct_vals, oc_t, nc, k, iteration, epsilon, use_sqrt):
    """``split_evidence_observed`` with the grouping done by tensor operations on the colours' device: per-colour counts, means and
    variances (``colour_mean_var``: sequential sums in ground order), then -- for the colours whose spread exceeds `epsilon` -- the distinct (colour, value)
    pairs with multiplicities and first positions.  Only those pairs reach the host (a few per colour), where the k-means of
    ``SuperRV.split_by_evidence`` runs on them; the pieces' colours go back as a small table.  `ovals_t` / `vcode_t`: values of the
    observed members and their dense codes (``distinct_vals[code]``, host) in ground order; `oc_t`: their colours (int64).
    Returns (new colours of the observed members, new number of colours); same pieces and numbering as the host function."""
    import torch
    from .c2fvi import _kmeans_distinct
    from .lifting import first_members
    dev = oc_t.device
    n, mean, var = colour_mean_var(oc_t, ovals_t, nc)
    spread = torch.sqrt(torch.nan_to_num(var)) if use_sqrt else torch.nan_to_num(var)
    todo = (n > 1) & (spread > epsilon)
    sel = torch.nonzero(todo[oc_t]).flatten()
    if sel.numel() == 0:
        return oc_t, nc
    nvals = int(distinct_vals.size)
    key, inv, cnt = torch.unique(oc_t[sel] * nvals + vcode_t[sel], return_inverse=True, return_counts=True)
    first = first_members(inv, int(key.numel()), int(sel.numel()))
    key_h, cnt_h, first_h = key.cpu().numpy(), cnt.cpu().numpy(), first.cpu().numpy()
    col_h, code_h = key_h // nvals, key_h % nvals
    table = col_h.copy()                                   # new colour of every (colour, value) pair
    bounds = np.flatnonzero(np.concatenate([[True], col_h[1:] != col_h[:-1], [True]]))      # key is sorted: pairs grouped by colour
    for a, b in zip(bounds[:-1], bounds[1:]):
        order = np.argsort(first_h[a:b], kind='stable')   # distinct values in first-appearance (member) order
        res = _kmeans_distinct(distinct_vals[code_h[a:b]][order], cnt_h[a:b][order], k, iteration)
        if res is None:
            continue
        assign = np.empty(b - a, dtype=np.int64)
        assign[order] = res[0]
        for piece in range(1, int(assign.max()) + 1):
            hit = assign == piece
            if hit.any():
                table[a:b][hit] = nc
                nc += 1
    out = oc_t.clone()
    out[sel] = torch.from_numpy(table).to(dev)[inv]
    return out, nc


class FlatRefiner:
    """single refinement half rounds on device-resident colour tensors (``lhvi_color_refine_factors`` / ``_rvs``): the hash
    relabelling, repeated through the radix sort when its table overflows"""

    def __init__(self, flat, dg, sym):
        torch = _abi.require_gpu()
        self.dg = dg
        self.sym = sym.to(torch.uint8).contiguous() if torch.is_tensor(sym) else _abi.to_dev(np.asarray(sym, dtype=np.uint8))
        self.ws_bytes = int(_abi.lib().lhvi_color_workspace_bytes(dg.g))
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=dg.device)
        self.res = torch.zeros(4, dtype=torch.int32, device=dg.device)

    def _half(self, call, what):
        for method in (_abi.COLOR_HASH, _abi.COLOR_SORT):
            call(method)
            n, collision, overflow = (int(x) for x in self.res.cpu()[:3])
            if collision:
                raise _abi.LhviError('colour refinement fingerprint collision (%s side)' % what)
            if not overflow:
                return n
        raise _abi.LhviError('colour refinement: table overflow reported by the sort path')

    def factors(self, rvc, fc):
        import torch
        out = torch.empty_like(fc)
        l = _abi.lib()
        self._half(lambda m: _abi.check(l.lhvi_color_refine_factors(
            self.dg.g, _abi.ptr(self.sym), _abi.ptr(rvc), _abi.ptr(fc), _abi.ptr(out), _abi.ptr(self.res), _abi.ptr(self.ws),
            self.ws_bytes, m, _abi.stream_ptr())), 'factor')
        return out

    def rvs(self, fc, rvc):
        import torch
        out = torch.empty_like(rvc)
        l = _abi.lib()
        self._half(lambda m: _abi.check(l.lhvi_color_refine_rvs(
            self.dg.g, _abi.ptr(fc), _abi.ptr(rvc), _abi.ptr(out), _abi.ptr(self.res), _abi.ptr(self.ws), self.ws_bytes, m,
            _abi.stream_ptr())), 'variable')
        return out


def run_c2f_flat(gflat, tg, engine, refiner, iteration, c2f, k_mean_k, k_mean_iteration, draw, rvc0, fc0, observer=None,
                 keep_history=False, timing=None):
    """``run_c2f`` for a ground ``FlatGraph`` (``RelationalGraph.ground_flat``, ``flatten(g)``): no Python object per ground atom.
    The colour arrays are tensors on `tg`'s device and stay there: the half rounds of the refinement (`refiner`: ``FlatRefiner``
    in the product), the representatives and the rows they contribute to the two lifted graphs of a sweep (``rv_side_graph_t``,
    ``lifting.lift_flat``) run there; what reaches the host per sweep is the colours of the OBSERVED variables (the k-means
    evidence splits) and lifted-size arrays.  `rvc0`, `fc0`: the coarse initial colours (``initial_colors_flat(gflat, False)``; arrays or tensors).
    `timing`: optional dict that receives per-sweep lists of seconds (`lift`: both re-liftings incl. refinement,
    `setup`: building the two solver states, `sweep`: the message kernels).
    Returns (final state, final lifted flat (factor side), rv colours, factor colours (tensors), history)."""
    import time
    import torch
    from .lifting import lift_flat
    dev = tg.device
    values = gflat.var_value
    obs_idx = np.flatnonzero(~np.isnan(values))
    ovals = values[obs_idx]
    obs_idx_t = torch.from_numpy(obs_idx).to(dev)
    as_t = lambda a, dt=torch.int32: (a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))).to(dev).to(dt)
    rvc, fc = as_t(rvc0), as_t(fc0)
    nc, nfc = int(rvc.max().item()) + 1, int(fc.max().item()) + 1
    ver = {'rv': 0, 'f': 0}                      # bumped whenever the rv / factor partition changes
    sync = torch.cuda.synchronize if dev.type == 'cuda' else (lambda: None)
    clock = {'lift': 0.0, 'setup': 0.0, 'sweep': 0.0}

    def tick(name, t0):
        if timing is not None:
            sync()
            clock[name] += time.perf_counter() - t0
        return time.perf_counter()

    distinct_vals, vcode = np.unique(ovals, return_inverse=True)
    ovals_t = torch.from_numpy(np.ascontiguousarray(ovals)).to(dev)
    vcode_t = torch.from_numpy(vcode.astype(np.int64)).to(dev)

    def split_evidence(k, its, epsilon, use_sqrt):
        nonlocal rvc, nc
        if not obs_idx.size:
            return
        oc = rvc[obs_idx_t].long()
        new_oc, nc2 = split_evidence_tensors(ovals_t, vcode_t, distinct_vals, oc, nc, k, its, epsilon, use_sqrt)
        if nc2 != nc:
            rvc = rvc.clone()
            rvc[obs_idx_t] = new_oc.to(rvc.dtype)
            ver['rv'] += 1
        nc = nc2

    def refine_rvs():
        """one rv half round; a refinement only splits, so an unchanged number of colours means an unchanged partition: the old ids
        stay then (the kernels number their colours afresh every time) and the sweep's graphs and states can be reused"""
        nonlocal rvc, nc
        new = refiner.rvs(fc, rvc)
        n_new = int(new.max().item()) + 1
        if n_new != nc:
            rvc, nc = new, n_new
            ver['rv'] += 1

    def refine_factors():
        nonlocal fc, nfc
        new = refiner.factors(rvc, fc)
        n_new = int(new.max().item()) + 1
        if n_new != nfc:
            fc, nfc = new, n_new
            ver['f'] += 1

    def evidence_variance():
        if not obs_idx.size:
            return np.zeros(0)
        oc = rvc[obs_idx_t].long()
        n, mean, var = colour_mean_var(oc, ovals_t, nc)
        return var[n > 0].cpu().numpy()

    def record():
        return (rvc.cpu().numpy().copy(), fc.cpu().numpy().copy()) if (keep_history or observer is not None) else None

    t0 = time.perf_counter()
    split_evidence(2, 50, 0.0, True)                                                # HLBP:440
    refine_factors()
    refine_rvs()
    history = []
    G1, pair_phi, rep = rv_side_graph_t(gflat, tg, rvc, fc)
    g1_ver = (ver['rv'], ver['f'])
    g2_ver = pe1_ver = pe2_ver = None
    pe1 = pe2 = None
    t0 = tick('lift', t0)
    st1 = engine.make(G1, sides='v')
    engine.init(st1)
    t0 = tick('setup', t0)
    k = 0
    history.append(record())
    engine.install(st1, draw(k, G1, engine.host(engine.get(st1, 'q'))))
    k += 1
    var = evidence_variance()
    epsilon = float(var.max()) if var.size else 0                                   # HLBP:460-465
    d = (epsilon - c2f) / iteration
    epsilon -= d
    st2 = G2 = None
    t0 = tick('sweep', t0)
    per_sweep = []
    import os
    no_reuse = os.environ.get('LHVI_C2F_NO_REUSE') == '1'          # diagnostic: rebuild both graphs and states every sweep
    for i in range(iteration):
        mark = dict(clock)
        if no_reuse:
            g1_ver = g2_ver = pe1_ver = pe2_ver = None
        if i > 0:
            # ---- split_evidence + split_rvs (HLBP:475-485)
            old_rvc = rvc
            split_evidence(k_mean_k, k_mean_iteration, epsilon, False)
            epsilon = max(epsilon - d, c2f)
            refine_rvs()
            if g1_ver == (ver['rv'], ver['f']) and pe1_ver == (g1_ver, g2_ver):
                # nothing split since this sweep's two graphs were built: same graphs, same states, same maps -- only the rows move
                t0 = tick('lift', t0)
                for name in ('f2v', 'eta'):
                    engine.set(st1, name, engine.gather(engine.get(st2, name), pe1))
                for name in ('q', 'particles', 'old_particles', 'uniq'):
                    engine.set(st1, name, engine.get(st2, name))
            else:
                if g1_ver != (ver['rv'], ver['f']):
                    G1, pair_phi, rep = rv_side_graph_t(gflat, tg, rvc, fc)
                    g1_ver = (ver['rv'], ver['f'])
                    parent = old_rvc.long()[rep].cpu().numpy()
                    st1 = None
                else:
                    parent = np.arange(G1.V)
                pe1 = edges_from_factor_side(G1, pair_phi, parent, G2)
                pe1_ver = (g1_ver, g2_ver)
                t0 = tick('lift', t0)
                new1 = st1 if st1 is not None else engine.make(G1, sides='v')
                for name in ('f2v', 'eta'):
                    engine.set(new1, name, engine.gather(engine.get(st2, name), pe1))
                for name in ('q', 'particles', 'old_particles', 'uniq'):
                    engine.set(new1, name, engine.gather(engine.get(st2, name), parent))
                st1 = new1
            t0 = tick('setup', t0)
        engine.v2f(st1)
        last = i == iteration - 1
        if not last:
            engine.proposal(st1)
        t0 = tick('sweep', t0)
        # ---- split_factors (HLBP:509-513; HLBP:536 for the last sweep)
        old_fc = fc
        refine_factors()
        if g2_ver == (ver['rv'], ver['f']) and pe2_ver == (g2_ver, g1_ver):
            t0 = tick('lift', t0)
            if last:                                   # (the last sweep computes no f -> v half: a fresh state's table is zero)
                engine.get(st2, 'f2v')[...] = 0
        else:
            if g2_ver != (ver['rv'], ver['f']):
                G2 = lift_flat(gflat, rvc, fc, dg=tg)
                G2.rep_ground = G1.rep_ground
                g2_ver = (ver['rv'], ver['f'])
                parent_f = old_fc.long()[_rep_of(fc, G2.F, gflat.F)].cpu().numpy()
                st2 = None
            else:
                parent_f = np.arange(G2.F)
            n_phi = int(max(pair_phi.max() if pair_phi.size else 0, parent_f.max() if parent_f.size else 0)) + 1
            pe2 = edges_from_variable_side(G2, parent_f, G1, pair_phi, n_phi)
            pe2_ver = (g2_ver, g1_ver)
            t0 = tick('lift', t0)
            if st2 is None:
                st2 = engine.make(G2)
            elif last:                                 # (a kept state on the last sweep: same as above)
                engine.get(st2, 'f2v')[...] = 0
        for name in ('v2f', 'eta'):
            engine.set(st2, name, engine.gather(engine.get(st1, name), pe2))
        for name in ('q', 'particles', 'old_particles', 'uniq'):
            engine.set(st2, name, engine.get(st1, name))
        t0 = tick('setup', t0)
        if not last:
            history.append(record())
            if observer is not None:
                observer(k, history[-1][0], old_fc.cpu().numpy(), G1, pair_phi, st1)
            engine.install(st2, draw(k, G2, engine.host(engine.get(st2, 'q'))))
            k += 1
            engine.f2v(st2)
            t0 = tick('sweep', t0)
        per_sweep.append({name: clock[name] - mark[name] for name in clock})
    if timing is not None:
        timing['per_sweep'] = per_sweep
        timing['total'] = dict(clock)
    return st2, G2, rvc, fc, history
