"""Coarse-to-fine lifted variational inference: ``C2FVarInference.VarInference`` of the reference
(``C2FVarInference.py:11-461``) on the GPU.

The reference starts from the coarsest lifting -- continuous evidence merged regardless of its values
(``CompressedGraph.init_cluster(False)``, C2FVI:302) -- and treats an evidence cluster whose members' values differ as a
*Gaussian observation* N(mean, variance) (C2FVI:120-136, 253-261).  Every ``update_obs_its`` ADAM iterations it splits the
evidence clusters whose standard deviation exceeds a shrinking threshold by k-means (C2FVI:33-37, CGWO:236-247), re-runs
colour passing, and lets the new clusters inherit the parameters and ADAM moments of the cluster they came from
(C2FVI:39-60).  This file restates that schedule on colour arrays; the numerical work of a round -- the expectation step
with Gaussian observations and the ADAM updates on the round's lifted graph -- is done by an *engine* (the HIP kernels of
``csrc/vi.hip`` in the product, the CPU oracle in the tests), so the schedule itself is pure host logic.

Because clusters only ever split and children inherit, the parameters are kept per GROUND variable between rounds: a
round gathers them through each cluster's first member and scatters them back afterwards.

Documented deviation (same as ``lhvi/c2f.py``): the reference seeds k-means and draws the initial parameters in the
iteration order of Python ``set``s of objects, which changes from run to run; here both follow ground-variable order.
Identical whenever a cluster holds at most ``k_mean_k`` distinct values and the initial parameters are injected
(``init=``), as in the golden fixtures.
"""
from __future__ import annotations

import time
from math import sqrt

import numpy as np

from . import _abi
from .c2f import DeviceRefiner, _first_member
from .flat import flatten
from .lifting import CompressedGraph, initial_colors, kmeans_assign
from .vi import _Variational


def _parents(old, new):
    """parent (old colour) of every new colour; a refinement never merges"""
    n = int(new.max()) + 1
    return old[_first_member(new, n)]


def _members_by_colour(rvc, mask):
    """members of every colour among `mask`, grouped once: yields (colour, member indices in ground order)"""
    idx = np.flatnonzero(mask)
    order = idx[np.argsort(rvc[idx], kind='stable')]
    cols, start = np.unique(rvc[order], return_index=True)
    bounds = np.append(start, order.size)
    for i, c in enumerate(cols):
        yield int(c), order[bounds[i]:bounds[i + 1]]


def evidence_variances(values, rvc):
    """{cluster: np.var of its members' values} for the evidence clusters (``SuperRV.get_variance``, CGWO:38-39)"""
    return {c: float(np.var(values[m])) for c, m in _members_by_colour(rvc, ~np.isnan(values))}


def split_rvs_tracked(refiner, values, rvc, fc, tracked):
    """``VarInference.split_rvs`` (C2FVI:39-60): one structural refinement of the rv clusters; an evidence cluster that
    splits puts all its pieces into ``clustered_evidence``, one that stays whole and has a single member leaves it"""
    new = refiner.rvs(fc, rvc)
    parent = _parents(rvc, new)
    nchild = np.bincount(parent, minlength=int(rvc.max()) + 1)
    size = np.bincount(new)
    obs = ~np.isnan(values)
    out = set()
    for c in np.unique(new[obs]):
        p = int(parent[c])
        if nchild[p] > 1:
            out.add(int(c))
        elif p in tracked and size[c] > 1:
            out.add(int(c))
    return new, out


def cp_run(refiner, values, rvc, fc, tracked):
    """``VarInference.cp_run`` (C2FVI:62-67)"""
    prev = -1
    while prev != int(rvc.max()) + 1:
        prev = int(rvc.max()) + 1
        fc = refiner.factors(rvc, fc)
        rvc, tracked = split_rvs_tracked(refiner, values, rvc, fc, tracked)
    return rvc, fc, tracked


def split_evidence_pass(values, rvc, tracked, k, iteration, epsilon, member_order=None):
    """``CompressedGraph.split_evidence`` (CGWO:236-247): one pass over the clusters in ``clustered_evidence``.
    ``member_order(members) -> permutation or None``: the order in which k-means walks a cluster's members (default: ground
    order; the reference walks a Python set, see ``lifting.kmeans_assign``)"""
    rvc = rvc.copy()
    tracked = set(tracked)
    nxt = int(rvc.max()) + 1
    for c, members in list(_members_by_colour(rvc, np.isin(rvc, sorted(tracked)))):     # (one grouping per pass, not one scan per cluster)
        vals = values[members]
        if not (np.sqrt(np.var(vals)) > epsilon):
            continue
        assign = kmeans_assign(vals, k, iteration, order=member_order(members) if member_order else None)
        if assign is None:                         # a single member or a single value: nothing to split
            if members.size == 1:
                tracked.discard(c)
            continue
        for idx in range(int(assign.max()) + 1):   # piece 0 keeps the colour (the reference reuses the SuperRV instance)
            sel = members[assign == idx]
            if idx > 0 and sel.size:
                rvc[sel] = nxt
                cid, nxt = nxt, nxt + 1
            else:
                cid = c
            if sel.size and np.var(values[sel]) > epsilon:        # (variance here, its square root above: CGWO:239,244)
                tracked.add(cid)
    return rvc, tracked


def split_evidence(values, rvc, tracked, k, iteration, epsilon, member_order=None):
    """``VarInference.split_evidence`` (C2FVI:33-37): passes until the number of clusters stops changing"""
    prev = -1
    while prev != int(rvc.max()) + 1:
        prev = int(rvc.max()) + 1
        rvc, tracked = split_evidence_pass(values, rvc, tracked, k, iteration, epsilon, member_order)
    return rvc, tracked


def run_c2fvi(g, engine, refiner, K, iteration, lr, opts, init=None, observer=None):
    """Drive one coarse-to-fine run.  ``engine.stage(flat, obs_var)`` returns an object with ``load(params)``,
    ``adam(n, t, lr) -> [free energy after every update]``, ``dump() -> params`` where params is a dict of arrays
    ``w_tau [K]``, ``eta_c [V,K,2]``, ``tau_d [V,K,D]`` and the ADAM moments ``m_* / s_*`` of each.
    ``opts``: the reference's class attributes (k_mean_k, k_mean_its, update_obs_its, output_its, min_obs_var, gaussian_obs).
    ``init``: (eta_c [Vg,K,2], tau_d [Vg,K,D]) per ground variable to start from instead of drawing (C2FVI:263-279).
    ``observer(round, dict)`` sees the state right before every round's ADAM updates.
    Returns dict(rvc, fc, flat, cg, stage, params (per ground variable), fe_log, obs_var)."""
    gflat = flatten(g)
    values = gflat.var_value
    obs = ~np.isnan(values)
    hid_c, hid_d = gflat.var_hidden & gflat.var_cont, gflat.var_hidden & ~gflat.var_cont
    D = int(gflat.var_nstates[hid_d].max()) if hid_d.any() else 1
    rvc, fc = initial_colors(g, is_split_cont_evidence=False)                       # C2FVI:302
    tracked = set(int(c) for c in np.unique(rvc[obs & gflat.var_cont]))             # CGWO:204-210
    Vg = gflat.V
    P = dict(w_tau=np.zeros(K), eta_c=np.ones((Vg, K, 2)), tau_d=np.zeros((Vg, K, D)))
    if init is not None:
        P['eta_c'] = np.nan_to_num(np.array(init[0], dtype=np.float64), nan=1.0)
        src = np.nan_to_num(np.array(init[1], dtype=np.float64), nan=0.0)
        P['tau_d'][:, :, :min(D, src.shape[2])] = src[:, :, :D]
    else:                                                                           # one draw per coarse hidden cluster
        for c, members in _members_by_colour(rvc, gflat.var_hidden):
            if gflat.var_cont[members[0]]:
                P['eta_c'][members, :, 0] = np.random.rand(K) * 3 - 1.5
            else:
                d = int(gflat.var_nstates[members[0]])
                P['tau_d'][members, :, :d] = np.random.rand(K, d) * 10
    for name in ('w_tau', 'eta_c', 'tau_d'):
        P['m_' + name] = np.zeros_like(P[name])
        P['s_' + name] = np.zeros_like(P[name])
    t = 0
    rvc, fc, tracked = cp_run(refiner, values, rvc, fc, tracked)                    # C2FVI:324
    ev = evidence_variances(values, rvc)
    epsilon = max([sqrt(v) for v in ev.values()] + [0])                             # C2FVI:326-331
    d = epsilon * opts['update_obs_its'] / (iteration - opts['output_its'])
    epsilon -= d
    fe_log = []
    stage = flat = cg = obs_var = None
    for rnd in range(int(iteration / opts['update_obs_its'])):                      # C2FVI:338-345
        rvc, tracked = split_evidence(values, rvc, tracked, opts['k_mean_k'], opts['k_mean_its'], epsilon,
                                      opts.get('kmeans_member_order'))
        rvc, fc, tracked = cp_run(refiner, values, rvc, fc, tracked)
        epsilon = max(epsilon - d, opts['min_obs_var'])
        cg = CompressedGraph(g)
        cg.set_colors(rvc, fc)
        flat = flatten(cg, require_device_potentials=True)
        rep = _first_member(rvc, flat.V)
        ev = evidence_variances(values, rvc)
        obs_var = np.zeros(flat.V)
        if opts['gaussian_obs']:
            for c, v in ev.items():
                if v > opts['min_obs_var']:
                    obs_var[c] = v
        if observer is not None:
            observer(rnd, dict(rvc=rvc, fc=fc, tracked=set(tracked), flat=flat, obs_var=obs_var, params=P, t=t))
        stage = engine.stage(flat, obs_var)
        if opts.get('log_map_likelihood'):
            # the reference's other log (C2FVI:393-404): -log phi of the GROUND graph at the current MAP after every update; the
            # MAP of a ground variable is its cluster's (same parameters), an observed one keeps its own value
            def loglik(cmap, rvc=rvc):
                from .utils import log_likelihood
                assignment = {}
                for i, rv in enumerate(gflat.rvs):
                    if rv.value is not None:
                        assignment[rv] = rv.value
                    else:
                        m = cmap[rvc[i]]
                        assignment[rv] = float(m) if rv.domain.continuous else type(rv.domain.values[0])(m)
                return log_likelihood(g, assignment)
            stage.loglik = loglik
        local = {name: (a if name.endswith('w_tau') else np.ascontiguousarray(a[rep])) for name, a in P.items()}
        stage.load(local)
        fe_log += stage.adam(opts['update_obs_its'], t, lr)
        t += opts['update_obs_its']
        for name, a in stage.dump().items():           # (a round without discrete hidden variables keeps a narrower tau_d)
            if name.endswith('w_tau'):
                P[name] = np.array(a, dtype=np.float64)
            else:
                P[name][..., :a.shape[-1]] = a[rvc]
    return dict(rvc=rvc, fc=fc, flat=flat, cg=cg, stage=stage, params=P, fe_log=fe_log, obs_var=obs_var, t=t)


# ---- the same schedule on arrays (ground FlatGraph in, no Python object per ground atom) ---------------------------------------
def _kmeans_distinct(distinct, cnt, k, iteration):
    """the k-means of ``lifting.kmeans_assign`` on a cluster's DISTINCT values (in first-appearance order of its members) and their
    multiplicities: centroids seeded with the first k of them, `iteration` Lloyd rounds (``np.bincount`` adds its weights in input
    order like the reference's loop over its Counter).  Returns (piece of every distinct value, nearest(x) for further points), or
    None when there is nothing to split."""
    kk = min(k, distinct.size)
    if kk <= 1:
        return None
    cnt = cnt.astype(np.float64)
    cen = distinct[:kk].copy()

    def nearest(x):
        if kk != 2:
            return np.abs(cen[None, :] - x[:, None]).argmin(axis=1)
        # two centroids (the reference's k): argmin of two columns without the 2-D temporaries -- the first wins a tie, and a NaN
        # centroid (a piece that lost every value) wins outright, the first one first, as ndarray.argmin has it
        if np.isnan(cen[0]):
            return np.zeros(x.size, dtype=np.int64)
        if np.isnan(cen[1]):
            return np.ones(x.size, dtype=np.int64)
        return (np.abs(cen[1] - x) < np.abs(cen[0] - x)).astype(np.int64)
    w = distinct * cnt
    for _ in range(iteration):
        idx = nearest(distinct)
        tot = np.bincount(idx, weights=w, minlength=kk)
        num = np.bincount(idx, weights=cnt, minlength=kk)
        with np.errstate(invalid='ignore', divide='ignore'):
            cen = tot / num
    return nearest(distinct), nearest


def _kmeans_vec(vals, k, iteration):
    """``lifting.kmeans_assign`` vectorised over the distinct values (same seeds -- the first k distinct values in member
    order --, same accumulation order), for clusters with many distinct values.  Returns the piece of every member or None."""
    if vals.size <= 1:
        return None
    distinct, first, inv, cnt = np.unique(vals, return_index=True, return_inverse=True, return_counts=True)
    order = np.argsort(first, kind='stable')            # distinct values in first-appearance (member) order
    res = _kmeans_distinct(distinct[order], cnt[order], k, iteration)
    if res is None:
        return None
    return res[1](vals)


def split_evidence_observed(ovals, oc, nc, tracked, k, iteration, epsilon):
    """``split_evidence`` on the OBSERVED members only (`ovals` their values, `oc` their colours, both in ground order; `nc`
    colours in all; `tracked` a boolean array over the colours): members are grouped once per pass by a stable sort of their
    colours.  Returns (oc, nc, tracked) with piece 0 of every split keeping its colour and the other pieces numbered from `nc` up."""
    while True:
        n = np.bincount(oc, minlength=nc).astype(np.float64)
        with np.errstate(invalid='ignore', divide='ignore'):
            mean = np.bincount(oc, weights=ovals, minlength=nc) / n
            var = np.bincount(oc, weights=(ovals - mean[oc]) ** 2, minlength=nc) / n
        todo = tracked[:nc] & (np.sqrt(np.nan_to_num(var)) > epsilon)
        if not todo.any():
            return oc, nc, tracked
        sel = np.flatnonzero(todo[oc])                        # observed members of the clusters to split
        order = sel[np.argsort(oc[sel], kind='stable')]       # grouped by colour, ground order inside a group
        cols, start = np.unique(oc[order], return_index=True)
        bounds = np.append(start, order.size)
        oc = oc.copy()
        nxt = nc
        grown = list(tracked[:nc])
        for gi, c in enumerate(cols):
            loc = order[bounds[gi]:bounds[gi + 1]]
            vals = ovals[loc]
            assign = _kmeans_vec(vals, k, iteration)
            if assign is None:
                if loc.size == 1:
                    grown[c] = False
                continue
            for piece in range(1, int(assign.max()) + 1):     # piece 0 keeps the colour and stays tracked (CGWO:240-245 only adds)
                part = loc[assign == piece]
                if part.size:
                    oc[part] = nxt
                    nxt += 1
                    grown.append(bool(np.var(ovals[part]) > epsilon))      # (variance here, its square root above: CGWO:239,244)
        tracked = np.array(grown, dtype=bool)
        if nxt == nc:
            return oc, nc, tracked
        nc = nxt


def cp_run_device(flat, sym, dg, rvc_d, fc_d, tracked, obs_mask_d):
    """``cp_run`` with the refinement run to its fixed point on the device (``lifting.refine_flat``: the same sequence of
    half rounds) on device-resident colours.  ``clustered_evidence`` after the loop in closed form: an evidence cluster is
    tracked iff it has more than one member and its ancestor at the start of the loop was tracked or ended up with several
    descendants (C2FVI:39-67: pieces of a split enter the set, a cluster that stays whole leaves it when it is a singleton --
    and the loop's last pass splits nothing).  Returns (rvc_d, fc_d, tracked [new colours], parent [new colour -> old colour])."""
    from .lifting import refine_flat
    torch = _abi.require_gpu()
    ol = rvc_d.long()
    new_r, new_f = refine_flat(flat, sym, rvc_d, fc_d, dg=dg, device_out=True)
    nl = new_r.long()
    n_new, n_old = int(nl.max().item()) + 1, int(ol.max().item()) + 1
    from .lifting import first_members
    rep = first_members(new_r, n_new, flat.V)
    parent = ol[rep]
    nchild = torch.bincount(parent, minlength=n_old)
    size = torch.bincount(nl, minlength=n_new)
    if tracked.size < n_old:
        raise _abi.LhviError('cp_run_device: the tracked flags do not cover the colours')
    tr = torch.from_numpy(np.ascontiguousarray(tracked[:n_old])).to(nl.device)
    out = obs_mask_d[rep] & (size > 1) & (tr[parent] | (nchild[parent] > 1))
    return new_r, new_f, out.cpu().numpy(), parent.cpu().numpy()


def run_c2fvi_flat(flat, engine, K, iteration, lr, opts, init=None, observer=None, dg=None):
    """``run_c2fvi`` for a ground ``FlatGraph`` (e.g. ``RelationalGraph.ground_flat``).  The colour arrays live on the device:
    every refinement runs there to its fixed point, ``lifting.lift_flat`` re-lifts from them, and what reaches the host per
    round is the colours of the OBSERVED members (the k-means evidence splits run there, vectorised) and lifted-size arrays.
    Parameters are kept per CLUSTER and handed down through the parent colour of each new cluster (clusters only split,
    children inherit: C2FVI:39-60).  ``init``: optional (eta_c [V, K, 2], tau_d [V, K, D]) per ground variable.  Returns
    dict(rvc, fc (host arrays), flat (lifted), stage, params (per cluster), fe_log, obs_var, t, relift_s (seconds per round))."""
    import time as _time
    from .lifting import initial_colors_device, lift_flat
    torch = _abi.require_gpu()
    dg = dg or _abi.DeviceGraph(flat)
    values = flat.var_value
    obs = ~np.isnan(values)
    obs_idx = np.flatnonzero(obs)
    ovals = values[obs_idx]
    obs_idx_d, obs_mask_d = _abi.to_dev(obs_idx), _abi.to_dev(obs)
    hid_d = flat.var_hidden & ~flat.var_cont
    D = int(flat.var_nstates[hid_d].max()) if hid_d.any() else 1
    rvc0_d, fc0_d, sym = initial_colors_device(flat, dg, is_split_cont_evidence=False)     # C2FVI:302 (on the device: lifting.py)
    rvc0 = rvc0_d.cpu().numpy()
    nc = int(rvc0.max()) + 1
    tracked = np.zeros(nc, dtype=bool)
    tracked[np.unique(rvc0[obs & flat.var_cont])] = True                           # CGWO:204-210
    rep0 = _first_member(rvc0, nc)
    P = dict(w_tau=np.zeros(K), eta_c=np.ones((nc, K, 2)), tau_d=np.zeros((nc, K, D)))
    if init is not None:
        P['eta_c'] = np.nan_to_num(np.asarray(init[0], dtype=np.float64)[rep0], nan=1.0)
        src = np.nan_to_num(np.asarray(init[1], dtype=np.float64)[rep0], nan=0.0)
        P['tau_d'][:, :, :min(D, src.shape[2])] = src[:, :, :D]
    else:                                                                          # one draw per coarse hidden cluster
        for c in np.flatnonzero(flat.var_hidden[rep0]):
            if flat.var_cont[rep0[c]]:
                P['eta_c'][c, :, 0] = np.random.rand(K) * 3 - 1.5
            else:
                d = int(flat.var_nstates[rep0[c]])
                P['tau_d'][c, :, :d] = np.random.rand(K, d) * 10
    for name in ('w_tau', 'eta_c', 'tau_d'):
        P['m_' + name] = np.zeros_like(P[name])
        P['s_' + name] = np.zeros_like(P[name])

    def inherit(parent):
        for name in list(P):
            if not name.endswith('w_tau'):
                P[name] = P[name][parent]

    def observed_variances(oc, n_colours):
        n = np.bincount(oc, minlength=n_colours).astype(np.float64)
        with np.errstate(invalid='ignore', divide='ignore'):
            mean = np.bincount(oc, weights=ovals, minlength=n_colours) / n
            return np.bincount(oc, weights=(ovals - mean[oc]) ** 2, minlength=n_colours) / n
    t = 0
    rvc_d, fc_d = rvc0_d, fc0_d
    rvc_d, fc_d, tracked, parent = cp_run_device(flat, sym, dg, rvc_d, fc_d, tracked, obs_mask_d)     # C2FVI:324
    inherit(parent)
    oc = rvc_d[obs_idx_d].cpu().numpy().astype(np.int64)
    var = observed_variances(oc, tracked.size)
    epsilon = float(np.sqrt(np.nanmax(var))) if obs.any() else 0.0                 # C2FVI:326-331
    d_eps = epsilon * opts['update_obs_its'] / (iteration - opts['output_its'])
    epsilon -= d_eps
    fe_log, relift = [], []
    stage = lflat = obs_var = None
    for rnd in range(int(iteration / opts['update_obs_its'])):                      # C2FVI:338-345
        t0 = _time.perf_counter()
        n_before = tracked.size
        new_oc, n_split, tracked = split_evidence_observed(ovals, oc, n_before, tracked, opts['k_mean_k'], opts['k_mean_its'], epsilon)
        moved = np.flatnonzero(new_oc != oc)                                       # the pieces that got fresh colours
        split_parent = np.arange(n_split)
        if moved.size:
            rvc_d = rvc_d.clone()
            rvc_d[obs_idx_d[_abi.to_dev(moved)]] = _abi.to_dev(new_oc[moved].astype(np.int32))
            split_parent[new_oc[moved]] = oc[moved]                                # a fresh colour's parent: the colour it left
        rvc_d, fc_d, tracked, parent = cp_run_device(flat, sym, dg, rvc_d, fc_d, tracked, obs_mask_d)
        inherit(split_parent[parent])
        epsilon = max(epsilon - d_eps, opts['min_obs_var'])
        lflat = lift_flat(flat, rvc_d, fc_d, dg=dg)
        oc = rvc_d[obs_idx_d].cpu().numpy().astype(np.int64)
        var = observed_variances(oc, lflat.V)
        obs_var = np.zeros(lflat.V)
        if opts['gaussian_obs']:
            ev = np.flatnonzero(~np.isnan(var) & (var > opts['min_obs_var']))
            obs_var[ev] = var[ev]
        torch.cuda.synchronize()
        relift.append(_time.perf_counter() - t0)
        if observer is not None:
            observer(rnd, dict(rvc=rvc_d.cpu().numpy(), fc=fc_d.cpu().numpy(), tracked=set(np.flatnonzero(tracked).tolist()), flat=lflat,
                               obs_var=obs_var, params=P, t=t))
        stage = engine.stage(lflat, obs_var)
        if opts.get('log_map_likelihood'):
            def loglik(cmap, rvc_d=rvc_d):
                from .utils import log_likelihood_flat
                x = _abi.to_dev(cmap)[rvc_d.long()]
                return log_likelihood_flat(dg, torch.where(obs_mask_d, dg.t['var_value'], x))
            stage.loglik = loglik
        stage.load(P)
        fe_log += stage.adam(opts['update_obs_its'], t, lr)
        t += opts['update_obs_its']
        for name, a in stage.dump().items():
            if name.endswith('w_tau'):
                P[name] = np.array(a, dtype=np.float64)
            else:
                P[name][..., :a.shape[-1]] = a
    return dict(rvc=rvc_d.cpu().numpy(), fc=fc_d.cpu().numpy(), flat=lflat, stage=stage, params=P, fe_log=fe_log, obs_var=obs_var,
                t=t, relift_s=relift)


class _DeviceStage(_Variational):
    """one round of the schedule on the device: the lifted graph of the round, Gaussian observations, ADAM"""

    def __init__(self, owner, flat, obs_var):
        self.K, self.T = owner.K, owner.T
        self.quad_x, self.quad_w = owner.quad_x, owner.quad_w
        self.reference_quirks = owner.reference_quirks
        self.var_threshold = owner.var_threshold
        self.time_log, self._dev, self._cache = [], None, {}
        self._obs_var_host = np.ascontiguousarray(obs_var, dtype=np.float64)     # (Gaussian observations are axes of T nodes: factor_lists)
        self._setup_flat(flat)
        self._dev['obs_var'] = _abi.to_dev(np.ascontiguousarray(obs_var, dtype=np.float64))

    def _struct(self):
        p = super()._struct()
        if 'obs_var' in self._dev:
            p.obs_var = _abi.ptr(self._dev['obs_var'])
        return p

    def load(self, P):
        self._upload_params(P['w_tau'], P['eta_c'], P['tau_d'])
        d = self._dev
        for name in ('w_tau', 'eta_c', 'tau_d'):
            for pre in ('m_', 's_'):
                dst = np.zeros(tuple(d[name].shape))
                src = np.asarray(P[pre + name], dtype=np.float64)
                w = min(dst.shape[-1], src.shape[-1])        # (a round's tau_d is as wide as ITS widest discrete variable)
                dst[..., :w] = src[..., :w]
                d[pre + name].copy_(_abi.to_dev(dst))

    loglik = None          # set by the schedule for run(log_fe=False): cluster MAPs [V] -> -log phi of the ground graph there

    def adam(self, n, t, lr):
        """n ADAM updates; returns what the reference logs after each of them: the free energy (``log_fe=True``), or with
        ``loglik`` set -- ``run(log_fe=False)``, C2FVI:393-404 -- ``log_likelihood`` of the ground graph at the current MAP"""
        self.time_log, self.total_time = [], 0
        self.alpha, self.b1, self.b2, self.eps, self.t = lr, 0.9, 0.999, 1e-8, t
        if self.loglik is not None:
            self.is_log, self.log_fe = False, True
            out = []
            for _ in range(n):
                self.ADAM_update(1)
                out.append(self.loglik(self.map_rows()))
            return out
        self.is_log, self.log_fe = True, True
        self.ADAM_update(n)
        return [fe for _, fe in self.time_log]

    def dump(self):
        d = self._dev
        out = {}
        for name in ('w_tau', 'eta_c', 'tau_d'):
            for pre in ('', 'm_', 's_'):
                out[pre + name] = d[pre + name].cpu().numpy()
        return out


class _DeviceEngine:
    def __init__(self, owner):
        self.owner = owner

    def stage(self, flat, obs_var):
        return _DeviceStage(self.owner, flat, obs_var)


class VarInference(_Variational):
    """``C2FVarInference.VarInference``: same constructor, class attributes, ``run`` / ``free_energy`` / ``belief`` /
    ``map`` surface and ``w`` / ``eta`` / ``time_log`` attributes as the reference (C2FVI:11-31,298-352,436-461)."""

    var_threshold = 0.1
    k_mean_k = 2
    k_mean_its = 10
    update_obs_its = 10
    output_its = 0
    min_obs_var = 0
    gaussian_obs = True

    def __init__(self, g, num_mixtures=5, num_quadrature_points=3):
        self._ground = g
        self.g = CompressedGraph(g)
        self._init_common(num_mixtures, num_quadrature_points)
        self.init = None              # (eta_c, tau_d) per ground variable to start from (parity tests inject the reference's draw)
        self.observer = None
        self.refiner = None           # lhvi.c2f.Refiner; default: colour refinement on the device
        self.kmeans_member_order = None   # members -> permutation: the order k-means walks a cluster in (default: ground order)

    def _options(self):
        opts = {k: getattr(self, k) for k in ('k_mean_k', 'k_mean_its', 'update_obs_its', 'output_its', 'min_obs_var', 'gaussian_obs',
                                              'kmeans_member_order')}
        opts['log_map_likelihood'] = bool(getattr(self, 'is_log', True) and not getattr(self, 'log_fe', True))      # C2FVI:393-404
        return opts

    def run(self, iteration=100, lr=0.1, is_log=True, log_fe=True):
        self.is_log, self.log_fe = is_log, log_fe
        t0 = time.process_time()
        res = run_c2fvi(self._ground, _DeviceEngine(self), self.refiner or DeviceRefiner(self._ground), self.K, iteration, lr,
                        self._options(), init=self.init, observer=self.observer)
        self._result = res
        self.g = res['cg']
        st = res['stage']
        self.flat, self.dg, self._dev, self._cache = st.flat, st.dg, st._dev, {}
        self._cont, self._disc, self.Dmax = st._cont, st._disc, st.Dmax
        self._uniform_states, self._has_disc = st._uniform_states, st._has_disc
        self.obs_var = res['obs_var']
        self.t = res['t']
        if is_log:
            self.total_time = time.process_time() - t0
            n = max(len(res['fe_log']), 1)
            self.time_log = [[self.total_time * (i + 1) / n, fe] for i, fe in enumerate(res['fe_log'])]

    def _struct(self):
        p = super()._struct()
        if self._dev is not None and 'obs_var' in self._dev:
            p.obs_var = _abi.ptr(self._dev['obs_var'])
        return p

    def _graph_like(self):
        return self.g

    def _ground_graph(self):
        return self._ground

    def _var_index(self, rv):
        c = getattr(rv, 'cluster', None)
        return self.flat.var_index[c if c in self.flat.var_index else rv]

    def rvs_belief(self, x, rvs):
        """C2FVI:243-261: a Gaussian observation contributes its pdf to every component instead of pinning x"""
        b = np.copy(self._w_host())
        for i, rv in enumerate(rvs):
            v = self._var_index(rv)
            if rv.value is not None:
                if self.gaussian_obs and self.obs_var[v] > self.min_obs_var:
                    b *= self.norm_pdf(x[i], (self.flat.var_value[v], self.obs_var[v]))
                elif x[i] != rv.value:
                    return 0
            elif rv.domain.continuous:
                eta = self._host('eta_c')[v]
                xi = x[i] if np.ndim(x[i]) == 0 else float(np.ravel(x[i])[0])
                for k in range(self.K):
                    b[k] *= self.norm_pdf(xi, eta[k])
            else:
                b *= self._host('eta_d')[v, :, rv.domain.values.index(x[i])]
        return np.sum(b)

    def belief(self, x, rv):
        return self.rvs_belief((x,), (rv.cluster,))        # C2FVI:436-437 (queries take ground rvs)
