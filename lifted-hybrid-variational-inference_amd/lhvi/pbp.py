"""Particle belief propagation on the GPU: ``EPBP`` (ground) and ``HybridLBP`` (lifted).

Same surface as the reference (``EPBPLogVersion.py:14-394``, ``HybridLBPLogVersion.py:15-536``):
``EPBP(g, n, proposal_approximation).run(iteration)``, ``.belief(x, rv, ...)``, ``.probability(a, b, rv)``,
``.map(rv)``, attributes ``.message .sample .q .eta_message``; class attributes ``var_threshold`` /
``max_log_value`` keep the reference's values.

``run`` keeps all state (particles, log-message tables, proposals) resident in HBM and issues one sweep as
four launches through the C ABI (``csrc/pbp.hip``): v2f, proposal, resample, f2v.

Sampling.  The reference draws particles with ``scipy.stats.norm(...).rvs`` on NumPy's global RNG while
iterating ``g.rvs`` (``EPBPLogVersion.py:61-70``).  ``sampler='host'`` (default) reproduces exactly that stream
(``standard_normal(n) * sqrt(var) + mu`` in ``g.rvs`` order), so a seeded run matches the reference draw for
draw when ``g.rvs`` is ordered; ``sampler='device'`` uses the counter-based Philox generator of
``lhvi_pbp_resample`` (no host round trip; what the benchmark uses); a callable
``sampler(k, flat, q) -> [V, n] array`` injects particles (parity tests inject the reference's draws).
"""
from __future__ import annotations

import os
from math import e, log, sqrt

import numpy as np

from . import _abi
from .flat import flatten


class _ParticleSweep:
    var_threshold = 3
    max_log_value = 700
    _epbp_discrete = True
    verbose = False
    listed_proposal = True          # the proposal kernel starts from per-variable records (else it walks the graph arrays)
    dynamic_f2v = True              # the persistent f2v kernels claim their work in chunks (else static striding)
    paired_light = True             # the light edges are served per factor (pair_desc) instead of per edge (light_desc)
    cq_routing = True               # conditionally quadratic MLN formulas go to the quadratic-family kernels (else: generic kernel)
    small_f2v = True                # heavy-class edges with at most 16 / 32 particles on both sides: four / two edges per wavefront
    long_grid_min_edges = 1 << 16   # edges with np + T > 128 join the heavy kernel's list (grid recurrence, T <= 128) from this many on
    listed_resample = True          # the device sampler draws for the hidden continuous variables only, two per wavefront
    sliced_proposal = True          # rows of more than prop_slice incident edges are cut into slices, a wavefront per slice
    prop_slice = 64
    packed_v2f = True               # variables with at most four particles share a wavefront in the v -> f half (sixteen each)
    overlap_f2v = True              # the short f -> v kernels (pair / light / cq / generic lists) on a second stream beside the heavy kernel, which
                                    # then leaves a workgroup per CU free (LHVI_PBP_SHARE_CUS): 15.20 -> 14.66 ms per sweep on the headline graph (the
                                    # heavy kernel 10.65 -> 11.26 ms, the 1.25 ms of the others hidden; same bits).  Only for a heavy list of at
    overlap_min_heavy = 1 << 16     # least this many edges: the few-particle kernels lose by it (4.55 -> 4.60 ms at n = 16), small graphs gain nothing
    fused_var_kernel = True         # few particles, device sampler: v -> f, proposal update and the new sample of a continuous variable in
                                    # ONE pass over its rows (lhvi_pbp_var_fused) instead of three launches; same bits
    fused_max_particles = 32        # ... up to this many particles.  With a row's loads in flight together (csrc/pbp.hip, FUSED_CH) the fused kernel
                                    # wins at every width: n = 10 / 16 / 20 / 32: 5.65 / 5.83 / 8.56 / 9.53 -> 4.97 / 5.03 / 8.25 / 9.30 ms per sweep
                                    # (profiles/r05_experiments.md item 7; walked edge by edge it was a draw at n <= 16 and a loss beyond)
    exact_queries = False           # map / probability / belief answer per-variable calls from ONE batched pass over all
                                    # variables, made at the first call after run() (True: one fminbound / log_area / quad per call)
    map_mode = 'fminbound'          # what the batched map() runs per variable: the reference's fminbound iteration (lhvi_pbp_map_brent)
                                    # or 'global': scan + bracket refinement (map_all)

    # ---- set-up ------------------------------------------------------------------------------
    def _setup(self, graph_like, flat=None, edge_key=None, sides='vf', edge_skip=None, owned=None):
        """`edge_key` (sharded runs): 0 / 1 per edge; every f2v work list is ordered key-0 edges first and ``part_counts``
        gives the length of that first part per list (heavy, light, fast, generic).
        `edge_skip` (owner-computes shards): boolean per edge; the f -> v message of a marked edge is somebody else's to compute,
        it enters no work list.
        `owned` (owner-computes shards): the per-variable lists of the v -> f half and of the proposal update hold the variables
        below this index only (the rank's own; the sampler's list keeps all of them: ghosts are drawn here too).
        `sides`: which half sweeps this state will run -- 'v' (v -> f, proposal, sampling: the variable-side state of a
        coarse-to-fine sweep), 'f' (f -> v and the queries: its factor-side state) or both; the work lists of the other half
        are not built."""
        flat = flat if flat is not None else flatten(graph_like, require_device_potentials=True)
        self.flat = flat
        self.dg = dg = _abi.DeviceGraph(flat)
        torch = _abi.require_gpu()
        n = self.n
        sizes = np.diff(flat.dom_ptr)
        cont = flat.dom_cont.astype(bool)
        self.T = int(sizes[cont].max()) if cont.any() else 0
        nst = flat.var_nstates
        if (flat.var_hidden & ~flat.var_cont & (nst > n)).any():
            raise _abi.LhviError('a discrete variable has more states than particle slots n=%d' % n)
        if (flat.var_hidden & flat.var_cont & (np.diff(flat.var_ptr) == 0)).any():
            # the reference fails in gaussian_product (`0 ** -1`, EPBP:30-41) on the first proposal update of such a variable
            raise ZeroDivisionError('a hidden continuous variable has no incident factor: its proposal is an empty product')
        self.np_host = np.where(flat.var_hidden, np.where(flat.var_cont, n, nst), 0).astype(np.int32)
        host_lists = {'np_dev': self.np_host}          # the per-variable lists below go to the device in one copy (_abi.upload)
        S = n + self.T
        self.f2v = dg.zeros(flat.E, S)
        self.v2f = dg.zeros(flat.E, n)
        self.eta = dg.zeros(flat.E, 2)
        self.q_dev = dg.zeros(flat.V, 2)
        self.particles = dg.zeros(flat.V, n)
        self.old_particles = dg.zeros(flat.V, n)
        self.uniq = torch.zeros(flat.V, n, dtype=torch.uint8, device=dg.device)
        self.f2v_ticket = torch.zeros(16, dtype=torch.int32, device=dg.device)    # work counters + statistics of the heavy f2v kernel (LHVI_PBP_TICKET_WORDS)
        self.flags = (_abi.PBP_EP if self.proposal_approximation == 'EP' else 0) | \
                     (_abi.PBP_EPBP_DISCRETE if self._epbp_discrete else 0) | \
                     (_abi.PBP_CQ if self.cq_routing and bool((flat.pot_kind == 8).any()) else 0) | \
                     (_abi.PBP_POW2_GROUPS if os.environ.get('LHVI_PBP_POW2_GROUPS', '0') == '1' else 0)   # (tuning aid: scripts/diag/narrow_groups.sh)
        self._views, self._batched = {}, {}
        self._draws = 0
        self.cq_desc, self.n_cq = None, 0
        self.fast_edges = self.generic_edges = self._fast_list = self._generic_list = torch.zeros(1, dtype=torch.int32, device=dg.device)
        # records of the hidden continuous variables for the proposal kernel (include/lhvi.h, lhvi_pbp_t.prop_desc)
        pv_all = np.flatnonzero(flat.var_hidden & flat.var_cont)
        pv = pv_all if owned is None else pv_all[pv_all < owned]
        pd = np.zeros((pv.size, 8), dtype=np.int32)
        pdeg = np.diff(flat.var_ptr)[pv]
        pdom = flat.var_dom[pv]
        pd[:, 0], pd[:, 1], pd[:, 2], pd[:, 3] = pv, pdeg, flat.dom_ptr[pdom], sizes[pdom]
        pbase = flat.var_ptr[pv].astype(np.int64)
        for k in range(4):
            if flat.var_edge.size:
                pd[:, 4 + k] = flat.var_edge[np.minimum(pbase + np.minimum(k, np.maximum(pdeg - 1, 0)), flat.var_edge.size - 1)]
        # the device sampler's list (include/lhvi.h, lhvi_pbp_t.resample_vars); the other rows are filled once, by the first draw
        rdom = flat.var_dom[pv_all]
        rr = np.zeros((pv_all.size, 8), dtype=np.int32)
        rr[:, 0], rr[:, 1] = pv_all, self.np_host[pv_all]
        rr[:, 2:4] = np.ascontiguousarray(flat.dom_lo[rdom], dtype=np.float64).view(np.int32).reshape(-1, 2)
        rr[:, 4:6] = np.ascontiguousarray(flat.dom_hi[rdom], dtype=np.float64).view(np.int32).reshape(-1, 2)
        host_lists['resample_vars'] = rr if pv_all.size else None
        self._static_rows = False
        static = np.flatnonzero(~(flat.var_hidden & flat.var_cont))
        host_lists['_static_idx'] = static.astype(np.int64) if static.size else None
        # rows longer than prop_slice entries go in as slices of that length, a wavefront each, ahead of the ordinary records
        self.prop_hub = self.prop_partial = None
        self.n_prop_hub = 0
        hubs = np.flatnonzero(pdeg > self.prop_slice) if self.sliced_proposal else np.zeros(0, dtype=np.int64)
        if hubs.size:
            L = int(self.prop_slice)
            nsl = (pdeg[hubs] + L - 1) // L
            first = np.concatenate([[0], np.cumsum(nsl)[:-1]])
            owner = np.repeat(np.arange(hubs.size), nsl)
            within = np.arange(int(nsl.sum())) - first[owner]
            sl = np.zeros((owner.size, 8), dtype=np.int32)
            sl[:, 0], sl[:, 2], sl[:, 3] = pd[hubs[owner], 0], pd[hubs[owner], 2], pd[hubs[owner], 3]
            sl[:, 1] = -np.minimum(L, pdeg[hubs][owner] - within * L)
            sl[:, 4], sl[:, 5] = within * L, np.arange(owner.size)
            ph = np.zeros((hubs.size, 4), dtype=np.int32)
            ph[:, 0], ph[:, 1], ph[:, 2] = pv[hubs], first, nsl
            pd = np.concatenate([sl, np.delete(pd, hubs, axis=0)])
            host_lists['prop_hub'], self.n_prop_hub = ph, int(hubs.size)
            self.prop_partial = dg.zeros(owner.size, 2)
        host_lists['prop_desc'] = np.ascontiguousarray(pd) if pv.size else None
        self.n_prop_desc = int(pd.shape[0])
        # the v -> f half's split of the hidden variables (include/lhvi.h, lhvi_pbp_t.v2f_wide / v2f_narrow)
        hidden_v = np.flatnonzero(flat.var_hidden)
        if owned is not None:
            hidden_v = hidden_v[hidden_v < owned]
        narrow = self.np_host[hidden_v] <= 4
        self.v2f_lists = None
        hub = ~narrow & (np.diff(flat.var_ptr)[hidden_v] > 64) & (self.np_host[hidden_v] <= 64)
        mid16 = ~narrow & ~hub & (self.np_host[hidden_v] <= 16)
        mid32 = ~narrow & ~hub & ~mid16 & (self.np_host[hidden_v] <= 32)
        if self.packed_v2f and hidden_v.size and (narrow.any() or hub.any() or mid16.any() or mid32.any()):
            wide = ~narrow & ~hub & ~mid16 & ~mid32
            parts = (('wide', wide), ('narrow', narrow), ('hub', hub), ('mid16', mid16), ('mid32', mid32))
            for name, m in parts:
                host_lists['v2f_' + name] = hidden_v[m].astype(np.int32) if m.any() else np.zeros(1, dtype=np.int32)
            if os.environ.get('LHVI_PBP_V2F_REC', '1') != '0':
                host_lists['v2f_wide'] = self._v2f_records(flat, hidden_v[wide])
        # ---- the fused per-variable kernel's records (lhvi_pbp_var_fused) and what is left for the three kernels
        self._fused = None
        pT = sizes[pdom] if pv.size else np.zeros(0, dtype=np.int64)
        fused_max = int(os.environ.get('LHVI_PBP_FUSED_MAX', self.fused_max_particles))          # (tuning aid: scripts/diag/fused_batch.sh)
        fz = (pdeg <= min(64, self.prop_slice)) & (pT <= 64) & (n <= min(32, fused_max)) if pv.size else np.zeros(0, dtype=bool)
        if (self.fused_var_kernel and os.environ.get('LHVI_PBP_FUSED', '1') != '0') and owned is None and self.sampler == 'device' and self.listed_proposal and self.listed_resample \
                and 'v2f_wide' in host_lists and fz.any():
            k16 = fz & (n <= 16) & (pT <= 32)
            k32a = fz & ~k16 & (pT <= 32)
            k32b = fz & ~k16 & ~k32a
            # sixteen words per variable (LHVI_PBP_FUSED_RECORDS16): the eight of include/lhvi.h, then np, var_ptr[v] and the first six
            # incident edges -- the kernel's row loads then hang on one load behind the record (LHVI_PBP_FUSED_REC16=0: eight words)
            wide_rec = os.environ.get('LHVI_PBP_FUSED_REC16', '1') != '0'
            fd = np.zeros((pv.size, 16 if wide_rec else 8), dtype=np.int32)
            fd[:, 0], fd[:, 1], fd[:, 2], fd[:, 3] = pv, pdeg, flat.dom_ptr[pdom], pT
            fd[:, 4:6] = np.ascontiguousarray(flat.dom_lo[pdom], dtype=np.float64).view(np.int32).reshape(-1, 2)
            fd[:, 6:8] = np.ascontiguousarray(flat.dom_hi[pdom], dtype=np.float64).view(np.int32).reshape(-1, 2)
            if wide_rec:
                fd[:, 8], fd[:, 9] = self.np_host[pv], flat.var_ptr[pv]
                for k in range(6):
                    has = pdeg > k
                    fd[has, 10 + k] = flat.var_edge[pbase[has] + k]
                self.flags |= _abi.PBP_FUSED_RECORDS16
            host_lists['fused_desc'] = np.ascontiguousarray(np.concatenate([fd[k16], fd[k32a], fd[k32b]]))
            fused_var = np.zeros(flat.V, dtype=bool)
            fused_var[pv[fz]] = True
            # the rest: proposal records (slices of hub rows sit at the head of pd and are never fused), sampler records, v -> f lists
            keep = ~fused_var[pd[:, 0]]
            host_lists['prop_desc_rest'] = np.ascontiguousarray(pd[keep]) if keep.any() else np.zeros((1, 8), dtype=np.int32)
            rkeep = ~fused_var[rr[:, 0]]
            host_lists['resample_rest'] = np.ascontiguousarray(rr[rkeep]) if rkeep.any() else np.zeros((1, 8), dtype=np.int32)
            rest_parts = []
            if 'v2f_wide' in host_lists:
                for name, m in parts:
                    vs = hidden_v[m]
                    vs = vs[~fused_var[vs]]
                    host_lists['v2f_rest_' + name] = (self._v2f_records(flat, vs) if name == 'wide' and os.environ.get('LHVI_PBP_V2F_REC', '1') != '0' else
                                                      vs.astype(np.int32) if vs.size else np.zeros(1, dtype=np.int32))
                    rest_parts.append((name, int(vs.size)))
            self._fused = dict(counts=(int(k16.sum()), int(k32a.sum()), int(k32b.sum())), n_prop_rest=int(keep.sum()),
                               n_resample_rest=int(rkeep.sum()), rest_parts=rest_parts)
        dev_lists = _abi.upload(host_lists)
        for name in ('np_dev', 'resample_vars', '_static_idx', 'prop_desc'):
            setattr(self, name, dev_lists[name])
        if self._fused is not None:
            F = self._fused
            F['desc'], F['prop_desc_rest'], F['resample_rest'] = dev_lists['fused_desc'], dev_lists['prop_desc_rest'], dev_lists['resample_rest']
            F['v2f_rest'] = tuple(x for name, cnt in F['rest_parts'] for x in (dev_lists['v2f_rest_' + name], cnt)) if F['rest_parts'] else None
        if 'prop_hub' in dev_lists:
            self.prop_hub = dev_lists['prop_hub']
        if 'v2f_wide' in dev_lists:
            self.v2f_lists = tuple(x for name, m in parts for x in (dev_lists['v2f_' + name], int(m.sum())))
            if os.environ.get('LHVI_PBP_V2F_REC', '1') != '0':          # (tuning aid: scripts/diag/v2f_records.sh)
                self.flags |= _abi.PBP_V2F_RECORDS         # the wide list travels as records: _v2f_records
        # static work lists of the f -> v half sweep (which kernel serves which edge)
        pad = torch.zeros(1, dtype=torch.int32, device=dg.device)       # keeps the pointers non-null when a list is empty
        self.cq_edges = pad[:0]
        self.part_counts = {'heavy': 0, 'light': 0, 'fast': 0, 'generic': 0, 'cq': 0, 'pair': 0, 'small16': 0, 'small32': 0}
        self.generic_pts_log2 = 6
        self.fast_desc = self.heavy_desc = self.light_desc = self.pair_desc = self.small16_desc = self.small32_desc = None
        self.n_heavy = self.n_light = self.n_pair = self.n_small16 = self.n_small32 = self.n_heavy_class = 0
        self.cq_terms = self.heavy_terms = self.heavy_grid_terms = 0
        self._has_f_side = 'f' in sides
        if not self._has_f_side:
            self.fast_edges = self.generic_edges = pad[:0]
            self._fast_list = self._generic_list = pad
            return
        cls = torch.zeros(max(flat.E, 1), dtype=torch.uint8, device=dg.device)
        _abi.check(_abi.lib().lhvi_pbp_classify(dg.g, dg.p, self._struct(), _abi.ptr(cls), _abi.stream_ptr()))
        cls = cls[:flat.E]
        if edge_skip is not None:
            cls = torch.where(_abi.to_dev(np.ascontiguousarray(edge_skip, dtype=bool)), torch.zeros_like(cls), cls)
        self.fast_edges = torch.nonzero((cls == 1) | (cls == 2)).flatten().to(torch.int32)
        self.generic_edges = torch.nonzero(cls == 3).flatten().to(torch.int32)
        self.cq_edges = torch.nonzero(cls == 4).flatten().to(torch.int32)
        key_dev = None
        if edge_key is not None:
            key_dev = _abi.to_dev(np.ascontiguousarray(edge_key, dtype=np.int32))
            self.fast_edges = self.fast_edges[torch.sort(key_dev[self.fast_edges.long()], stable=True).indices].contiguous()
            self.generic_edges = self.generic_edges[torch.sort(key_dev[self.generic_edges.long()], stable=True).indices].contiguous()
            self.cq_edges = self.cq_edges[torch.sort(key_dev[self.cq_edges.long()], stable=True).indices].contiguous()

        def first_part(edges):      # entries of an (ordered) edge list with key 0
            return int((key_dev[edges.long()] == 0).sum().item()) if key_dev is not None and edges.numel() else 0
        self.part_counts = {'heavy': 0, 'light': 0, 'fast': 0, 'generic': first_part(self.generic_edges),
                            'cq': first_part(self.cq_edges), 'pair': 0, 'small16': 0, 'small32': 0}
        self._fast_list = self.fast_edges if self.fast_edges.numel() else pad
        self._generic_list = self.generic_edges if self.generic_edges.numel() else pad
        # lanes per generic edge: smallest power of two covering its output points (particles + integral points)
        ge = self.generic_edges.cpu().numpy()
        tv = flat.edge_var[ge]
        pts = self.np_host[tv] + np.where(flat.var_cont[tv], flat.var_nstates[tv], 0) if ge.size else np.zeros(0, dtype=int)
        self.generic_pts_log2 = int(min(6, max(0, int(np.ceil(np.log2(max(int(pts.max()), 1)))) if ge.size else 6)))
        ncq = int(self.cq_edges.numel())
        if ncq:
            # two-partner edges of conditionally quadratic factors (include/lhvi.h, lhvi_pbp_describe_cq)
            cqd = torch.empty(ncq * 2 * _abi.PBP_DESC_BYTES, dtype=torch.uint8, device=dg.device)
            _abi.check(_abi.lib().lhvi_pbp_describe_cq(dg.g, dg.p, self._struct(), _abi.ptr(self.cq_edges), ncq,
                                                       _abi.ptr(cqd), _abi.stream_ptr()))
            self.cq_desc, self.n_cq = cqd, ncq
            cw = cqd.view(torch.int32).view(ncq, 2 * _abi.PBP_DESC_BYTES // 4).to(torch.int64)
            # (output point, partner particle, state) terms: type 1: (np + T) * ny * S;  type 2: S * nx * ny
            self.cq_terms = int(torch.where(cw[:, 2] == 1, (cw[:, 4] + cw[:, 5]) * cw[:, 9] * cw[:, 3],
                                            cw[:, 3] * cw[:, 12] * cw[:, 9]).sum().item())
        nf = int(self.fast_edges.numel())
        if nf:
            desc = torch.empty(nf * _abi.PBP_DESC_BYTES, dtype=torch.uint8, device=dg.device)
            _abi.check(_abi.lib().lhvi_pbp_describe(dg.g, dg.p, self._struct(), _abi.ptr(self.fast_edges), nf,
                                                    _abi.ptr(desc), _abi.stream_ptr()))
            # split off the edges the specialised kernel serves (descriptor words: 4 = class, 6 = potential kind, 7 = nj, 8 = np, 9 = T)
            words = desc.view(torch.int32).view(nf, _abi.PBP_DESC_BYTES // 4)
            # (a uniform grid of up to 128 integral points is tabulated by the recurrence, whatever np + T; otherwise two rounds of 64 points)
            base = (words[:, 4] == 1) & (words[:, 6] != 4)
            # edges with few particles on both sides go four / two to a wavefront, whatever their number of integral points
            # (include/lhvi.h, small16_desc)
            small16 = small32 = torch.zeros_like(base)
            if self.small_f2v:
                small16 = base & (words[:, 7] <= 16) & (words[:, 8] <= 16)
                small32 = base & ~small16 & (words[:, 7] <= 32) & (words[:, 8] <= 32)
            small = small16 | small32
            on_recurrence = (words[:, 15] == 1) & (words[:, 7] >= 24) & (words[:, 9] <= 128) & (words[:, 8] <= 128)
            two_rounds = words[:, 8] + words[:, 9] <= 128
            if int((on_recurrence & ~two_rounds & ~small).sum().item()) < self.long_grid_min_edges:
                on_recurrence = two_rounds          # (a short list would only add a launch to a launch-bound sweep)
            heavy = (base & (words[:, 7] <= 64) & (two_rounds | on_recurrence)) | small       # the heavy CLASS: the three lists together
            rows = desc.view(nf, _abi.PBP_DESC_BYTES)
            self.small16_desc, self.small32_desc = rows[small16].contiguous(), rows[small32].contiguous()
            self.n_small16, self.n_small32 = int(self.small16_desc.shape[0]), int(self.small32_desc.shape[0])
            self.n_heavy_class = int(heavy.sum().item())
            self.heavy_desc = rows[heavy & ~small].contiguous()
            self.n_heavy = int(self.heavy_desc.shape[0])
            # (output point, partner particle) terms of the heavy kernel: sum over its edges of (np + T) * nj
            hw = words[heavy].to(torch.int64)
            self.heavy_terms = int(((hw[:, 8] + hw[:, 9]) * hw[:, 7]).sum().item())
            # of those, the terms at the integral points of edges served by the grid recurrence (word 15: uniform grid;
            # at least 24 partner particles, T <= 128; the kernel's range guard is data dependent and assumed to pass)
            # (the few-particle kernel takes the recurrence for every edge with a uniform grid, whatever its particle count)
            on_grid = (hw[:, 15] == 1) & (hw[:, 9] <= 128) & ((hw[:, 7] >= 24) | small[heavy])
            self.heavy_grid_terms = int((hw[:, 9] * hw[:, 7])[on_grid].sum().item())
            light = ~heavy & (words[:, 14] != 0)          # word 14: set by lhvi_pbp_describe for the light kernel's edges
            self.light_desc = rows[light].contiguous()
            self.n_light = int(self.light_desc.shape[0])
            rest = ~heavy & ~light
            self.fast_desc = rows[rest].contiguous()
            all_fast = self.fast_edges
            self.fast_edges = all_fast[rest].contiguous()
            self._fast_list = self.fast_edges if self.fast_edges.numel() else pad
            # (sharded runs: every list is ordered interior part first -- all_fast is -- and split at these counts)
            self.part_counts.update(heavy=first_part(all_fast[heavy & ~small]), light=first_part(all_fast[light]),
                                    fast=first_part(self.fast_edges), small16=first_part(all_fast[small16]),
                                    small32=first_part(all_fast[small32]))
            self._build_pairs(key_dev)

    def _v2f_records(self, flat, vs):
        """``lhvi_pbp_t.v2f_wide`` as records (LHVI_PBP_V2F_RECORDS, include/lhvi.h): variable, incident edges, particles, domain and
        the first four incident edges in row order, so that the kernel's row loads hang on one scalar load"""
        rec = np.zeros((max(int(vs.size), 1), 8), dtype=np.int32)
        if vs.size:
            deg = np.diff(flat.var_ptr)[vs]
            rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3] = vs, deg, self.np_host[vs], flat.var_dom[vs]
            base = flat.var_ptr[vs].astype(np.int64)
            for k in range(4):
                if flat.var_edge.size:
                    rec[:, 4 + k] = flat.var_edge[np.minimum(base + np.minimum(k, np.maximum(deg - 1, 0)), flat.var_edge.size - 1)]
        return rec

    def _build_pairs(self, key_dev):
        """``lhvi_pbp_t.pair_desc``: one record per HybridQuadratic(1 discrete, 1 continuous) factor from the per-edge light
        descriptors (words 0 e, 1 target, 2 partner, 3 partner's canonical edge, 7 nj, 8 np, 9 T, 10 grid base, 12-13 partner
        value, 14 type, 16-27 coefficients), on the device.  A type-1 entry (continuous target) is joined with the type-2 entry
        (discrete target) of the same factor when there is one; what is left of either type becomes a one-sided record."""
        torch = _abi.require_gpu()
        if not self.paired_light or self.light_desc is None or self.n_light == 0:
            return
        dev = self.light_desc.device
        w = self.light_desc.view(torch.int32).view(-1, 32).long()
        dd = self.light_desc.view(torch.float64).view(-1, 16)
        i1, i2 = torch.nonzero(w[:, 14] == 1).flatten(), torch.nonzero(w[:, 14] == 2).flatten()
        w1, w2, d1, d2 = w[i1], w[i2], dd[i1], dd[i2]
        lookup = torch.full((max(self.flat.E, 1),), -1, dtype=torch.int64, device=dev)
        lookup[w2[:, 0]] = torch.arange(i2.numel(), device=dev)
        j = torch.where(torch.isnan(d1[:, 6]), lookup[w1[:, 3]], torch.full_like(w1[:, 3], -1))      # partner hidden: its own entry
        taken = torch.zeros(i2.numel(), dtype=torch.bool, device=dev)
        taken[j[j >= 0]] = True
        rest = torch.nonzero(~taken).flatten()
        n1, n2 = int(i1.numel()), int(rest.numel())
        out = torch.zeros(n1 + n2, 128, dtype=torch.uint8, device=dev)
        ow, od = out.view(torch.int32).view(-1, 32), out.view(torch.float64).view(-1, 16)
        nan = float('nan')
        if n1:
            jj = j.clamp_min(0)
            has = j >= 0
            ow[:n1, 0] = w1[:, 0].int()
            ow[:n1, 1] = torch.where(has, w2[jj, 0], torch.full_like(jj, -1)).int() if i2.numel() else -1
            ow[:n1, 2], ow[:n1, 3] = w1[:, 1].int(), w1[:, 2].int()
            ow[:n1, 4], ow[:n1, 5], ow[:n1, 6], ow[:n1, 7] = w1[:, 8].int(), w1[:, 9].int(), w1[:, 10].int(), w1[:, 7].int()
            od[:n1, 4], od[:n1, 5] = nan, d1[:, 6]
            od[:n1, 6:12] = d1[:, 8:14]
            ow[:n1, 24] = w1[:, 3].int()
            ow[:n1, 25] = (torch.where(has, w2[jj, 3], torch.zeros_like(jj)).int() if i2.numel() else 0)
        if n2:
            r2, rd = w2[rest], d2[rest]
            ow[n1:, 0], ow[n1:, 1] = -1, r2[:, 0].int()
            ow[n1:, 2], ow[n1:, 3] = r2[:, 2].int(), r2[:, 1].int()
            ow[n1:, 4], ow[n1:, 7] = r2[:, 7].int(), r2[:, 8].int()
            od[n1:, 4], od[n1:, 5] = rd[:, 6], nan
            od[n1:, 6:12] = rd[:, 8:14]
            ow[n1:, 25] = r2[:, 3].int()
        first = 0
        if key_dev is not None and n1 + n2:      # sharded runs: records of factors that touch no boundary variable first
            # (a record's key: that of its edges -- both belong to one factor, and a factor's edges share their key)
            edge = torch.where(ow[:, 0] >= 0, ow[:, 0], ow[:, 1]).long()
            order = torch.sort(key_dev[edge], stable=True).indices
            out = out[order].contiguous()
            first = int((key_dev[edge] == 0).sum().item())
        self.pair_desc, self.n_pair = out, n1 + n2
        self.part_counts['pair'] = first

    def _struct(self):
        s = _abi.PbpStruct()
        s.n, s.T, s.flags = self.n, self.T, self.flags
        s.var_threshold, s.max_log_value = float(self.var_threshold), float(self.max_log_value)
        s.particles, s.old_particles = _abi.ptr(self.particles), _abi.ptr(self.old_particles)
        s.np, s.uniq, s.q = _abi.ptr(self.np_dev), _abi.ptr(self.uniq), _abi.ptr(self.q_dev)
        s.fast_edges, s.n_fast = _abi.ptr(self._fast_list), int(self.fast_edges.numel())
        s.generic_edges, s.n_generic = _abi.ptr(self._generic_list), int(self.generic_edges.numel())
        s.generic_pts_log2 = int(getattr(self, 'generic_pts_log2', 6))
        s.fast_desc = _abi.ptr(getattr(self, 'fast_desc', None))
        s.heavy_desc, s.n_heavy = _abi.ptr(getattr(self, 'heavy_desc', None)), int(getattr(self, 'n_heavy', 0))
        s.light_desc, s.n_light = _abi.ptr(getattr(self, 'light_desc', None)), int(getattr(self, 'n_light', 0))
        if getattr(self, 'n_small16', 0):
            s.small16_desc, s.n_small16 = _abi.ptr(self.small16_desc), int(self.n_small16)
        if getattr(self, 'n_small32', 0):
            s.small32_desc, s.n_small32 = _abi.ptr(self.small32_desc), int(self.n_small32)
        if self.paired_light and getattr(self, 'pair_desc', None) is not None:
            s.pair_desc, s.n_pair = _abi.ptr(self.pair_desc), int(self.n_pair)
        s.cq_desc, s.n_cq = _abi.ptr(getattr(self, 'cq_desc', None)), int(getattr(self, 'n_cq', 0))
        if getattr(self, 'v2f_lists', None) is not None:
            w, nw, nr, nn, hb, nh, m16, n16, m32, n32 = self.v2f_lists
            s.v2f_wide, s.n_v2f_wide, s.v2f_narrow, s.n_v2f_narrow = _abi.ptr(w), nw, _abi.ptr(nr), nn
            s.v2f_hub, s.n_v2f_hub = _abi.ptr(hb), nh
            s.v2f_mid16, s.n_v2f_mid16, s.v2f_mid32, s.n_v2f_mid32 = _abi.ptr(m16), n16, _abi.ptr(m32), n32
        s.f2v_ticket = _abi.ptr(self.f2v_ticket) if self.dynamic_f2v else None
        if self.listed_proposal and getattr(self, 'prop_desc', None) is not None:
            s.prop_desc, s.n_prop_desc = _abi.ptr(self.prop_desc), self.n_prop_desc
            if getattr(self, 'prop_hub', None) is not None:
                s.prop_hub, s.n_prop_hub, s.prop_partial = _abi.ptr(self.prop_hub), self.n_prop_hub, _abi.ptr(self.prop_partial)
        return s

    # ---- sampling ----------------------------------------------------------------------------
    def _host_draw(self):
        """the reference's generate_sample stream: NumPy global RNG, g.rvs order (EPBP:61-70)"""
        flat = self.flat
        q = self.q_dev.cpu().numpy()
        out = np.zeros((flat.V, self.n))
        rows = np.flatnonzero(flat.var_hidden & flat.var_cont)
        if rows.size:
            # one call for all rows: the legacy generator fills a (rows, n) array in order and keeps its spare Box-Muller value in
            # its state, so this is the stream of one standard_normal(n) per variable in g.rvs order
            z = np.random.standard_normal((rows.size, self.n))
            d = flat.var_dom[rows]
            out[rows] = np.clip(z * np.sqrt(q[rows, 1])[:, None] + q[rows, 0][:, None], flat.dom_lo[d][:, None], flat.dom_hi[d][:, None])
        return out

    def _install(self, host_particles):
        flat = self.flat
        p = np.nan_to_num(np.array(host_particles, dtype=np.float64), nan=0.0)
        disc = np.flatnonzero(flat.var_hidden & ~flat.var_cont)
        if disc.size:                                  # the particles of a discrete variable are its states
            d = flat.var_dom[disc]
            nst = (flat.dom_ptr[d + 1] - flat.dom_ptr[d]).astype(np.int64)
            k = np.arange(int(nst.max()))[None, :]
            live = k < nst[:, None]
            rows, cols = np.nonzero(live)
            p[disc[rows], cols] = flat.dom_val[flat.dom_ptr[d][rows] + cols]
        self.old_particles, self.particles = self.particles, self.old_particles
        self.particles.copy_(_abi.to_dev(p))

    def _generate_sample(self, rest_only=False):
        """`rest_only`: the fused kernel has already drawn for its variables into the other buffer (``_fused_sweep_head``): swap,
        then draw for the remaining listed variables only"""
        l, st = _abi.lib(), _abi.stream_ptr()
        k = self._draws
        self._draws += 1
        if self.sampler == 'device':
            self.old_particles, self.particles = self.particles, self.old_particles
            s = self._struct()
            gid = _abi.ptr(getattr(self, 'var_gid', None))
            if rest_only:
                F = self._fused
                if F['n_resample_rest']:
                    s.resample_vars, s.n_resample_vars = _abi.ptr(F['resample_rest']), int(F['n_resample_rest'])
                    _abi.check(l.lhvi_pbp_resample_uniq(self.dg.g, s, gid, int(self.seed), int(k), _abi.ptr(self.particles), _abi.ptr(self.uniq), st))
                self._views, self._batched = {}, {}
                return
            if self.listed_resample and self.n <= 64 and getattr(self, 'resample_vars', None) is not None:
                if not self._static_rows:
                    # first device draw of this state: the full (unlisted) draw into the CURRENT buffer only -- the other one may
                    # hold the particles the v -> f messages were evaluated at (a coarse-to-fine state, or one a host sampler
                    # filled) and must keep them; it only receives the rows no later listed draw writes: the states of the
                    # discrete variables and the rows of the observed ones
                    _abi.check(l.lhvi_pbp_resample_uniq(self.dg.g, s, gid, int(self.seed), int(k), _abi.ptr(self.particles), _abi.ptr(self.uniq), st))
                    if self._static_idx is not None:
                        self.old_particles.index_copy_(0, self._static_idx, self.particles.index_select(0, self._static_idx))
                    self._static_rows = True
                    self._views, self._batched = {}, {}
                    return
                s.resample_vars, s.n_resample_vars = _abi.ptr(self.resample_vars), int(self.resample_vars.shape[0])
            _abi.check(l.lhvi_pbp_resample_uniq(self.dg.g, s, gid, int(self.seed), int(k), _abi.ptr(self.particles), _abi.ptr(self.uniq), st))
            self._views, self._batched = {}, {}
            return
        elif callable(self.sampler):
            self._install(self.sampler(k, self.flat, self.q_dev.cpu().numpy()))
        else:
            self._install(self._host_draw())
        _abi.check(l.lhvi_pbp_uniq(self.dg.g, self.n, _abi.ptr(self.particles), _abi.ptr(self.np_dev),
                                   _abi.ptr(self.uniq), st))
        self._views, self._batched = {}, {}

    # ---- the sweep (EPBP.run EPBP:225-289; HybridLBP.run with c2f=-1 HLBP:430-536) -------------
    def _run_sweeps(self, iteration):
        l, st, g, p = _abi.lib(), _abi.stream_ptr(), self.dg.g, self.dg.p
        _abi.check(l.lhvi_pbp_init(g, self._struct(), _abi.ptr(self.eta), _abi.ptr(self.q_dev), _abi.ptr(self.f2v),
                                   _abi.ptr(self.v2f), st))
        self._generate_sample()
        for i in range(iteration):
            self.sweep(last=(i == iteration - 1))
        self._views, self._batched = {}, {}

    def sweep(self, last=False, f2v_events=None):
        """one flooding sweep: v2f, and unless `last`: proposal update, new sample, f2v.
        `f2v_events`: optional (start, end) torch.cuda.Event pair recorded around the f2v launch on its stream"""
        l, st, g, p = _abi.lib(), _abi.stream_ptr(), self.dg.g, self.dg.p
        if not last and getattr(self, '_fused', None) is not None and self._static_rows and self.sampler == 'device':
            # few particles: one pass per continuous variable does its v -> f messages, its proposal update and its new sample
            # (lhvi_pbp_var_fused, into the buffer the swap below makes current); the three kernels serve the other variables
            F = self._fused
            s = self._struct()
            n16, n32a, n32b = F['counts']
            _abi.check(l.lhvi_pbp_var_fused(g, s, _abi.ptr(self.f2v), _abi.ptr(self.v2f), _abi.ptr(self.eta), _abi.ptr(self.q_dev),
                                            _abi.ptr(getattr(self, 'var_gid', None)), int(self.seed), int(self._draws),
                                            _abi.ptr(self.old_particles), _abi.ptr(self.uniq), _abi.ptr(F['desc']), n16, n32a, n32b, st))
            w, nw, nr, nn, hb, nh, m16, c16, m32, c32 = F['v2f_rest']
            s.v2f_wide, s.n_v2f_wide, s.v2f_narrow, s.n_v2f_narrow = _abi.ptr(w), nw, _abi.ptr(nr), nn
            s.v2f_hub, s.n_v2f_hub = _abi.ptr(hb), nh
            s.v2f_mid16, s.n_v2f_mid16, s.v2f_mid32, s.n_v2f_mid32 = _abi.ptr(m16), c16, _abi.ptr(m32), c32
            _abi.check(l.lhvi_pbp_v2f(g, s, _abi.ptr(self.f2v), _abi.ptr(self.v2f), st))
            s.prop_desc, s.n_prop_desc = _abi.ptr(F['prop_desc_rest']), int(F['n_prop_rest'])
            _abi.check(l.lhvi_pbp_proposal(g, s, _abi.ptr(self.f2v), _abi.ptr(self.eta), _abi.ptr(self.q_dev), st))
            self._generate_sample(rest_only=True)
            self._launch_f2v(self._struct(), f2v_events)
            return
        _abi.check(l.lhvi_pbp_v2f(g, self._struct(), _abi.ptr(self.f2v), _abi.ptr(self.v2f), st))
        if not last:
            _abi.check(l.lhvi_pbp_proposal(g, self._struct(), _abi.ptr(self.f2v), _abi.ptr(self.eta),
                                           _abi.ptr(self.q_dev), st))
            self._generate_sample()
            self._launch_f2v(self._struct(), f2v_events)

    def _launch_f2v(self, s, f2v_events=None):
        """the f -> v half sweep (`lhvi_pbp_f2v`).  `f2v_events`: (start, end) events recorded around the heavy kernel: the
        call is then split with the SKIP flags into three, one kernel each, on the same stream."""
        l, g, p, st = _abi.lib(), self.dg.g, self.dg.p, _abi.stream_ptr()
        if not getattr(self, '_has_f_side', True):
            raise _abi.LhviError('this state was set up for the v -> f half only (sides="v")')
        args = (_abi.ptr(self.v2f), _abi.ptr(self.f2v))
        want = os.environ.get('LHVI_PBP_OVERLAP')                     # (tuning aid: scripts/diag/f2v_overlap.sh)
        if (self.overlap_f2v or want == '1') and want != '0' and self.n_heavy >= self.overlap_min_heavy:
            # the long kernel(s) of the half sweep on this stream, one workgroup per CU short of a full device; the short kernels
            # (pair / light / cq / fast, then generic: other rows of f2v) on a second stream beside them.  The heavy kernel is bound
            # by VALU issue and the LDS and indifferent to 6 or 7 waves per SIMD; the pair kernel waits for its loads most of its
            # 1.1 ms -- side by side the second one costs next to nothing.
            torch = _abi.require_gpu()
            main = torch.cuda.current_stream()
            if getattr(self, '_side', None) is None:
                self._side = torch.cuda.Stream(device=self.dg.device)
                self._fork, self._join = torch.cuda.Event(), torch.cuda.Event()
            base = s.flags
            self._fork.record(main)
            self._side.wait_event(self._fork)
            with torch.cuda.stream(self._side):
                sst = _abi.stream_ptr()
                s.flags = base | _abi.PBP_SKIP_GENERIC | _abi.PBP_SKIP_HEAVY
                _abi.check(l.lhvi_pbp_f2v(g, p, s, *args, sst))
                s.flags = base | _abi.PBP_SKIP_FAST
                _abi.check(l.lhvi_pbp_f2v(g, p, s, *args, sst))
                self._join.record(self._side)
            s.flags = base | _abi.PBP_SKIP_GENERIC | _abi.PBP_SKIP_LIGHT | _abi.PBP_SHARE_CUS
            if f2v_events:
                f2v_events[0].record()
            _abi.check(l.lhvi_pbp_f2v(g, p, s, *args, st))
            if f2v_events:
                f2v_events[1].record()
            main.wait_event(self._join)
            s.flags = base
            return
        if f2v_events:
            base = s.flags
            s.flags = base | _abi.PBP_SKIP_GENERIC | _abi.PBP_SKIP_LIGHT
            f2v_events[0].record()
            _abi.check(l.lhvi_pbp_f2v(g, p, s, *args, st))
            f2v_events[1].record()
            s.flags = base | _abi.PBP_SKIP_GENERIC | _abi.PBP_SKIP_HEAVY
            _abi.check(l.lhvi_pbp_f2v(g, p, s, *args, st))
            s.flags = base | _abi.PBP_SKIP_FAST
            _abi.check(l.lhvi_pbp_f2v(g, p, s, *args, st))
            s.flags = base
        else:
            _abi.check(l.lhvi_pbp_f2v(g, p, s, *args, st))

    # ---- dict views with the reference's keys ------------------------------------------------------
    def _host(self, name):
        if name not in self._views:
            self._views[name] = getattr(self, name).cpu().numpy()
        return self._views[name]

    @property
    def sample(self):
        flat, P = self.flat, self._host('particles')
        out = {}
        for v, rv in enumerate(flat.rvs):
            if flat.var_hidden[v]:
                out[rv] = P[v].copy() if flat.var_cont[v] else rv.domain.values
        return out

    @property
    def q(self):
        flat, Q = self.flat, self._host('q_dev')
        return {rv: (float(Q[v, 0]), float(Q[v, 1])) for v, rv in enumerate(flat.rvs)
                if flat.var_hidden[v] and (flat.var_cont[v] or self._epbp_discrete)}

    @property
    def eta_message(self):
        flat, E = self.flat, self._host('eta')
        out = {}
        for k in range(flat.var_edge.size):
            e = int(flat.var_edge[k])
            v = flat.edge_var[e]
            if flat.var_hidden[v] and (flat.var_cont[v] or self._epbp_discrete):
                out[(flat.factors[flat.edge_fac[e]], flat.rvs[v])] = (float(E[e, 0]), float(E[e, 1]))
        return out

    @property
    def message(self):
        """log messages as dicts keyed by point, like the reference (built on demand; large graphs: use .f2v/.v2f)"""
        flat, n = self.flat, self.n
        F2V, V2F, P = self._host('f2v'), self._host('v2f'), self._host('particles')
        out = {}
        for k in range(flat.var_edge.size):
            e = int(flat.var_edge[k])
            v = flat.edge_var[e]
            if not flat.var_hidden[v]:
                continue
            rv, f = flat.rvs[v], flat.factors[flat.edge_fac[e]]
            npv = self.np_host[v]
            pts = [float(x) for x in P[v, :npv]]
            m = {x: float(F2V[e, j]) for j, x in enumerate(pts)}
            if flat.var_cont[v]:
                d = flat.var_dom[v]
                grid = flat.dom_val[flat.dom_ptr[d]:flat.dom_ptr[d + 1]]
                m.update({float(x): float(F2V[e, n + t]) for t, x in enumerate(grid)})
            out[(f, rv)] = m
            out[(rv, f)] = {x: float(V2F[e, j]) for j, x in enumerate(pts)}
        return out

    # ---- queries (A8) ------------------------------------------------------------------------
    def _belief_rv_points(self, v, xs):
        """belief_rv(x) for one variable index at a batch of points, on the device (EPBP:196-202)"""
        torch = _abi.require_gpu()
        xs = np.atleast_1d(np.asarray(xs, dtype=np.float64))
        qvar = _abi.to_dev(np.array([v], dtype=np.int32))
        x = _abi.to_dev(xs.reshape(1, -1))
        out = torch.empty_like(x)
        _abi.check(_abi.lib().lhvi_pbp_belief_points(self.dg.g, self.dg.p, self._struct(), _abi.ptr(self.v2f), 1,
                                                     _abi.ptr(qvar), int(xs.size), _abi.ptr(x), _abi.ptr(out),
                                                     _abi.stream_ptr()))
        return out.cpu().numpy().reshape(-1)

    def belief_rv_batch(self, rvs, xs):
        """log-beliefs of many variables at `xs[i]` points each, one launch (extension, not in the reference)"""
        torch = _abi.require_gpu()
        idx = np.array([self._var_of(rv) for rv in rvs], dtype=np.int32)
        x = _abi.to_dev(np.asarray(xs, dtype=np.float64).reshape(len(rvs), -1))
        out = torch.empty_like(x)
        _abi.check(_abi.lib().lhvi_pbp_belief_points(self.dg.g, self.dg.p, self._struct(), _abi.ptr(self.v2f),
                                                     len(rvs), _abi.ptr(_abi.to_dev(idx)), int(x.shape[1]),
                                                     _abi.ptr(x), _abi.ptr(out), _abi.stream_ptr()))
        return out.cpu().numpy()

    # ---- batched queries: every variable at once (extension) ------------------------
    def belief_rv_all(self, x):
        """log-beliefs ``belief_rv`` (EPBP:196-202) of EVERY variable at ``x[v, :]`` (n points per variable; a device
        tensor or array of shape (V, n)): one f2v launch with the query points in the place of the target particles
        tabulates all messages, one pass adds them up per variable (count-weighted on a lifted graph with a stable
        partition: rows are clusters then).  Returns a (V, n) device tensor; rows of observed variables are 0."""
        torch = _abi.require_gpu()
        if self.flat.lifted and not getattr(self, '_stable_partition', False):
            # a lifted query walks the GROUND variable's factors (HLBP:313-317); that equals the count-weighted sum over
            # the cluster's edges only when the partition is stable (run with c2f = -1)
            raise NotImplementedError('batched queries on a lifted graph need a stable partition (run(c2f=-1))')
        l, st = _abi.lib(), _abi.stream_ptr()
        xq = x if torch.is_tensor(x) else _abi.to_dev(np.ascontiguousarray(x, dtype=np.float64))
        assert tuple(xq.shape) == (self.flat.V, self.n) and xq.is_contiguous()
        if getattr(self, '_query_f2v', None) is None or self._query_f2v.shape != self.f2v.shape:
            self._query_f2v = torch.empty_like(self.f2v)
        s = self._struct()
        s.particles, s.old_particles = _abi.ptr(xq), _abi.ptr(self.particles)      # partners: the current sample
        _abi.check(l.lhvi_pbp_f2v(self.dg.g, self.dg.p, s, _abi.ptr(self.v2f), _abi.ptr(self._query_f2v), st))
        out = torch.empty_like(xq)
        _abi.check(l.lhvi_pbp_var_sum(self.dg.g, s, _abi.ptr(self._query_f2v), _abi.ptr(out), st))
        return out

    def map_all(self, steps=5, scan=64):
        """MAP of every variable at once: a uniform scan of the domain with at least `scan` points (ceil(scan / n) passes
        of n points, so a solver with few particles still sees a narrow mode), then ``steps - 1`` times a new n-point grid
        on the bracket around the best point (the bracket shrinks by (n-1)/2 per step; 5 steps with n = 64 resolve 1e-6 of
        the domain width, the reference's fminbound stops at 1e-5).  Finds the mode the scan sees, where ``EPBP.map``
        (EPBP:377-394) finds the one fminbound's golden section runs into.  Returns (map, log-belief) as arrays of length
        V; observed variables return their value."""
        torch = _abi.require_gpu()
        l, st = _abi.lib(), _abi.stream_ptr()
        n, flat = self.n, self.flat
        x = torch.empty(flat.V, n, dtype=torch.float64, device=self.particles.device)
        best = torch.empty(flat.V, dtype=torch.float64, device=x.device)
        val = torch.empty_like(best)
        s = self._struct()
        _abi.check(l.lhvi_pbp_domain_grid(self.dg.g, s, _abi.ptr(x), st))
        passes = -(-int(scan) // n) if n >= 3 else 1
        if passes > 1 and bool((flat.var_hidden & flat.var_cont).any()):
            cont = _abi.to_dev(flat.var_hidden & flat.var_cont)[:, None]
            lo = _abi.to_dev(np.where(flat.var_cont, flat.dom_lo[flat.var_dom], 0.0))[:, None]
            hi = _abi.to_dev(np.where(flat.var_cont, flat.dom_hi[flat.var_dom], 1.0))[:, None]
            h = (hi - lo) / (passes * (n - 1))
            j = torch.arange(n, dtype=torch.float64, device=x.device)[None, :]
            bx = bv = None
            for p in range(passes):                  # pass p covers [lo + p (n-1) h, lo + (p+1)(n-1) h], ends shared
                xp = torch.where(cont, lo + (p * (n - 1) + j) * h, x)
                vp, ap = self.belief_rv_all(xp).max(dim=1)
                xb = xp.gather(1, ap[:, None])[:, 0]
                if bx is None:
                    bx, bv = xb, vp
                else:
                    better = vp > bv                 # first maximum, like the reference's argmax over a list
                    bx, bv = torch.where(better, xb, bx), torch.where(better, vp, bv)
            a, b = torch.maximum(bx[:, None] - h, lo), torch.minimum(bx[:, None] + h, hi)
            x = torch.where(cont, a + (b - a) * (j / (n - 1)), x).contiguous()
        for _ in range(max(int(steps), 1)):
            logb = self.belief_rv_all(x)
            _abi.check(l.lhvi_pbp_refine_grid(self.dg.g, s, _abi.ptr(logb), _abi.ptr(x), _abi.ptr(best), _abi.ptr(val), st))
        return best.cpu().numpy(), val.cpu().numpy()

    def _brent(self, row_var, pairs=None, xtol=1e-5, maxfun=500):
        """``lhvi_pbp_map_brent`` on the rows `row_var` (variables of the solver's graph; with `pairs` = (ptr, edge, mult) the rows'
        beliefs are sums over explicit (edge, multiplicity) lists).  Returns device tensors (x, log-belief, evaluations)."""
        torch = _abi.require_gpu()
        rv_t = row_var if torch.is_tensor(row_var) else _abi.to_dev(np.ascontiguousarray(row_var, dtype=np.int32))
        rv_t = rv_t.to(torch.int32).contiguous()
        nq = int(rv_t.numel())
        x = torch.empty(nq, dtype=torch.float64, device=rv_t.device)
        fx = torch.empty_like(x)
        nfev = torch.empty(nq, dtype=torch.int32, device=rv_t.device)
        qptr = qedge = qmult = None
        if pairs is not None:
            qptr, qedge, qmult = (t.contiguous() for t in pairs)
            assert qptr.dtype == torch.int64 and qedge.dtype == torch.int32 and qmult.dtype == torch.float64 and qptr.numel() == nq + 1
        _abi.check(_abi.lib().lhvi_pbp_map_brent(self.dg.g, self.dg.p, self._struct(), _abi.ptr(self.v2f), nq, _abi.ptr(rv_t),
                                                 _abi.ptr(qptr), _abi.ptr(qedge), _abi.ptr(qmult), float(xtol), int(maxfun),
                                                 _abi.ptr(x), _abi.ptr(fx), _abi.ptr(nfev), _abi.stream_ptr()))
        return x, fx, nfev

    def map_fminbound_all(self, xtol=1e-5, maxfun=500):
        """MAP of every variable of the solver's graph the way the reference finds it (EPBP:377-394, HLBP:405-424): every
        continuous hidden variable runs ``scipy.optimize.fminbound``'s iteration on ``-belief_rv`` over its domain -- all of
        them in ONE launch, a thread each (``lhvi_pbp_map_brent``) -- and every discrete one takes its first state with the
        largest belief.  Returns (map [V], log-belief [V], evaluations [V]) as host arrays; observed rows hold their value."""
        flat = self.flat
        if flat.lifted and not getattr(self, '_stable_partition', False):
            raise NotImplementedError('batched queries on a lifted graph need a stable partition (run(c2f=-1))')
        hid = np.flatnonzero(flat.var_hidden).astype(np.int32)
        out = np.nan_to_num(np.array(flat.var_value, dtype=np.float64), nan=0.0)
        val, nfev = np.zeros(flat.V), np.zeros(flat.V, dtype=np.int32)
        if hid.size:
            x, fx, k = self._brent(hid, xtol=xtol, maxfun=maxfun)
            out[hid], val[hid], nfev[hid] = x.cpu().numpy(), fx.cpu().numpy(), k.cpu().numpy()
        return out, val, nfev

    def quad_all(self, margin=20.0, epsabs=1.49e-8, epsrel=1.49e-8):
        """The normaliser of ``EPBP.belief`` (EPBP:325-328: ``quad`` of ``e ** belief_rv`` over the domain widened by 20 on both
        sides) of every continuous hidden variable in ONE launch (``lhvi_pbp_quad``: QUADPACK's 21-point Gauss-Kronrod rule in its
        adaptive bisection, to scipy's default tolerances; agrees with ``scipy.integrate.quad`` to ~1e-7 relative).  Returns host
        arrays (z [V], status [V]); z is NaN where the integrand overflowed (the reference raises OverflowError there) and for
        rows that are not continuous hidden variables."""
        torch = _abi.require_gpu()
        flat = self.flat
        if flat.lifted and not getattr(self, '_stable_partition', False):
            raise NotImplementedError('batched queries on a lifted graph need a stable partition (run(c2f=-1))')
        rows = np.flatnonzero(flat.var_hidden & flat.var_cont).astype(np.int32)
        z, st = np.full(flat.V, np.nan), np.zeros(flat.V, dtype=np.int32)
        if rows.size:
            dom = flat.var_dom[rows]
            rv_t = _abi.to_dev(rows)
            lo, hi = _abi.to_dev(flat.dom_lo[dom] - margin), _abi.to_dev(flat.dom_hi[dom] + margin)
            zt = torch.empty(rows.size, dtype=torch.float64, device=rv_t.device)
            et = torch.empty_like(zt)
            stt = torch.empty(rows.size, dtype=torch.int32, device=rv_t.device)
            _abi.check(_abi.lib().lhvi_pbp_quad(self.dg.g, self.dg.p, self._struct(), _abi.ptr(self.v2f), int(rows.size), _abi.ptr(rv_t),
                                                None, None, None, _abi.ptr(lo), _abi.ptr(hi), float(epsabs), float(epsrel),
                                                _abi.ptr(zt), _abi.ptr(et), _abi.ptr(stt), _abi.stream_ptr()))
            z[rows], st[rows] = zt.cpu().numpy(), stt.cpu().numpy()
        return z, st

    def _per_var(self, a):
        torch = _abi.require_gpu()
        t = torch.as_tensor(np.broadcast_to(np.asarray(a, dtype=np.float64), (self.flat.V,)).copy()) if not torch.is_tensor(a) else a
        return t.to(self.particles.device, torch.float64)

    def log_area_all(self, a, b, npts=20, shift=None):
        """``log_area`` (EPBP:291-308, HLBP:318-341) of every continuous hidden variable at once: trapezoid of
        exp(belief_rv - shift) on linspace(a[v], b[v], npts), shift = ``log_message_balance`` of the tabulated values
        unless given.  `a`, `b`: scalars or arrays / tensors of length V.  Returns device tensors
        (area [V], shift [V]); rows of discrete and observed variables are NaN."""
        torch = _abi.require_gpu()
        if npts < 2:
            raise ValueError('npts must be at least 2')
        a, b = self._per_var(a), self._per_var(b)
        n = self.n
        x = torch.empty(self.flat.V, n, dtype=torch.float64, device=a.device)
        _abi.check(_abi.lib().lhvi_pbp_domain_grid(self.dg.g, self._struct(), _abi.ptr(x), _abi.stream_ptr()))
        cont = _abi.to_dev(self.flat.var_hidden & self.flat.var_cont)
        step = (b - a) / (npts - 1)                                   # numpy.linspace: start + i * step, last point = stop
        lin = torch.arange(npts, dtype=torch.float64, device=a.device)[None, :] * step[:, None] + a[:, None]
        lin[:, npts - 1] = b
        y = torch.empty_like(lin)
        for c in range(0, npts, n):                                   # n query points per variable and pass
            w = min(n, npts - c)
            x[:, :w] = torch.where(cont[:, None], lin[:, c:c + w], x[:, :w])
            y[:, c:c + w] = self.belief_rv_all(x)[:, :w]
        if shift is None:
            mean, mx = y.mean(dim=1), y.max(dim=1).values
            shift = torch.where(mx - mean > self.max_log_value, mx - self.max_log_value, mean)
        w = torch.exp(y - shift[:, None])
        area = (w[:, :-1] + w[:, 1:]).sum(dim=1) * (lin[:, 1] - lin[:, 0]) * 0.5
        nan = torch.full_like(area, float('nan'))
        return torch.where(cont, area, nan), torch.where(cont, shift, nan)

    def belief_all(self, x):
        """normalised beliefs of EVERY variable (``HybridLBP.belief`` HLBP:343-382 for all of them at once; for EPBP the
        same 20-point trapezoid normaliser in the place of ``scipy.integrate.quad``).  `x`: (V, m) query points, m <= n.
        Continuous hidden rows: exp(belief_rv(x) - shift) / area over the domain; discrete hidden rows: the normalised
        beliefs of the variable's states in columns [0, #states) (x is ignored there); observed rows: 1 where x equals
        the value.  Returns a (V, m) device tensor."""
        torch = _abi.require_gpu()
        flat = self.flat
        xq = x if torch.is_tensor(x) else _abi.to_dev(np.ascontiguousarray(x, dtype=np.float64))
        xq = xq.reshape(flat.V, -1)
        m = xq.shape[1]
        if m > self.n:
            raise ValueError('at most n query points per variable')
        lo = np.where(flat.var_cont, flat.dom_lo[flat.var_dom], 0.0)
        hi = np.where(flat.var_cont, flat.dom_hi[flat.var_dom], 1.0)
        z, shift = self.log_area_all(lo, hi, 20)
        grid = torch.empty(flat.V, self.n, dtype=torch.float64, device=xq.device)
        _abi.check(_abi.lib().lhvi_pbp_domain_grid(self.dg.g, self._struct(), _abi.ptr(grid), _abi.stream_ptr()))
        cont = _abi.to_dev(flat.var_hidden & flat.var_cont)
        grid[:, :m] = torch.where(cont[:, None], xq, grid[:, :m])     # discrete rows keep their states
        logb = self.belief_rv_all(grid)
        out = torch.exp(logb[:, :m] - shift[:, None] - torch.log(z)[:, None])
        disc = _abi.to_dev(flat.var_hidden & ~flat.var_cont)
        if bool(disc.any()):
            nst = _abi.to_dev(self.np_host.astype(np.int64))
            live = torch.arange(self.n, device=xq.device)[None, :] < nst[:, None]
            w = torch.where(live, torch.exp(logb), torch.zeros_like(logb))
            w = w / w.sum(dim=1, keepdim=True)
            out = torch.where(disc[:, None], w[:, :m], out)
        obs = _abi.to_dev(~flat.var_hidden)
        val = _abi.to_dev(np.nan_to_num(flat.var_value, nan=0.0))
        return torch.where(obs[:, None], (xq == val[:, None]).to(torch.float64), out)

    def probability_all(self, a, b):
        """``probability(a, b, rv)`` (EPBP:356-375, HLBP:384-403) of every continuous hidden variable at once: 5-point
        trapezoid on [a[v], b[v]] over the 20-point one on the domain.  Returns a device tensor [V] (NaN elsewhere)."""
        flat = self.flat
        lo = np.where(flat.var_cont, flat.dom_lo[flat.var_dom], 0.0)
        hi = np.where(flat.var_cont, flat.dom_hi[flat.var_dom], 1.0)
        z, shift = self.log_area_all(lo, hi, 20)
        num, _ = self.log_area_all(a, b, 5, shift=shift)
        return num / z

    # ---- per-variable queries answered from one batched pass (the reference's callers loop `infer.map(rv)` over every rv:
    # Demo/RGM/demo.py:32-35, Demo/HMLN/DemoPaperPopularity.py:44-47) --------------------------------------------------------
    def _query_cache(self, name, fill):
        c = self.__dict__.setdefault('_batched', {})
        if name not in c:
            c[name] = fill()
        return c[name]

    def _cached_map(self):
        """MAP of every row of the solver's graph as a host array: the reference's ``fminbound`` iterates for all rows in one
        launch (``map_fminbound_all``); with ``map_mode = 'global'`` the scan + bracket refinement of ``map_all`` (the highest
        mode a 64-point scan sees, which need not be the one Brent's golden section runs into)"""
        if self.map_mode == 'global':
            return self._query_cache('map_global', lambda: self.map_all(steps=6 if self.n >= 32 else 9)[0])
        return self._query_cache('map', lambda: self.map_fminbound_all()[0])

    def _cached_area(self):
        """(z, shift) of ``log_area`` over the domain with 20 points for every row (``log_area_all``), host arrays"""
        def fill():
            flat = self.flat
            lo = np.where(flat.var_cont, flat.dom_lo[flat.var_dom], 0.0)
            hi = np.where(flat.var_cont, flat.dom_hi[flat.var_dom], 1.0)
            z, shift = self.log_area_all(lo, hi, 20)
            return z.cpu().numpy(), shift.cpu().numpy()
        return self._query_cache('area', fill)

    def log_message_balance(self, message):
        """EPBP.log_message_balance (EPBP:204-215) on a host dict (used by log_area)"""
        values = list(message.values())
        mean_m = float(np.mean(values))
        max_m = max(values)
        shift = max_m - self.max_log_value if max_m - mean_m > self.max_log_value else mean_m
        for k in message:
            message[k] = message[k] - shift
        return shift

    def log_area(self, f, a, b, n, shift=None):
        """``log_area`` with the reference's signature (EPBP:291-308, HLBP:324-341): trapezoid of exp(f(x) - shift) on
        linspace(a, b, n) for a callable log-density ``f``; returns (area, shift)"""
        x = np.linspace(a, b, n)
        d = x[1] - x[0]
        y = {i: f(v) for i, v in enumerate(x)}
        if shift is None:
            shift = self.log_message_balance(y)
        else:
            y = {k: val - shift for k, val in y.items()}
        res, prev = 0, e ** y[0]
        for i in range(1, n):
            cur = e ** y[i]
            res += (prev + cur) * d
            prev = cur
        return res * 0.5, shift

    @staticmethod
    def get_cluster(instance):
        return instance.cluster

    def _log_area(self, v, a, b, npts, shift=None):
        """EPBP.log_area (EPBP:291-308): trapezoid of exp(belief_rv - shift) on linspace(a, b, npts)"""
        x = np.linspace(a, b, npts)
        d = x[1] - x[0]
        y = dict(enumerate(self._belief_rv_points(v, x).tolist()))
        if shift is None:
            shift = self.log_message_balance(y)
        else:
            y = {k: val - shift for k, val in y.items()}
        res, prev = 0, e ** y[0]
        for i in range(1, npts):
            cur = e ** y[i]
            res += (prev + cur) * d
            prev = cur
        return res * 0.5, shift

    @staticmethod
    def message_normalization(message):
        z = 0
        for k, v in message.items():
            z = z + v
        for k, v in message.items():
            message[k] = v / z

    @staticmethod
    def norm_pdf(x, mu, sig):
        u = (x - mu) / sig
        return np.exp(-u * u * 0.5) / (2.506628274631 * sig)


class EPBP(_ParticleSweep):
    """Expectation particle BP on a ground graph (``EPBPLogVersion.py``)."""

    var_threshold = 3
    max_log_value = 700
    _epbp_discrete = True

    def __init__(self, g=None, n=50, proposal_approximation='EP', sampler='host', seed=0):
        self.g = g
        self.n = n
        self.proposal_approximation = proposal_approximation
        self.sampler, self.seed = sampler, seed
        self.cache = dict()

    def run(self, iteration=10, log_enable=False):
        self._setup(self.g)
        self.cache = dict()
        self._run_sweeps(iteration)

    def _var_of(self, rv):
        return self.flat.var_index[rv]

    def belief_rv(self, x, rv, sample=None):
        return float(self._belief_rv_points(self._var_of(rv), [x])[0])

    def belief(self, x, rv, log_belief=False, inf_integral=False):
        """EPBP.belief (EPBP:310-354): normaliser by adaptive quadrature of exp(belief_rv)"""
        if rv.value is not None:
            if log_belief:
                return 0 if x == rv.value else -np.inf
            return 1 if x == rv.value else 0
        v = self._var_of(rv)
        if rv.domain.continuous:
            if rv in self.cache:
                z, shift = self.cache[rv]
            elif not self.exact_queries and not inf_integral:
                # the normalisers of all variables from one launch (quad_all), made at the first call after run()
                zs, status = self._query_cache('quad', self.quad_all)
                if status[v] == 3:
                    raise OverflowError('(34, \'Numerical result out of range\')')      # e ** belief_rv overflowed, as in the reference (EPBP:326)
                z, shift = float(zs[v]), 0
                self.cache[rv] = (z, shift)
            else:
                from scipy.integrate import quad
                lb, ub = (-np.inf, np.inf) if inf_integral else (rv.domain.values[0] - 20, rv.domain.values[1] + 20)
                z = quad(lambda val: e ** float(self._belief_rv_points(v, [val])[0]), lb, ub)[0]
                shift = 0
                self.cache[rv] = (z, shift)
            logz = log(z)
            lb_x = float(self._belief_rv_points(v, [x])[0]) - shift - logz
            return lb_x if log_belief else e ** lb_x
        vals = rv.domain.values
        b = dict(zip(vals, (e ** t for t in self._belief_rv_points(v, vals).tolist())))
        self.message_normalization(b)
        return log(b[x]) if log_belief else b[x]

    def probability(self, a, b, rv):
        if rv.value is None and rv.domain.continuous:
            v = self._var_of(rv)
            if self.exact_queries:
                z, shift = self._log_area(v, rv.domain.values[0], rv.domain.values[1], 20)
            else:                               # the normaliser of every variable from one batched pass (same 20-point trapezoid)
                zs, shifts = self._cached_area()
                z, shift = float(zs[v]), float(shifts[v])
            num, _ = self._log_area(v, a, b, 5, shift)
            return num / z
        return None

    def map(self, rv):
        if rv.value is not None:
            return rv.value
        v = self._var_of(rv)
        if not self.exact_queries:
            m = self._cached_map()[v]
            return float(m) if rv.domain.continuous else type(rv.domain.values[0])(m)
        if rv.domain.continuous:
            from scipy.optimize import fminbound
            return fminbound(lambda val: -float(self._belief_rv_points(v, [val])[0]),
                             rv.domain.values[0], rv.domain.values[1], disp=False)
        pts = list(self.sample[rv])
        vals = self._belief_rv_points(v, pts)
        return pts[int(np.argmax(vals))]


class HybridLBP(_ParticleSweep):
    """Lifted particle BP (``HybridLBPLogVersion.py``): colour passing, then the counted sweep."""

    var_threshold = 5
    max_log_value = 700
    _epbp_discrete = False

    def __init__(self, g, n=50, k_mean_k=2, k_mean_iteration=10, proposal_approximation='EP', sampler='host', seed=0):
        from .lifting import CompressedGraph
        self.g = CompressedGraph(g)
        self.n = n
        self.k_mean_k = k_mean_k
        self.k_mean_iteration = k_mean_iteration
        self.proposal_approximation = proposal_approximation
        self.sampler, self.seed = sampler, seed
        self.query_cache = dict()

    c2f_on_objects = False          # run(c2f >= 0) through Python objects per cluster (lhvi.c2f.run_c2f) instead of arrays

    @classmethod
    def on_flat(cls, flat, n=50, k_mean_k=2, k_mean_iteration=10, proposal_approximation='EP', sampler='device', seed=0):
        """The lifted sweep on arrays -- no Python object per ground atom anywhere.  `flat` is either a ready-made lifted
        ``FlatGraph`` (``lifting.lift_flat`` of a stable partition) or a GROUND one (``RelationalGraph.ground_flat``,
        ``flatten(g)``, ``build_flat``): then ``run_flat`` does the lifting itself, to the stable partition (``c2f=-1``) or coarse
        to fine (``c2f >= 0``).  Query with the batched calls (``belief_rv_all``, ``map_all``, ``belief_all``,
        ``probability_all``: rows are clusters, stable partition) or per ground variable with ``belief_rv_ground``."""
        self = cls.__new__(cls)
        self.g = None
        self.n, self.k_mean_k, self.k_mean_iteration = n, k_mean_k, k_mean_iteration
        self.proposal_approximation, self.sampler, self.seed = proposal_approximation, sampler, seed
        self.query_cache = dict()
        self._flat_in = flat
        return self

    def run_flat(self, iteration=10, c2f=-1, timing=None):
        """``run(iteration, c2f)`` for a solver made by ``on_flat``.  ``timing``: dict that receives the per-sweep seconds of a
        coarse-to-fine run (``lhvi.c2f.run_c2f_flat``)."""
        self.query_cache = dict()
        flat = self._flat_in
        if flat.lifted:
            if c2f != -1:
                raise ValueError('a coarse-to-fine run starts from the GROUND graph (on_flat(ground_flat))')
            self._setup(None, flat=flat)
            self._run_sweeps(iteration)
            self._stable_partition = True
            return
        from .lifting import initial_colors_device, lift_flat, refine_flat
        dg = _abi.DeviceGraph(flat)
        if c2f == -1:
            rvc0, fc0, sym = initial_colors_device(flat, dg, True)
            rvc, fc = refine_flat(flat, sym, rvc0, fc0, dg=dg, device_out=True)
            self._ground = dict(flat=flat, dg=dg, rvc=rvc, fc=fc)
            self._setup(None, flat=lift_flat(flat, rvc, fc, dg=dg))
            self._run_sweeps(iteration)
            self._stable_partition = True
            return
        self._run_c2f_arrays(flat, dg, iteration, c2f, timing, keep_history=getattr(self, 'c2f_keep_history', False))

    def _c2f_draw(self):
        def draw(k, flat, q_host):
            if callable(self.sampler):
                return self.sampler(k, flat, q_host)
            if self.sampler == 'device':
                return None                       # engine.install draws on the device
            out = np.zeros((flat.V, self.n))      # the reference's stream: standard_normal in cluster order (HLBP:75-87)
            rows = np.flatnonzero(flat.var_hidden & flat.var_cont)
            if rows.size:                         # (one call: the same stream as one standard_normal(n) per cluster, see _host_draw)
                z = np.random.standard_normal((rows.size, self.n))
                dmn = flat.var_dom[rows]
                out[rows] = np.clip(z * np.sqrt(q_host[rows, 1])[:, None] + q_host[rows, 0][:, None], flat.dom_lo[dmn][:, None],
                                    flat.dom_hi[dmn][:, None])
            return out
        return draw

    def _run_c2f_arrays(self, gflat, dg, iteration, c2f, timing=None, keep_history=False):
        """``run(iteration, c2f >= 0)`` on arrays (``lhvi.c2f.run_c2f_flat``): colours, refinement and both re-liftings of every
        sweep on the device"""
        from . import c2f as _c2f
        from .lifting import initial_colors_device
        rvc0, fc0, sym = initial_colors_device(gflat, dg, False)                     # HLBP:432
        observer = getattr(self, 'c2f_observer', None)
        st, G2, rvc, fc, history = _c2f.run_c2f_flat(
            gflat, dg, _DeviceEngine(self), _c2f.FlatRefiner(gflat, dg, sym), iteration, c2f, self.k_mean_k, self.k_mean_iteration,
            self._c2f_draw(), rvc0, fc0, observer=observer, keep_history=keep_history, timing=timing)
        self.c2f_history = history
        self._stable_partition = False
        self.__dict__.update({k: v for k, v in st.__dict__.items()})
        self._ground = dict(flat=gflat, dg=dg, rvc=rvc, fc=fc)
        self._views, self._batched = {}, {}

    def run(self, iteration=10, log_enable=False, c2f=-1):
        """``HybridLBP.run`` (HLBP:430-536).  ``c2f == -1``: colour passing to the stable partition, then the sweeps.
        ``c2f >= 0``: coarse start (continuous evidence merged regardless of value), evidence clusters are split by
        k-means while their variance exceeds a shrinking threshold, rv / factor clusters are refined once per sweep and
        every new cluster inherits the messages, sites, proposal and particles of the cluster it came from."""
        self.query_cache = dict()
        self._stable_partition = False
        self._ground = None
        if c2f == -1:
            self.g.init_cluster(True)
            prev = -1
            while self.g.num_rv_clusters != prev:         # HLBP:433-438
                prev = self.g.num_rv_clusters
                self.g.split_factors()
                self.g.split_rvs()
            self.g.array_flat = True                        # (stable: lifting.lifted_flat instead of walking the cluster objects)
            self._setup(self.g)
            self._run_sweeps(iteration)
            self.g.split_factors()                          # HLBP:536 (a no-op on a stable partition)
            self._stable_partition = True
            return
        from . import c2f as _c2f
        ground = self.g.g
        if not self.c2f_on_objects:
            # the ground objects are flattened once; clusters become objects again only at the end, for the rv.cluster queries
            from .lifting import CompressedGraph
            gflat = flatten(ground, require_device_potentials=True)
            self._run_c2f_arrays(gflat, _abi.DeviceGraph(gflat), iteration, c2f, keep_history=True)
            cg = CompressedGraph(ground)
            cg.set_colors(self._ground['rvc'].cpu().numpy(), self._ground['fc'].cpu().numpy())
            self.g = cg
            lf = self.flat
            lf.rvs, lf.factors = sorted(cg.rvs), sorted(cg.factors)
            lf.var_index = {c: i for i, c in enumerate(lf.rvs)}
            lf.fac_index = {c: i for i, c in enumerate(lf.factors)}
            return
        engine = _DeviceEngine(self)
        refiner = _c2f.DeviceRefiner(ground)
        st, flat, cg, rvc, fc, history = _c2f.run_c2f(ground, engine, refiner, iteration, c2f, self.k_mean_k,
                                                      self.k_mean_iteration, self._c2f_draw(),
                                                      observer=getattr(self, 'c2f_observer', None))
        self.c2f_history = history
        self.g = cg
        # adopt the final factor-side state for the queries
        self.__dict__.update({k: v for k, v in st.__dict__.items()})
        self._views, self._batched = {}, {}

    # ---- queries of GROUND variables on arrays (a solver made by on_flat) --------------------------------------------------------
    def _ground_pairs(self):
        """(ground variable, lifted edge, multiplicity) of every distinct pair: ``belief_rv_query`` (HLBP:313-317) walks the
        GROUND rv's factors f and evaluates the message of (f.cluster, position of the rv)"""
        G = self._ground
        if 'pairs' not in G:
            torch = _abi.require_gpu()
            gf, lf, dg = G['flat'], self.flat, G['dg']
            fcl = G['fc'].long()
            dev = fcl.device
            efac = dg.t['edge_fac'].long()
            pos = torch.arange(gf.E, device=dev) - dg.t['fac_ptr'].long()[efac]
            lifted_e = _abi.to_dev(lf.edge_canon.astype(np.int64))[_abi.to_dev(lf.fac_ptr.astype(np.int64))[fcl[efac]] + pos]
            slot_edge = dg.t['var_edge'].long()
            slot_var = dg.t['slot_var'].long()
            key, cnt = torch.unique(slot_var * max(lf.E, 1) + lifted_e[slot_edge], return_counts=True)     # sorted: grouped per variable
            var = key // max(lf.E, 1)
            ptr = torch.zeros(gf.V + 1, dtype=torch.int64, device=dev)
            torch.cumsum(torch.bincount(var, minlength=gf.V), 0, out=ptr[1:])
            G['pairs'] = (ptr, (key % max(lf.E, 1)).to(torch.int32), cnt.to(torch.float64))
        return G['pairs']

    def belief_rv_ground(self, vs, xs):
        """log-beliefs ``belief_rv_query(x, rv)`` of the GROUND variables `vs` (indices into the ground FlatGraph) at `xs[i, :]`
        (the same number of points each), one launch.  Works after any ``run_flat`` (the partition need not be stable)."""
        torch = _abi.require_gpu()
        ptr, edge, mult = self._ground_pairs()
        dev = edge.device
        vs_t = _abi.to_dev(np.asarray(vs, dtype=np.int64))
        x = xs if torch.is_tensor(xs) else _abi.to_dev(np.ascontiguousarray(xs, dtype=np.float64))
        x = x.reshape(vs_t.numel(), -1)
        deg = ptr[vs_t + 1] - ptr[vs_t]
        total = int(deg.sum().item())
        owner = torch.repeat_interleave(torch.arange(vs_t.numel(), device=dev), deg, output_size=total)
        start = torch.cumsum(deg, 0) - deg
        idx = ptr[vs_t][owner] + (torch.arange(total, device=dev) - start[owner])
        xe = x[owner].contiguous()
        out = torch.empty_like(xe)
        if total:
            _abi.check(_abi.lib().lhvi_pbp_edge_points(self.dg.g, self.dg.p, self._struct(), _abi.ptr(self.v2f), total,
                                                       _abi.ptr(edge[idx].contiguous()), int(x.shape[1]), _abi.ptr(xe), _abi.ptr(out),
                                                       _abi.stream_ptr()))
        res = torch.zeros_like(x)
        res.index_add_(0, owner, out * mult[idx][:, None])
        return res

    def _var_of(self, ground_rv):
        return ground_rv                      # queries walk the GROUND rv's factors, like belief_rv_query (HLBP:313-317)

    def _ground_edges(self, ground_rv):
        """(lifted edge id, multiplicity) of every ground factor of `ground_rv`: edge = (f.cluster, position of the rv)"""
        flat, acc = self.flat, {}
        for f in ground_rv.nb:
            fi = flat.fac_index[f.cluster]
            pos = next(i for i, r in enumerate(f.nb) if r is ground_rv)
            e = int(flat.edge_canon[flat.fac_ptr[fi] + pos])
            acc[e] = acc.get(e, 0) + 1
        return acc

    def _belief_rv_points(self, ground_rv, xs):
        """belief_rv_query(x, rv) = sum over the ground rv's factors f of message_f_to_rv(x, f.cluster, rv.cluster)"""
        torch = _abi.require_gpu()
        xs = np.atleast_1d(np.asarray(xs, dtype=np.float64))
        acc = self._ground_edges(ground_rv)
        edges = np.array(list(acc), dtype=np.int32)
        mult = np.array([acc[e] for e in acc], dtype=np.float64)
        x = _abi.to_dev(np.tile(xs, (edges.size, 1)))
        out = torch.empty_like(x)
        _abi.check(_abi.lib().lhvi_pbp_edge_points(self.dg.g, self.dg.p, self._struct(), _abi.ptr(self.v2f), int(edges.size),
                                                   _abi.ptr(_abi.to_dev(edges)), int(xs.size), _abi.ptr(x), _abi.ptr(out),
                                                   _abi.stream_ptr()))
        return (out.cpu().numpy() * mult[:, None]).sum(axis=0)

    def belief_rv_batch(self, rvs, xs):
        xs = np.asarray(xs, dtype=np.float64).reshape(len(rvs), -1)
        return np.stack([self._belief_rv_points(rv, x) for rv, x in zip(rvs, xs)])

    def belief_rv_query(self, x, rv, sample=None):
        return float(self._belief_rv_points(rv, [x])[0])

    @property
    def q(self):
        flat, Q = self.flat, self._host('q_dev')
        return {rv: (float(Q[v, 0]), float(Q[v, 1])) for v, rv in enumerate(flat.rvs) if flat.var_hidden[v] and flat.var_cont[v]}

    # ---- batched queries of GROUND variables (any partition): what the per-variable calls below are answered from ---------------
    def _ground_rows(self):
        """hidden ground variables of the run's ground graph, their domain bounds and kind (arrays path only)"""
        G = self._ground
        if 'rows' not in G:
            gf = G['flat']
            hid = np.flatnonzero(gf.var_hidden)
            cont = gf.var_cont[hid]
            lo = np.where(cont, gf.dom_lo[gf.var_dom[hid]], 0.0)
            hi = np.where(cont, gf.dom_hi[gf.var_dom[hid]], 1.0)
            G['rows'] = (hid, cont, lo, hi, {int(v): i for i, v in enumerate(hid)})
        return G['rows']

    def ground_map_all(self, scan=64, steps=6):
        """MAP of every hidden GROUND variable in a few batched passes of ``belief_rv_ground``: a uniform scan of the domain
        with `scan` points, then `steps` times a new 17-point grid on the bracket around the best point (the bracket shrinks 8x
        per step: 1e-6 of the domain width after 6).  Discrete variables: the first state with the largest belief.  Returns
        (ground variable ids, MAP values)."""
        torch = _abi.require_gpu()
        hid, cont, lo, hi, _ = self._ground_rows()
        gf = self._ground['flat']
        out = np.zeros(hid.size)
        ci = np.flatnonzero(cont)
        if ci.size:
            dev = self.particles.device
            lo_t, hi_t = _abi.to_dev(lo[ci])[:, None], _abi.to_dev(hi[ci])[:, None]
            j = torch.arange(scan, dtype=torch.float64, device=dev)[None, :]
            x = lo_t + (hi_t - lo_t) * (j / (scan - 1))
            vals = self.belief_rv_ground(hid[ci], x)
            bv, arg = vals.max(dim=1)
            bx = x.gather(1, arg[:, None])[:, 0]
            h = ((hi_t - lo_t) / (scan - 1))[:, 0]
            m = 17
            jj = torch.arange(m, dtype=torch.float64, device=dev)[None, :]
            for _ in range(steps):
                a = torch.maximum(bx - h, lo_t[:, 0])[:, None]
                b = torch.minimum(bx + h, hi_t[:, 0])[:, None]
                x = (a + (b - a) * (jj / (m - 1))).contiguous()
                vals = self.belief_rv_ground(hid[ci], x)
                v2, arg = vals.max(dim=1)
                better = v2 >= bv
                bx = torch.where(better, x.gather(1, arg[:, None])[:, 0], bx)
                bv = torch.where(better, v2, bv)
                h = ((b - a) / (m - 1))[:, 0]
            out[ci] = bx.cpu().numpy()
        di = np.flatnonzero(~cont)
        if di.size:
            nst = gf.var_nstates[hid[di]]
            D = int(nst.max())
            states = np.zeros((di.size, D))
            for r, v in enumerate(hid[di]):
                d = gf.var_dom[v]
                vals = gf.dom_val[gf.dom_ptr[d]:gf.dom_ptr[d + 1]]
                states[r, :vals.size] = vals
                states[r, vals.size:] = vals[0]
            lb = self.belief_rv_ground(hid[di], states).cpu().numpy()
            lb[np.arange(D)[None, :] >= nst[:, None]] = -np.inf
            out[di] = states[np.arange(di.size), lb.argmax(axis=1)]
        return hid, out

    def ground_log_area_all(self, npts=20):
        """``log_area`` (HLBP:324-341) over the domain of every hidden continuous GROUND variable: (ids, area, shift)"""
        torch = _abi.require_gpu()
        hid, cont, lo, hi, _ = self._ground_rows()
        ci = np.flatnonzero(cont)
        lin = np.linspace(lo[ci], hi[ci], npts, axis=1)          # the per-call form's abscissae, bit for bit (numpy.linspace per row)
        y = self.belief_rv_ground(hid[ci], lin).cpu().numpy() if ci.size else np.zeros((0, npts))
        mean, mx = y.mean(axis=1), y.max(axis=1)
        shift = np.where(mx - mean > self.max_log_value, mx - self.max_log_value, mean)
        w = np.exp(y - shift[:, None])
        area = (w[:, :-1] + w[:, 1:]).sum(axis=1) * (lin[:, 1] - lin[:, 0]) * 0.5
        return hid[ci], area, shift

    def _batched_ok(self, ground_rv):
        """can a per-variable query be answered from a batched pass?  (a stable partition: rows are clusters; an unstable one:
        through the ground arrays of the arrays path)"""
        if self.exact_queries:
            return None
        if getattr(self, '_stable_partition', False) and getattr(ground_rv, 'cluster', None) in self.flat.var_index:
            return 'cluster'
        G = getattr(self, '_ground', None)
        if G is not None and ground_rv in G['flat'].var_index:
            return 'ground'
        return None

    def ground_map_fminbound_all(self, xtol=1e-5, maxfun=500):
        """MAP of every hidden GROUND variable the way ``HybridLBP.map`` finds it (HLBP:405-424: ``fminbound`` on
        ``-belief_rv_query``; discrete: first state with the largest belief), all in one launch; works on any partition.
        Returns (ground variable ids, MAP values)."""
        hid, cont, lo, hi, _ = self._ground_rows()
        ptr, edge, mult = self._ground_pairs()
        torch = _abi.require_gpu()
        hid_t = _abi.to_dev(hid.astype(np.int64))
        deg = ptr[hid_t + 1] - ptr[hid_t]
        qptr = torch.zeros(hid.size + 1, dtype=torch.int64, device=deg.device)
        torch.cumsum(deg, 0, out=qptr[1:])
        total = int(qptr[-1].item()) if hid.size else 0
        owner = torch.repeat_interleave(torch.arange(hid.size, device=deg.device), deg, output_size=total)
        idx = ptr[hid_t][owner] + (torch.arange(total, device=deg.device) - qptr[:-1][owner])
        rows = self._ground['rvc'].to(torch.int32)[hid_t]
        x, _, _ = self._brent(rows, pairs=(qptr, edge[idx], mult[idx]), xtol=xtol, maxfun=maxfun)
        return hid, x.cpu().numpy()

    def _cached_ground(self, what):
        if what == 'map':
            glob = self.map_mode == 'global'
            what = 'map_global' if glob else 'map'

            def fill():
                ids, vals = self.ground_map_all() if glob else self.ground_map_fminbound_all()
                return dict(zip(ids.tolist(), vals.tolist()))
        else:
            def fill():
                ids, area, shift = self.ground_log_area_all(20)
                return {int(v): (float(a), float(s_)) for v, a, s_ in zip(ids, area, shift)}
        return self._query_cache('ground_' + what, fill)

    def _area_of(self, rv):
        """(z, shift) of the 20-point ``log_area`` over rv's domain (HLBP:361-365), batched when possible"""
        how = self._batched_ok(rv)
        if how == 'cluster':
            zs, shifts = self._cached_area()
            c = self.flat.var_index[rv.cluster]
            return float(zs[c]), float(shifts[c])
        if how == 'ground':
            return self._cached_ground('area')[self._ground['flat'].var_index[rv]]
        return self._log_area(self._var_of(rv), rv.domain.values[0], rv.domain.values[1], 20)

    def belief(self, x, rv, inf_integral=False):
        """HybridLBP.belief (HLBP:343-382): 20-point trapezoid normaliser, cached per cluster"""
        if rv.value is not None:
            return 1 if x == rv.value else 0
        sig, v = (rv.cluster, frozenset(self._ground_edges(rv).items())), self._var_of(rv)
        if rv.domain.continuous:
            if sig not in self.query_cache:
                self.query_cache[sig] = self._area_of(rv)
            z, shift = self.query_cache[sig]
            return e ** (float(self._belief_rv_points(v, [x])[0]) - shift - log(z))
        if sig not in self.query_cache:
            vals = rv.domain.values
            b = dict(zip(vals, (e ** t for t in self._belief_rv_points(v, vals).tolist())))
            self.message_normalization(b)
            self.query_cache[sig] = b
        return self.query_cache[sig][x]

    def probability(self, a, b, rv):
        if rv.value is None and rv.domain.continuous:
            v = self._var_of(rv)
            z, shift = self._area_of(rv)
            num, _ = self._log_area(v, a, b, 5, shift)
            return num / z
        return None

    def map(self, rv):
        if rv.value is not None:
            return rv.value
        how = self._batched_ok(rv)
        if how == 'cluster':
            m = self._cached_map()[self.flat.var_index[rv.cluster]]
            return float(m) if rv.domain.continuous else type(rv.domain.values[0])(m)
        if how == 'ground':
            m = self._cached_ground('map')[self._ground['flat'].var_index[rv]]
            return float(m) if rv.domain.continuous else type(rv.domain.values[0])(m)
        v = self._var_of(rv)
        if rv.domain.continuous:
            from scipy.optimize import fminbound
            return fminbound(lambda val: -float(self._belief_rv_points(v, [val])[0]),
                             rv.domain.values[0], rv.domain.values[1], disp=False)
        vals = list(rv.domain.values)
        return vals[int(np.argmax(self._belief_rv_points(v, vals)))]


class _DeviceEngine:
    """lhvi.c2f engine backed by the HIP kernels: every state is a solver-shaped object holding one lifted graph's arrays"""

    _names = {'q': 'q_dev'}

    def __init__(self, owner):
        self.owner = owner
        self.draws = 0

    def make(self, flat, sides='vf'):
        o = self.owner
        st = HybridLBP.__new__(HybridLBP)
        st.n, st.proposal_approximation, st.sampler, st.seed = o.n, o.proposal_approximation, o.sampler, o.seed
        st.query_cache = dict()
        st._setup(None, flat=flat, sides=sides)
        return st

    def get(self, st, name):
        return getattr(st, self._names.get(name, name))

    def set(self, st, name, value):
        setattr(st, self._names.get(name, name), value)

    def init(self, st):
        _abi.check(_abi.lib().lhvi_pbp_init(st.dg.g, st._struct(), _abi.ptr(st.eta), _abi.ptr(st.q_dev), _abi.ptr(st.f2v),
                                            _abi.ptr(st.v2f), _abi.stream_ptr()))

    def v2f(self, st):
        _abi.check(_abi.lib().lhvi_pbp_v2f(st.dg.g, st._struct(), _abi.ptr(st.f2v), _abi.ptr(st.v2f), _abi.stream_ptr()))

    def proposal(self, st):
        _abi.check(_abi.lib().lhvi_pbp_proposal(st.dg.g, st._struct(), _abi.ptr(st.f2v), _abi.ptr(st.eta), _abi.ptr(st.q_dev),
                                                _abi.stream_ptr()))

    def f2v(self, st):
        _abi.check(_abi.lib().lhvi_pbp_f2v(st.dg.g, st.dg.p, st._struct(), _abi.ptr(st.v2f), _abi.ptr(st.f2v), _abi.stream_ptr()))

    def install(self, st, host_particles):
        st._draws = self.draws
        self.draws += 1
        if host_particles is None:
            st.sampler = 'device'
        else:
            st.sampler = lambda kk, flat, q: host_particles
        st._generate_sample()

    @staticmethod
    def host(t):
        return t.cpu().numpy()

    @staticmethod
    def gather(t, index):
        return t.index_select(0, _abi.to_dev(np.asarray(index, dtype=np.int64))).contiguous()
