"""Running the particle sweep on one GPU or edge-sharded over the GPUs of a node.

One process per GPU (``torch.distributed``, backend ``nccl`` = RCCL over xGMI).  Factors -- with all their
edges -- are partitioned across ranks, so the f2v half sweep is local.  The v2f half needs, per variable, the
sum over *all* incident edges; for boundary variables (incident factors on more than one rank) every rank
computes the partial sum over its local edges and one ``all_to_all_single`` per sweep exchanges the partials
(SURVEY.md section 8(e)).  Particles of a boundary variable are identical on all its owners because the device
sampler is keyed by the variable's global id.
"""
from __future__ import annotations

import numpy as np

from . import _abi


def joint_terms(flat, np_host, edge_mask=None):
    """number of (output point, joint partner particle) terms one f2v launch evaluates (`edge_mask`: the edges it serves)"""
    hid = flat.var_hidden
    tv = flat.edge_var
    npts = np.where(hid[tv], np_host[tv] + np.where(flat.var_cont[tv], flat.var_nstates[tv], 0), 0).astype(np.int64)
    arity = np.diff(flat.fac_ptr)[flat.edge_fac]
    terms = npts.copy()
    # pairwise / unary graphs (the benchmark): partner = the other edge of the factor
    pair = arity == 2
    partner = np.where(flat.edge_pos == 0, np.arange(flat.E) + 1, np.arange(flat.E) - 1)
    partner = np.clip(partner, 0, flat.E - 1)
    pn = np.where(hid[flat.edge_var[partner]], np_host[flat.edge_var[partner]], 1)
    terms = np.where(pair, npts * pn, npts)
    return int(terms.sum() if edge_mask is None else terms[edge_mask].sum())


class SingleRunner:
    """whole graph on one GPU"""

    def __init__(self, bp):
        self.bp = bp

    def init(self):
        bp = self.bp
        _abi.check(_abi.lib().lhvi_pbp_init(bp.dg.g, bp._struct(), _abi.ptr(bp.eta), _abi.ptr(bp.q_dev),
                                            _abi.ptr(bp.f2v), _abi.ptr(bp.v2f), _abi.stream_ptr()))
        bp._generate_sample()

    def sweep(self, f2v_events=None):
        self.bp.sweep(last=False, f2v_events=f2v_events)

    def local_edges(self):
        return self.bp.flat.E

    def work_fraction(self):
        flat = self.bp.flat
        return float(flat.var_hidden[flat.edge_var].mean())

    def f2v_joint_terms(self):
        return joint_terms(self.bp.flat, self.bp.np_host)

    def heavy_stats(self):
        """(edges, joint terms) of the continuous x continuous kernel's work list"""
        return int(getattr(self.bp, 'n_heavy_class', 0) or self.bp.n_heavy), self.bp.heavy_terms

    def heavy_grid_terms(self):
        """joint terms at the integral points of the edges the heavy kernel serves by the grid recurrence"""
        return self.bp.heavy_grid_terms


# =================================================================================================
# edge sharding
# =================================================================================================
def bfs_variable_order(flat):
    """position of every variable in a breadth-first sweep of the factor graph (all components).  Cutting that order into
    equal blocks gives shells whose variables mostly touch the neighbouring shells only: far fewer boundary variables, and
    fewer peers per boundary variable, than cutting the (arbitrary) construction order of the factors."""
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import breadth_first_order
    V = flat.V
    # variable-variable adjacency through shared factors (first scope variable linked to every other: enough for BFS)
    first = flat.edge_var[flat.fac_ptr[:-1][flat.edge_fac]]
    rows, cols = first, flat.edge_var
    keep = rows != cols
    A = csr_matrix((np.ones(int(keep.sum()), dtype=np.int8), (rows[keep], cols[keep])), shape=(V, V))
    A = A + A.T
    pos = np.full(V, -1, dtype=np.int64)
    nxt = 0
    start = 0
    while nxt < V:
        while start < V and pos[start] >= 0:
            start += 1
        if start >= V:
            break
        order, _ = breadth_first_order(A, start, directed=False, return_predecessors=True)
        order = order[pos[order] < 0]
        pos[order] = nxt + np.arange(order.size)
        nxt += order.size
    return pos


def partition_factors(flat, world, partition='refined'):
    """owner rank of every factor -- the one global step of the sharding (a breadth-first sweep of the whole graph).  In a
    multi-process run rank 0 computes it and broadcasts the array (``ShardedRunner``); everything else a rank needs is
    derived from its own slice of the graph.  'refined' (default): a factor goes where its first hidden variable lives under
    the refined variable partition (``partition_variables``: about half the boundary variables of the breadth-first blocks)."""
    F = flat.F
    dtype = np.int32
    if partition == 'refined' and world > 1:
        owner = partition_variables(flat, world, 'refined')
        hid = np.isnan(flat.var_value)[flat.edge_var]
        pos = np.arange(flat.E) - flat.fac_ptr[flat.edge_fac]
        # where its first hidden argument lives (even factor ids) or its last one (odd ids): a cut factor's two candidate ranks
        # share such factors evenly; a factor without a hidden argument: its first argument
        lo_key = np.where(hid, pos, pos + 64)
        hi_key = np.where(hid, pos + 64, pos)
        if F == 0:
            return np.zeros(0, dtype=dtype)
        first = np.minimum.reduceat(lo_key, flat.fac_ptr[:-1]) % 64
        last = np.maximum.reduceat(hi_key, flat.fac_ptr[:-1]) % 64
        any_hid = np.maximum.reduceat(hid.astype(np.int8), flat.fac_ptr[:-1]) > 0
        pick = np.where(any_hid & (np.arange(F) % 2 == 1), last, first)
        return owner[flat.edge_var[flat.fac_ptr[:-1] + pick]].astype(dtype)
    if partition == 'bfs' and world > 1:
        pos = bfs_variable_order(flat)
        fpos = np.minimum.reduceat(pos[flat.edge_var], flat.fac_ptr[:-1]) if F else np.zeros(0, dtype=np.int64)
        forder = np.argsort(fpos, kind='stable')
        fac_owner = np.empty(F, dtype=dtype)
        fac_owner[forder] = ((np.arange(F, dtype=np.int64) * world) // max(F, 1)).astype(dtype)
        return fac_owner
    return ((np.arange(F, dtype=np.int64) * world) // max(F, 1)).astype(dtype)


def _broadcast_from_first(own, group=None):
    """broadcast a host array from the first rank of `group` (None: the default group) to the others"""
    import torch
    import torch.distributed as td
    t = torch.from_numpy(own)
    if td.get_backend(group) == 'nccl':
        t = t.cuda()
    td.broadcast(t, td.get_global_rank(group, 0) if group is not None else 0, group=group)
    return t.cpu().numpy()


def broadcast_partition(flat, rank, world, partition='refined', group=None):
    """``partition_factors`` on rank 0 (of `group`), broadcast to the other ranks of the initialised process group (4 bytes per
    factor); the other ranks never run the global breadth-first sweep"""
    own = partition_factors(flat, world, partition) if rank == 0 else np.empty(flat.F, dtype=np.int32)
    return _broadcast_from_first(own, group)


def broadcast_variable_partition(flat, rank, world, partition='refined', group=None):
    """``partition_variables`` on rank 0 (of `group`), broadcast to the other ranks (4 bytes per variable)"""
    own = partition_variables(flat, world, partition) if rank == 0 else np.empty(flat.V, dtype=np.int32)
    return _broadcast_from_first(own, group)


class ShardPlan:
    """Factor-partitioned shard of a ground ``FlatGraph`` for rank ``rank`` of ``world`` (pure NumPy/SciPy, no GPU).

    * ``partition='refined'`` (default): see ``partition_factors``; ``'bfs'``: factors are ordered by the breadth-first position of their earliest scope variable and
      cut into ``world`` equal blocks (locality-aware); ``'block'``: equal blocks of the construction order;
    * the shard's factors (``fac_ids``, ascending global id) with all their edges (``edge_ids``) are local;
    * local variables = the variables those factors touch (``var_gid``): the ``n_interior`` interior ones first, then the
      boundary ones, ascending global id inside each block; ``edge_boundary[e]`` = 1 when the edge's factor touches a
      boundary variable;
    * ``var_degree`` = a variable's degree in the WHOLE graph (site clamp / initial site of the proposal);
    * a local hidden variable is a *boundary* variable when some of its edges live on another rank; ``bvars`` lists them
      (ascending gid) and ``peer_rows[s]`` = the entries of ``bvars`` shared with rank ``s`` (ascending gid on both sides,
      so the two ends of a pair agree on the order without communicating).
    """

    def __init__(self, flat, rank, world, partition='refined', fac_owner=None):
        """`fac_owner`: the factor partition when it was computed elsewhere (``partition_factors`` on rank 0); apart from
        it the plan touches only this rank's factors and the adjacency rows of their variables -- O(E / world) work"""
        from .flat import build_flat
        if flat.lifted or (flat.edge_canon != np.arange(flat.E)).any():
            raise NotImplementedError('sharding expects a ground graph')
        self.rank, self.world = rank, world
        if fac_owner is None:
            fac_owner = partition_factors(flat, world, partition)
        self.fac_ids = np.flatnonzero(fac_owner == rank)
        arity = np.diff(flat.fac_ptr)[self.fac_ids]
        local_ptr = np.zeros(self.fac_ids.size + 1, dtype=np.int64)
        np.cumsum(arity, out=local_ptr[1:])
        # global edge ids of the shard, factor-major like the local numbering
        self.edge_ids = (np.repeat(flat.fac_ptr[self.fac_ids].astype(np.int64) - local_ptr[:-1], arity) +
                         np.arange(int(local_ptr[-1]), dtype=np.int64))
        local_edge_var = flat.edge_var[self.edge_ids]
        gids, local_deg = np.unique(local_edge_var, return_counts=True)
        degree = np.diff(flat.var_ptr)                       # a variable's degree in the whole graph (view, no pass over E)
        # boundary = hidden variable with edges on another rank (observed variables need no sums).  Local numbering: the
        # interior variables first, then the boundary ones, ascending global id inside each block -- the per-variable
        # kernels can then sweep [0, n_interior) while the boundary rows are in flight
        is_b = (degree[gids] > local_deg) & np.isnan(flat.var_value[gids])
        gids = gids[np.argsort(is_b, kind='stable')]
        self.n_interior = int((~is_b).sum())
        self.var_gid = gids.astype(np.int64)
        lid = np.full(flat.V, -1, dtype=np.int64)
        lid[gids] = np.arange(gids.size)
        self.var_degree = degree[gids].astype(np.float64)
        self.flat = build_flat(local_ptr.astype(np.int32), lid[local_edge_var].astype(np.int32),
                               flat.fac_pot[self.fac_ids], [], flat.var_value[gids], flat.var_dom[gids], flat.domains)
        # the potential table is global and small: keep it whole so fac_pot stays valid
        self.flat.pot_kind, self.flat.pot_off, self.flat.pot_param = flat.pot_kind, flat.pot_off, flat.pot_param
        # boundary bookkeeping: which ranks own edges of each boundary variable
        is_b = np.arange(gids.size) >= self.n_interior
        self.bvars = np.flatnonzero(is_b).astype(np.int32)               # local ids = [n_interior, V), ascending gid
        # an edge is interior when its factor touches no boundary variable: its message needs nothing from the exchange
        fac_b = np.maximum.reduceat(is_b[self.flat.edge_var].astype(np.int8), local_ptr[:-1]) if self.fac_ids.size else np.zeros(0, np.int8)
        self.edge_boundary = np.repeat(fac_b, arity).astype(np.int32)
        self.bslot = np.full(gids.size, -1, dtype=np.int32)
        self.bslot[self.bvars] = np.arange(self.bvars.size, dtype=np.int32)
        bg = gids[self.bvars]
        # (boundary variable, owner rank) pairs from the adjacency rows of THIS rank's boundary variables only
        bdeg = degree[bg].astype(np.int64)
        bstart = np.zeros(bg.size + 1, dtype=np.int64)
        np.cumsum(bdeg, out=bstart[1:])
        slots = np.repeat(flat.var_ptr[bg].astype(np.int64) - bstart[:-1], bdeg) + np.arange(int(bstart[-1]), dtype=np.int64)
        owners = fac_owner[flat.edge_fac[flat.var_edge[slots]]].astype(np.int64)
        keys = np.unique(np.repeat(bg.astype(np.int64), bdeg) * world + owners)
        pair_var, pair_owner = keys // world, keys % world
        # exchange rows: peer-major, shared variables in ascending gid inside a peer block.  Both ends build the same
        # order, so row r of my send buffer and row r of my receive buffer belong to the same (variable, peer).
        self.peer_rows = {}
        row_bvar, row_peer = [], []
        self.counts = []
        for s in range(world):
            if s == rank:
                self.counts.append(0)
                continue
            vs = pair_var[pair_owner == s]
            shared = np.intersect1d(bg, vs, assume_unique=True)          # ascending gid
            rows = np.searchsorted(bg, shared).astype(np.int64)          # indices into bvars
            self.peer_rows[s] = rows
            row_bvar.append(rows)
            row_peer.append(np.full(rows.size, s, dtype=np.int32))
            self.counts.append(int(rows.size))
        row_bvar = np.concatenate(row_bvar) if row_bvar else np.zeros(0, dtype=np.int64)
        row_peer = np.concatenate(row_peer) if row_peer else np.zeros(0, dtype=np.int32)
        self.n_rows = int(row_bvar.size)
        # CSR boundary variable -> its rows, in ascending peer order (stable sort keeps the peer-major order)
        order = np.argsort(row_bvar, kind='stable')
        self.brow_idx = order.astype(np.int32)
        self.brow_peer = row_peer[order].astype(np.int32)
        self.brow_ptr = np.zeros(self.bvars.size + 1, dtype=np.int32)
        np.cumsum(np.bincount(row_bvar, minlength=self.bvars.size), out=self.brow_ptr[1:])

    def send_counts(self):
        return list(self.counts)


def owner_exchange_layout(plan, bwidth):
    """Buffers of the reduce-to-owner exchange of one rank (NumPy only).  ``bwidth[b]`` = doubles in boundary variable b's row.

    Every boundary variable has one owner among the ranks that hold edges of it -- entry ``gid % k`` of the ascending list of
    its k ranks, which every one of them computes alike.  Step A: each other rank sends the owner its row; the owner adds the
    rows in ascending rank order.  Step B: the owner sends the total back.  2 (k - 1) rows per variable instead of the
    k (k - 1) of the all-to-all form.  Element offsets:

    * buffer A = [rows to the owners, owner-major | own rows of the variables owned here | rows received, sender-major],
      ``pack_off[b]`` = where boundary variable b's local row goes (first or second block);
    * buffer B = [totals to the replicas, receiver-major | totals received, owner-major | totals of the owned variables],
      ``total_off[b]`` = where b's total is read after step B (second or third block);
    * ``items`` = the owned variables; ``src_ptr / src_off`` their rows in A in ascending rank order; ``dst_ptr / dst_off``
      their total's places in B;  ``a_send / a_recv / b_send / b_recv`` = elements per peer of the two collectives.
    Inside a peer's block rows are in ascending global id on both ends, so no index travels."""
    rank, world = plan.rank, plan.world
    nb = int(plan.bvars.size)
    bwidth = np.asarray(bwidth, dtype=np.int64)
    npeers = np.diff(plan.brow_ptr).astype(np.int64)
    pair_b = np.repeat(np.arange(nb, dtype=np.int64), npeers)
    pair_peer = plan.brow_peer.astype(np.int64)
    gid = plan.var_gid[plan.bvars].astype(np.int64)
    k = npeers + 1
    p_own = np.bincount(pair_b, weights=(pair_peer < rank), minlength=nb).astype(np.int64)      # own position among the sorted ranks
    idx = gid % k
    look = plan.brow_ptr[:-1].astype(np.int64) + idx - (idx > p_own)
    owner = np.where(idx == p_own, rank, pair_peer[np.minimum(look, max(pair_peer.size - 1, 0))] if pair_peer.size else rank)
    owned = owner == rank

    def block(bs, peers, base):
        """rows (bs[i], peers[i]) laid out peer-major, ascending b inside a peer: offsets per row, elements per peer, end"""
        order = np.lexsort((bs, peers))
        w = bwidth[bs[order]]
        off = np.empty(bs.size, dtype=np.int64)
        off[order] = base + np.concatenate([[0], np.cumsum(w)[:-1]]) if bs.size else 0
        per_peer = np.bincount(peers, weights=bwidth[bs], minlength=world).astype(np.int64) if bs.size else np.zeros(world, np.int64)
        return off, per_peer, base + int(w.sum())
    mine, theirs = np.flatnonzero(owned), np.flatnonzero(~owned)
    own_pairs = owned[pair_b]
    ob, op = pair_b[own_pairs], pair_peer[own_pairs]                 # (owned variable, replica rank)
    # buffer A
    a_send_off, a_send, end = block(theirs, owner[theirs], 0)
    own_off = end + np.concatenate([[0], np.cumsum(bwidth[mine])[:-1]]) if mine.size else np.zeros(0, np.int64)
    end += int(bwidth[mine].sum())
    a_recv_base = end
    a_recv_off, a_recv, a_end = block(ob, op, end)
    pack_off = np.zeros(nb, dtype=np.int64)
    pack_off[theirs], pack_off[mine] = a_send_off, own_off
    # buffer B
    b_send_off, b_send, end = block(ob, op, 0)
    b_recv_base = end
    b_recv_off, b_recv, end = block(theirs, owner[theirs], end)
    tot_off = end + np.concatenate([[0], np.cumsum(bwidth[mine])[:-1]]) if mine.size else np.zeros(0, np.int64)
    b_end = end + int(bwidth[mine].sum())
    total_off = np.zeros(nb, dtype=np.int64)
    total_off[theirs], total_off[mine] = b_recv_off, tot_off
    # the owner's sums: sources in ascending rank order, destinations in any order
    item_of = np.full(nb, -1, dtype=np.int64)
    item_of[mine] = np.arange(mine.size)
    sb = np.concatenate([item_of[ob], np.arange(mine.size)])
    srank = np.concatenate([op, np.full(mine.size, rank, dtype=np.int64)])
    soff = np.concatenate([a_recv_off, own_off])
    order = np.lexsort((srank, sb))
    src_ptr = np.zeros(mine.size + 1, dtype=np.int64)
    np.cumsum(np.bincount(sb, minlength=mine.size), out=src_ptr[1:])
    doff = np.concatenate([b_send_off, tot_off])
    dorder = np.argsort(sb, kind='stable')
    return dict(owner=owner, owned=owned, items=mine, width=bwidth[mine].astype(np.int32), pack_off=pack_off, total_off=total_off,
                src_ptr=src_ptr.astype(np.int32), src_off=soff[order], dst_ptr=src_ptr.astype(np.int32), dst_off=doff[dorder],
                a_size=a_end, b_size=b_end, a_recv_base=a_recv_base, b_recv_base=b_recv_base,
                a_send=a_send.tolist(), a_recv=a_recv.tolist(), b_send=b_send.tolist(), b_recv=b_recv.tolist())


class _EventWork:
    """what ``exchange(async_op=True)`` hands back when the exchange ran on a side stream: ``wait()`` makes the CURRENT stream wait
    for the event recorded behind it (the call ``torch.distributed``'s work handle offers, on an event of ours)"""

    def __init__(self, event):
        self.event = event

    def wait(self):
        import torch
        torch.cuda.current_stream().wait_event(self.event)


class LoopbackGroup:
    """In-process stand-in for the all-to-all of `world` simulated ranks (single-GPU parity tests of the sharded path)."""

    def __init__(self, world):
        self.world = world
        self.sends = [None] * world

    def post(self, rank, send, counts):
        self.sends[rank] = (send, counts)

    def collect(self, rank, width):
        import torch
        out = []
        for s in range(self.world):
            if s == rank:
                continue
            send, counts = self.sends[s]
            off = sum(counts[:rank])
            out.append(send[off:off + counts[rank]])
        return torch.cat(out) if out else self.sends[rank][0][:0]


class ShardedRunner:
    """One rank's part of the edge-sharded particle sweep (EPBP semantics).

    Per sweep: local sites + information-form partials of the BOUNDARY variables -> their rows packed straight into the
    peer-ordered send buffer -> ONE all_to_all, started asynchronously -> while it is in flight, the whole sweep of the
    interior part (partials, v2f, proposal, resample of the interior variables; f2v of the edges whose factor touches no
    boundary variable) -> wait -> the boundary part: v2f and proposal finish read the received rows in place (no unpack
    pass), resample (Philox keyed by global id, so replicas of a boundary variable draw identical particles without
    communicating), f2v of the remaining edges.  The two parts are variable ranges ([0, n_interior) and [n_interior, V)
    of the plan's numbering) and prefixes / suffixes of the f2v work lists, so no kernel needs an index list.
    """

    def __init__(self, flat, n, seed, rank, world, proposal_approximation='simple', group=None, overlap=True,
                 fac_owner=None, owner_reduce=False):
        import torch
        from .pbp import EPBP
        self.plan = plan = ShardPlan(flat, rank, world, fac_owner=fac_owner)
        self.rank, self.world, self.group = rank, world, group
        bp = EPBP(None, n=n, proposal_approximation=proposal_approximation, sampler='device', seed=seed)
        bp._setup(None, flat=plan.flat, edge_key=plan.edge_boundary)
        self.bp = bp
        dev = bp.dg.device
        bp.var_gid = _abi.to_dev(plan.var_gid)
        self.var_degree = _abi.to_dev(plan.var_degree)
        self.bslot = _abi.to_dev(plan.bslot)
        self.bvars = _abi.to_dev(plan.bvars)
        nb = int(plan.bvars.size)
        self.nb, self.W = nb, n + 2
        self.n_int = int(plan.n_interior)
        # two-part schedule: needs both parts non-empty and the fused resample kernel (n <= 64), which takes a range
        self.overlap = bool(overlap) and world > 1 and nb > 0 and self.n_int > 0 and n <= 64
        self.f2v_extra = []                 # event pairs around the second heavy-kernel launch of a sweep (bench)
        self.record_phases = False          # sweep() records HIP events at its phase boundaries (bench.py --gpus N)
        self._phase_events = []
        self.ph = torch.zeros(plan.flat.V, 2, dtype=torch.float64, device=dev)
        # exchange rows, packed back to back in peer-major order: n + 2 doubles for a continuous boundary variable, its
        # np states for a discrete one (nothing else of a discrete variable is exchanged)
        lf = plan.flat
        bcont = lf.var_cont[plan.bvars]
        bwidth = np.where(bcont, n + 2, bp.np_host[plan.bvars]).astype(np.int64)
        row_bvar = np.concatenate([plan.peer_rows[s] for s in range(world) if s != rank]) if plan.n_rows else np.zeros(0, np.int64)
        row_width = bwidth[row_bvar]
        row_off = np.concatenate([[0], np.cumsum(row_width)]).astype(np.int64)
        self.n_elems = int(row_off[-1])
        self.send = torch.zeros(max(self.n_elems, 1), dtype=torch.float64, device=dev)
        self.recv = torch.zeros(max(self.n_elems, 1), dtype=torch.float64, device=dev)
        self.brow_ptr = _abi.to_dev(plan.brow_ptr)
        self.brow_off = _abi.to_dev(row_off[:-1][plan.brow_idx] if plan.n_rows else np.zeros(1, dtype=np.int64))
        self.brow_peer = _abi.to_dev(plan.brow_peer if plan.n_rows else np.zeros(1, dtype=np.int32))
        # elements per peer (what the all_to_all splits by); rows per peer stay in plan.counts
        ends = np.cumsum([c for c in plan.counts])
        self.counts = [int(row_off[e] - row_off[e - c]) for e, c in zip(ends, plan.counts)]
        # reduce-to-owner form of the exchange (owner_exchange_layout): two smaller collectives with the owners' sums between
        self.owner_reduce = bool(owner_reduce) and world > 1 and nb > 0
        if self.owner_reduce:
            lay = self.lay = owner_exchange_layout(plan, bwidth)
            self.bufA = torch.zeros(max(lay['a_size'], 1), dtype=torch.float64, device=dev)
            self.bufB = torch.zeros(max(lay['b_size'], 1), dtype=torch.float64, device=dev)
            self.one_ptr = _abi.to_dev(np.arange(nb + 1, dtype=np.int32))
            self.pack_off, self.total_off = _abi.to_dev(lay['pack_off']), _abi.to_dev(lay['total_off'])
            self.red = {k: _abi.to_dev(np.ascontiguousarray(lay[k]) if lay[k].size else np.zeros(1, dtype=lay[k].dtype))
                        for k in ('width', 'src_ptr', 'src_off', 'dst_ptr', 'dst_off')}
            self.n_items = int(lay['items'].size)
            self.a_send_elems, self.b_send_elems = int(sum(lay['a_send'])), int(sum(lay['b_send']))
            self.a_recv_elems, self.b_recv_elems = int(sum(lay['a_recv'])), int(sum(lay['b_recv']))
            self.counts = list(lay['a_send'])            # (what a loopback group cuts this rank's first send by)
            self.side = None

    def _struct(self, part=None, leave_room=True):
        """`part`: None = everything, 0 = interior variables, 1 = boundary variables (variable range only)"""
        s = self.bp._struct()
        if self.overlap and leave_room:
            s.flags |= _abi.PBP_LEAVE_ROOM         # the exchange is in flight beside the persistent f2v kernels
        s.bslot, s.var_degree = _abi.ptr(self.bslot), _abi.ptr(self.var_degree)
        s.brow_ptr, s.brow_off, s.brow_peer = _abi.ptr(self.brow_ptr), _abi.ptr(self.brow_off), _abi.ptr(self.brow_peer)
        s.recv, s.rank = _abi.ptr(self.recv), int(self.rank)
        if self.owner_reduce:                                # one row per boundary variable: the total, in buffer B
            s.flags |= _abi.PBP_BOUNDARY_TOTALS
            s.brow_ptr, s.brow_off, s.recv = _abi.ptr(self.one_ptr), _abi.ptr(self.total_off), _abi.ptr(self.bufB)
        s.prop_desc, s.n_prop_desc = None, 0                 # the sharded proposal addresses variables by range
        s.prop_hub, s.n_prop_hub, s.prop_partial = None, 0, None
        s.v2f_wide, s.n_v2f_wide, s.v2f_narrow, s.n_v2f_narrow, s.v2f_hub, s.n_v2f_hub = None, 0, None, 0, None, 0     # ... and so does the sharded v -> f half
        s.v2f_mid16, s.n_v2f_mid16, s.v2f_mid32, s.n_v2f_mid32 = None, 0, None, 0
        if part is not None:
            s.var_lo, s.var_hi = (0, self.n_int) if part == 0 else (self.n_int, self.plan.flat.V)
        return s

    def _edge_part(self, s, part):
        """restrict the f2v work lists of `s` to the interior prefix (part 0) or the boundary suffix (part 1)"""
        bp, pc = self.bp, self.bp.part_counts

        def cut(total, first):
            return (0, first) if part == 0 else (first, total - first)
        D = _abi.PBP_DESC_BYTES
        off, cnt = cut(bp.n_heavy, pc['heavy'])
        s.heavy_desc, s.n_heavy = (bp.heavy_desc.data_ptr() + off * D if cnt else None), cnt
        for name in ('small16', 'small32'):              # (heavy-class edges with few particles: four / two per wavefront)
            off, cnt = cut(int(getattr(bp, 'n_' + name, 0)), pc.get(name, 0))
            setattr(s, name + '_desc', (getattr(bp, name + '_desc').data_ptr() + off * D) if cnt else None)
            setattr(s, 'n_' + name, cnt)
        off, cnt = cut(bp.n_light, pc['light'])
        s.light_desc, s.n_light = (bp.light_desc.data_ptr() + off * D if cnt else None), cnt
        off, cnt = cut(int(getattr(bp, 'n_pair', 0)), pc.get('pair', 0))
        s.pair_desc, s.n_pair = (bp.pair_desc.data_ptr() + off * D if cnt else None), cnt
        off, cnt = cut(int(getattr(bp, 'n_cq', 0)), pc.get('cq', 0))
        s.cq_desc, s.n_cq = (bp.cq_desc.data_ptr() + off * 2 * D if cnt else None), cnt
        off, cnt = cut(int(bp.fast_edges.numel()), pc['fast'])
        if cnt:
            s.fast_edges, s.fast_desc = bp.fast_edges.data_ptr() + 4 * off, bp.fast_desc.data_ptr() + off * D
        s.n_fast = cnt                       # 0 with a non-null list pointer = nothing to do
        off, cnt = cut(int(bp.generic_edges.numel()), pc['generic'])
        if cnt:
            s.generic_edges = bp.generic_edges.data_ptr() + 4 * off
        s.n_generic = cnt
        return s

    def init(self):
        bp = self.bp
        _abi.check(_abi.lib().lhvi_pbp_init(bp.dg.g, self._struct(), _abi.ptr(bp.eta), _abi.ptr(bp.q_dev),
                                            _abi.ptr(bp.f2v), _abi.ptr(bp.v2f), _abi.stream_ptr()))
        bp._generate_sample()

    # -- phase 1: everything that must precede the exchange -----------------------------------------
    def pre(self, part=None):
        """site updates + information-form partials (of one part), boundary rows packed into the send buffer"""
        bp, l, st = self.bp, _abi.lib(), _abi.stream_ptr()
        s = self._struct(part)
        _abi.check(l.lhvi_pbp_proposal_partial(bp.dg.g, s, _abi.ptr(bp.f2v), _abi.ptr(bp.eta), _abi.ptr(self.ph), st))
        if part != 0 and self.owner_reduce:
            s.brow_off = _abi.ptr(self.pack_off)             # one row per boundary variable: to its owner, or kept when owned here
            _abi.check(l.lhvi_pbp_boundary_pack(bp.dg.g, s, _abi.ptr(bp.f2v), _abi.ptr(self.ph), self.nb, _abi.ptr(self.bvars),
                                                _abi.ptr(self.bufA), st))
        elif part != 0:
            _abi.check(l.lhvi_pbp_boundary_pack(bp.dg.g, s, _abi.ptr(bp.f2v), _abi.ptr(self.ph), self.nb, _abi.ptr(self.bvars),
                                                _abi.ptr(self.send), st))
        return self.bufA[:self.a_send_elems] if self.owner_reduce else self.send[:self.n_elems]

    def owner_sums(self, rows=None, stream=None):
        """reduce-to-owner form, between its two collectives: the totals of the variables owned here from the rows received
        (`rows`: a loopback group's copy of them; None = already in buffer A); returns the block of totals to send back"""
        lay = self.lay
        if rows is not None:
            self.bufA[lay['a_recv_base']:lay['a_recv_base'] + self.a_recv_elems].copy_(rows)
        r = self.red
        _abi.check(_abi.lib().lhvi_pbp_boundary_reduce(self.n_items, _abi.ptr(r['width']), _abi.ptr(r['src_ptr']), _abi.ptr(r['src_off']),
                                                       _abi.ptr(r['dst_ptr']), _abi.ptr(r['dst_off']), _abi.ptr(self.bufA),
                                                       _abi.ptr(self.bufB), stream if stream is not None else _abi.stream_ptr()))
        return self.bufB[:self.b_send_elems]

    def _install(self, recv):
        """rows handed in by a loopback group go where the kernels read them (a real collective wrote them there already)"""
        if self.owner_reduce:
            dst = self.bufB[self.lay['b_recv_base']:self.lay['b_recv_base'] + self.b_recv_elems]
        else:
            dst = self.recv[:recv.shape[0]]
        if recv.data_ptr() != dst.data_ptr():
            dst.copy_(recv)

    def exchange(self, send, async_op=False):
        """the one collective of a sweep; returns the receive buffer (and the work handle when `async_op`)"""
        import torch.distributed as td
        if self.owner_reduce:
            return self._exchange_owner(async_op)
        recv = self.recv[:self.n_elems]         # symmetric: the rows shared with rank s are sent to and received from s
        splits = list(self.counts)
        work = None
        if td.get_backend(self.group) == 'nccl':
            work = td.all_to_all_single(recv.view(-1), send.reshape(-1), output_split_sizes=splits, input_split_sizes=splits,
                                        async_op=async_op, group=self.group)
        else:
            # rehearsal backend (gloo): same collective on host copies
            h_send = send.reshape(-1).cpu()
            h_recv = h_send.new_empty(h_send.shape)
            td.all_to_all_single(h_recv, h_send, output_split_sizes=splits, input_split_sizes=splits, group=self.group)
            recv.view(-1).copy_(h_recv)
        return (recv, work) if async_op else recv

    def _exchange_owner(self, async_op):
        """rows to the owners -> the owners' sums -> totals back.  Asynchronous form: the sums and the second collective are
        issued on a side stream, so that they run as soon as the first collective ends and not behind the interior part of the
        sweep queued on the compute stream"""
        import torch
        import torch.distributed as td
        lay = self.lay
        sendA = self.bufA[:self.a_send_elems]
        recvA = self.bufA[lay['a_recv_base']:lay['a_recv_base'] + self.a_recv_elems]
        sendB = self.bufB[:self.b_send_elems]
        recvB = self.bufB[lay['b_recv_base']:lay['b_recv_base'] + self.b_recv_elems]
        if td.get_backend(self.group) != 'nccl':
            # rehearsal backend (gloo): same collectives on host copies
            h = torch.empty(self.a_recv_elems, dtype=torch.float64)
            td.all_to_all_single(h, sendA.cpu(), output_split_sizes=lay['a_recv'], input_split_sizes=lay['a_send'], group=self.group)
            recvA.copy_(h)
            self.owner_sums()
            h = torch.empty(self.b_recv_elems, dtype=torch.float64)
            td.all_to_all_single(h, sendB.cpu(), output_split_sizes=lay['b_recv'], input_split_sizes=lay['b_send'], group=self.group)
            recvB.copy_(h)
            return (recvB, None) if async_op else recvB
        if not async_op:
            td.all_to_all_single(recvA, sendA, output_split_sizes=lay['a_recv'], input_split_sizes=lay['a_send'], group=self.group)
            self.owner_sums()
            td.all_to_all_single(recvB, sendB, output_split_sizes=lay['b_recv'], input_split_sizes=lay['b_send'], group=self.group)
            return recvB
        if self.side is None:
            self.side = torch.cuda.Stream()
            self.packed, self.exchanged = torch.cuda.Event(), torch.cuda.Event()
        # explicit ordering of the two buffers across the streams (not the collectives' work handles alone): the side stream
        # starts when the compute stream has packed buffer A (`packed`); the compute stream may read buffer B -- and, one sweep
        # later, overwrite buffer A -- only after `exchanged`, recorded on the side stream behind the first collective (reads A's
        # send block, writes its receive block), the owners' sums (read A, write B) and the second collective (reads / writes B)
        self.packed.record(torch.cuda.current_stream())
        self.side.wait_event(self.packed)
        with torch.cuda.stream(self.side):
            td.all_to_all_single(recvA, sendA, output_split_sizes=lay['a_recv'], input_split_sizes=lay['a_send'], group=self.group)
            self.owner_sums(stream=self.side.cuda_stream)
            td.all_to_all_single(recvB, sendB, output_split_sizes=lay['b_recv'], input_split_sizes=lay['b_send'], group=self.group)
            self.exchanged.record(self.side)
        return recvB, _EventWork(self.exchanged)

    # -- phase 2: the rest of the sweep, reading the peers' rows straight from the receive buffer ---------------------
    def _f2v(self, s, f2v_events=None):
        self.bp._launch_f2v(s, f2v_events)

    def post(self, recv, f2v_events=None):
        """plain schedule: everything after the exchange, over all variables and edges"""
        bp, l, st = self.bp, _abi.lib(), _abi.stream_ptr()
        # the kernels read the received rows in place (v2f: particle-part sums; proposal_finish: information-form sums in
        # rank order, so every replica of a boundary variable forms bit-identical q and draws bit-identical particles)
        self._install(recv)
        s = self._struct()
        _abi.check(l.lhvi_pbp_v2f(bp.dg.g, s, _abi.ptr(bp.f2v), _abi.ptr(bp.v2f), st))
        _abi.check(l.lhvi_pbp_proposal_finish(bp.dg.g, s, _abi.ptr(self.ph), _abi.ptr(bp.q_dev), st))
        bp._generate_sample()
        self._f2v(self._struct(), f2v_events)

    def _var_part(self, part, swapped):
        """v2f, proposal and new particles of one variable range.  `swapped`: the particle buffers were already exchanged
        for this sweep (by the other part), so the sample v2f weighs is in `old_particles`"""
        bp, l, st = self.bp, _abi.lib(), _abi.stream_ptr()
        s = self._struct(part)
        if swapped:
            s.particles = _abi.ptr(bp.old_particles)
        _abi.check(l.lhvi_pbp_v2f(bp.dg.g, s, _abi.ptr(bp.f2v), _abi.ptr(bp.v2f), st))
        _abi.check(l.lhvi_pbp_proposal_finish(bp.dg.g, s, _abi.ptr(self.ph), _abi.ptr(bp.q_dev), st))
        if not swapped:
            bp.old_particles, bp.particles = bp.particles, bp.old_particles
            bp._draws += 1
            bp._views = {}
        s = self._struct(part)
        _abi.check(l.lhvi_pbp_resample_uniq(bp.dg.g, s, _abi.ptr(bp.var_gid), int(bp.seed), int(bp._draws - 1),
                                            _abi.ptr(bp.particles), _abi.ptr(bp.uniq), st))

    def interior(self, f2v_events=None):
        """the part of the sweep that needs nothing from the peers"""
        self.pre(part=0)
        self._var_part(0, swapped=False)
        self._f2v(self._edge_part(self._struct(), 0), f2v_events)

    def boundary(self, recv, f2v_events=None):
        """the rest, once the peers' rows have arrived (after `interior`)"""
        self._install(recv)
        self._var_part(1, swapped=True)
        # (the collective has completed: this launch may fill the device; with the reduce-to-owner form nothing is in flight either)
        self._f2v(self._edge_part(self._struct(leave_room=False), 1), f2v_events)

    def sweep(self, f2v_events=None):
        if not self.overlap:
            send = self.pre()
            recv = self.exchange(send) if self.world > 1 else send
            self.post(recv, f2v_events)
            return
        import torch
        marks = []

        def mark():
            if self.record_phases:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                marks.append(ev)
        mark()
        send = self.pre(part=1)                             # boundary rows first: the exchange starts as early as it can
        mark()
        recv, work = self.exchange(send, async_op=True)
        extra = None
        if f2v_events:
            extra = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self.f2v_extra.append(extra)
        self.interior(extra)
        mark()
        if work is not None:
            work.wait()                                     # the compute stream waits for the collective
        mark()
        self.boundary(recv, f2v_events)
        mark()
        if marks:
            self._phase_events.append(marks)

    def phase_ms(self):
        """mean HIP-event time of each phase of the two-part schedule on this rank's compute stream (call after a
        synchronisation): boundary partials + pack, the interior part, the wait for the collective that is NOT hidden behind
        it, the boundary part"""
        if not self._phase_events:
            return None
        names = ('pack', 'interior', 'exchange_wait', 'boundary')
        acc = {k: 0.0 for k in names}
        for m in self._phase_events:
            for k, a, b in zip(names, m[:-1], m[1:]):
                acc[k] += a.elapsed_time(b)
        n = len(self._phase_events)
        out = {k: v / n for k, v in acc.items()}
        out.update(sweeps=n, boundary_variables=int(self.nb), interior_variables=int(self.n_int),
                   exchange='reduce to owner + totals back' if self.owner_reduce else 'all-to-all of the ranks\' rows',
                   exchanged_MB_per_sweep=8e-6 * ((self.a_send_elems + self.b_send_elems) if self.owner_reduce else self.n_elems))
        return out

    def local_edges(self):
        return self.plan.flat.E

    def work_fraction(self):
        flat = self.plan.flat
        return float(flat.var_hidden[flat.edge_var].mean())

    def f2v_joint_terms(self):
        return joint_terms(self.plan.flat, self.bp.np_host)

    def heavy_stats(self):
        return self.bp.n_heavy, self.bp.heavy_terms

    def heavy_grid_terms(self):
        return self.bp.heavy_grid_terms


# =================================================================================================
# owner-computes sharding: variables are partitioned, a rank computes every message whose target it owns
# =================================================================================================
def hidden_adjacency(flat):
    """variable-variable adjacency through shared factors, hidden variables only (an observed variable sends and receives no
    message: a factor it shares with a hidden one is never cut), as a symmetric scipy CSR matrix of pair counts"""
    from scipy.sparse import csr_matrix
    V = flat.V
    hid = np.isnan(flat.var_value)
    arity = np.diff(flat.fac_ptr)
    rows, cols = [], []
    for i in range(int(arity.max()) if flat.F else 0):
        for j in range(i + 1, int(arity.max())):
            f = np.flatnonzero(arity > j)
            a, b = flat.edge_var[flat.fac_ptr[f] + i].astype(np.int64), flat.edge_var[flat.fac_ptr[f] + j].astype(np.int64)
            keep = hid[a] & hid[b] & (a != b)
            rows += [a[keep], b[keep]]
            cols += [b[keep], a[keep]]
    if not rows:
        return csr_matrix((V, V), dtype=np.float32)
    r, c = np.concatenate(rows), np.concatenate(cols)
    return csr_matrix((np.ones(r.size, dtype=np.float32), (r, c)), shape=(V, V))


def refine_partition(flat, owner, world, rounds=40, eps=0.03, seed=0):
    """Fewer cut factors for the same balance: rounds of label propagation under a capacity.  Every round counts, for every hidden
    variable, its neighbours per part (one sparse product), proposes the part that holds most of them when that beats its own
    (half of the candidates per round, drawn with a seeded generator: neighbours do not swap places for ever) and accepts the
    proposals in order of gain while the receiving part stays below (1 + eps) of the mean load (load = degree + 1, the work
    measure of ``partition_variables``).  On the benchmark's random 4-regular graph the breadth-first blocks cut 81 % of the
    hidden-hidden factors at 8 parts (a random assignment: 87 %), forty rounds bring that to 40 %: half the rows to exchange,
    half the ghosts to re-draw, three times the work that needs no ghost.  Deterministic (seeded); runs on rank 0 only."""
    from scipy.sparse import csr_matrix
    if world <= 1 or flat.V == 0:
        return owner
    V = flat.V
    A = hidden_adjacency(flat)
    owner = np.asarray(owner, dtype=np.int64).copy()
    w = np.diff(flat.var_ptr).astype(np.float64) + 1.0
    cap = w.sum() / world * (1.0 + eps)
    rng = np.random.default_rng(seed)
    idx = np.arange(V)
    for _ in range(rounds):
        onehot = csr_matrix((np.ones(V, dtype=np.float32), (idx, owner)), shape=(V, world))
        cnt = np.asarray((A @ onehot).todense())
        best = cnt.argmax(axis=1)
        gain = cnt[idx, best] - cnt[idx, owner]
        cand = np.flatnonzero((gain > 0) & (rng.random(V) < 0.5))
        if cand.size == 0:
            break
        order = cand[np.argsort(-gain[cand], kind='stable')]
        load = np.bincount(owner, weights=w, minlength=world)
        moved = 0
        for d in range(world):
            c = order[best[order] == d]
            if c.size == 0:
                continue
            c = c[np.cumsum(w[c]) <= max(cap - load[d], 0.0)]
            if c.size == 0:
                continue
            load[d] += w[c].sum()
            np.subtract.at(load, owner[c], w[c])
            owner[c] = d
            moved += int(c.size)
        if moved == 0:
            break
    return owner.astype(np.int32)


def partition_variables(flat, world, partition='refined'):
    """owner rank of every variable.  'bfs': the breadth-first order of the factor graph cut into `world` blocks of equal total
    degree (a variable's share of the f -> v work is its number of incident edges); 'refined' (default): those blocks improved by
    ``refine_partition`` (same balance to 3 %, about half the cut factors on an expander); 'block': the construction order cut
    into blocks."""
    V = flat.V
    pos = bfs_variable_order(flat) if partition in ('bfs', 'refined') and world > 1 else np.arange(V, dtype=np.int64)
    order = np.argsort(pos, kind='stable')
    w = np.diff(flat.var_ptr).astype(np.int64)[order] + 1
    before = np.cumsum(w) - w
    owner = np.empty(V, dtype=np.int32)
    owner[order] = (before * world // max(int(w.sum()), 1)).astype(np.int32)
    if partition == 'refined' and world > 1:
        owner = refine_partition(flat, owner, world)
    return owner


class OwnerPlan:
    """One rank's part of the owner-computes split of a ground ``FlatGraph`` (pure NumPy, no GPU).

    * the rank OWNS the variables with ``var_owner == rank``; it holds every factor that touches a hidden variable it owns, with
      ALL of that factor's edges (``fac_ids``, ``edge_ids``: global ids, ascending);
    * local variables (``var_gid``): the owned ones first (``n_owned``), then the *ghosts* -- hidden variables owned elsewhere
      that share a factor with an owned hidden one (``n_ghost``) --, then observed variables owned elsewhere; ascending global id
      inside each block.  A local variable's adjacency row lists its local edges in the order of its row in the whole graph, so an
      owned variable -- all of whose factors are local -- is swept in exactly the single-GPU summation order;
    * ``edge_skip``: edges whose f -> v message is not this rank's to compute (their variable is not an owned hidden one);
      ``edge_key``: 1 on the edges of *cut* factors (factors with a ghost), whose f -> v messages need the exchange;
    * exchange, per peer `s` (both ends build the same order, so no index travels): ``send_rows[s]`` = local edges (owned hidden
      variable, cut factor that touches a hidden variable owned by `s`) in ascending (global factor, position) -- their v -> f
      rows go to `s`; ``recv_rows[s]`` = local edges whose variable is a ghost owned by `s`, same order; ``send_q[s]`` / ``recv_q[s]``
      = the continuous variables among the former's / the ghosts owned by `s`, ascending global id -- their proposals.
    """

    def __init__(self, flat, rank, world, var_owner=None, partition='refined'):
        from .flat import build_flat
        if flat.lifted or (flat.edge_canon != np.arange(flat.E)).any():
            raise NotImplementedError('sharding expects a ground graph')
        self.rank, self.world = rank, world
        if var_owner is None:
            var_owner = partition_variables(flat, world, partition)
        var_owner = np.asarray(var_owner)
        hidden = np.isnan(flat.var_value)
        mine_h = (var_owner == rank) & hidden
        arity_all = np.diff(flat.fac_ptr)
        touch = np.maximum.reduceat(mine_h[flat.edge_var].astype(np.int8), flat.fac_ptr[:-1]) if flat.F else np.zeros(0, np.int8)
        touch = np.where(arity_all > 0, touch, 0)
        self.fac_ids = np.flatnonzero(touch)
        arity = arity_all[self.fac_ids]
        local_ptr = np.zeros(self.fac_ids.size + 1, dtype=np.int64)
        np.cumsum(arity, out=local_ptr[1:])
        self.edge_ids = (np.repeat(flat.fac_ptr[self.fac_ids].astype(np.int64) - local_ptr[:-1], arity) +
                         np.arange(int(local_ptr[-1]), dtype=np.int64))
        gvar = flat.edge_var[self.edge_ids]
        gids = np.unique(gvar)
        kind = np.where(var_owner[gids] == rank, 0, np.where(hidden[gids], 1, 2))
        gids = gids[np.argsort(kind, kind='stable')]
        self.n_owned, self.n_ghost = int((kind == 0).sum()), int((kind == 1).sum())
        self.var_gid = gids.astype(np.int64)
        lid = np.full(flat.V, -1, dtype=np.int64)
        lid[gids] = np.arange(gids.size)
        lf = build_flat(local_ptr.astype(np.int32), lid[gvar].astype(np.int32), flat.fac_pot[self.fac_ids], [], flat.var_value[gids],
                        flat.var_dom[gids], flat.domains)
        lf.pot_kind, lf.pot_off, lf.pot_param = flat.pot_kind, flat.pot_off, flat.pot_param
        # adjacency rows in the order of the whole graph's rows (restricted to the local edges)
        loc_edge = np.full(flat.E, -1, dtype=np.int64)
        loc_edge[self.edge_ids] = np.arange(self.edge_ids.size)
        gdeg = np.diff(flat.var_ptr)[gids].astype(np.int64)
        start = np.zeros(gids.size + 1, dtype=np.int64)
        np.cumsum(gdeg, out=start[1:])
        slots = np.repeat(flat.var_ptr[gids].astype(np.int64) - start[:-1], gdeg) + np.arange(int(start[-1]), dtype=np.int64)
        le = loc_edge[flat.var_edge[slots]]
        keep = le >= 0
        owner_of_slot = np.repeat(np.arange(gids.size), gdeg)[keep]
        lf.var_edge = le[keep].astype(np.int32)
        lf.var_ptr = np.zeros(gids.size + 1, dtype=np.int32)
        np.cumsum(np.bincount(owner_of_slot, minlength=gids.size), out=lf.var_ptr[1:])
        self.flat = lf
        lv = lf.edge_var
        is_owned_h = (np.arange(gids.size) < self.n_owned) & np.isnan(lf.var_value)
        is_ghost = (np.arange(gids.size) >= self.n_owned) & (np.arange(gids.size) < self.n_owned + self.n_ghost)
        self.edge_skip = ~is_owned_h[lv]
        fac_cut = np.maximum.reduceat(is_ghost[lv].astype(np.int8), local_ptr[:-1]) if self.fac_ids.size else np.zeros(0, np.int8)
        self.edge_key = np.repeat(fac_cut, arity).astype(np.int32)
        # ---- exchange lists
        efac = lf.edge_fac.astype(np.int64)
        gfid = self.fac_ids[efac]
        epos = lf.edge_pos.astype(np.int64)
        order_key = gfid * (int(arity_all.max()) if flat.F else 1) + epos          # (global factor, position)
        ghost_e = np.flatnonzero(is_ghost[lv])
        owner_e = var_owner[gids[lv]]
        # (owned-hidden edge, peer) pairs: all ordered pairs (i, j) of edges of a cut factor with i owned hidden, j a ghost
        cut_fac = np.flatnonzero(fac_cut)
        pairs_e, pairs_p = [], []
        if cut_fac.size:
            a_max = int(arity[cut_fac].max())
            base = local_ptr[cut_fac]
            ar = arity[cut_fac]
            for i in range(a_max):
                for j in range(a_max):
                    if i == j:
                        continue
                    ok = (i < ar) & (j < ar)
                    ei, ej = base[ok] + i, base[ok] + j
                    sel = is_owned_h[lv[ei]] & is_ghost[lv[ej]]
                    pairs_e.append(ei[sel])
                    pairs_p.append(owner_e[ej[sel]].astype(np.int64))
        pe = np.concatenate(pairs_e) if pairs_e else np.zeros(0, np.int64)
        pp = np.concatenate(pairs_p) if pairs_p else np.zeros(0, np.int64)
        E_l = max(int(lf.E), 1)
        uniq_pairs = np.unique(pp * E_l + pe)
        pp, pe = uniq_pairs // E_l, uniq_pairs % E_l
        self.send_rows, self.recv_rows, self.send_q, self.recv_q = {}, {}, {}, {}
        cont = lf.var_cont
        for s in range(world):
            if s == rank:
                continue
            e = pe[pp == s]
            self.send_rows[s] = e[np.argsort(order_key[e], kind='stable')].astype(np.int64)
            vq = np.unique(lv[e])
            vq = vq[cont[vq]]
            self.send_q[s] = vq[np.argsort(gids[vq], kind='stable')].astype(np.int64)
            e = ghost_e[owner_e[ghost_e] == s]
            self.recv_rows[s] = e[np.argsort(order_key[e], kind='stable')].astype(np.int64)
            vq = np.flatnonzero(is_ghost & (var_owner[gids] == s) & cont)
            self.recv_q[s] = vq[np.argsort(gids[vq], kind='stable')].astype(np.int64)

    def layout(self, n, np_host=None):
        """element offsets of the two buffers of the one all_to_all.  Per peer (ascending rank): the rows of the CONTINUOUS
        variables' edges first -- `n` doubles each --, then the rows of the discrete ones (their number of states each), then the
        proposals (two doubles each), then padding up to a multiple of `n` doubles -- so that every peer's block, and with it
        every continuous row, starts a whole number of rows from the buffer's base: the receive buffer is laid behind the E
        message rows of the rank's v -> f array and a continuous ghost edge's row is READ WHERE IT ARRIVED (``ghost_rows``: its
        ``edge_canon``), no unpack pass; only the short discrete rows and the proposals are scattered after the exchange.
        Inside a block rows keep the ascending (global factor, position) order on both ends."""
        lf = self.flat
        lv = lf.edge_var
        cont = lf.var_cont
        width = np.where(cont, n, lf.var_nstates).astype(np.int64)         # doubles in a variable's v -> f row
        out = {}
        for side, rows, qs in (('send', self.send_rows, self.send_q), ('recv', self.recv_rows, self.recv_q)):
            row_edge, row_off, row_w, q_var, q_off, counts = [], [], [], [], [], []
            c_edge, c_off = [], []
            off = 0
            for s in range(self.world):
                if s == self.rank:
                    counts.append(0)
                    continue
                e = rows[s]
                ec, ed = e[cont[lv[e]]], e[~cont[lv[e]]]                    # (a boolean cut of a sorted list keeps its order)
                c_edge.append(ec), c_off.append(off + n * np.arange(ec.size, dtype=np.int64))
                end = off + n * int(ec.size)
                w = width[lv[ed]]
                row_edge.append(ed), row_off.append(end + np.cumsum(w) - w), row_w.append(w)
                end += int(w.sum())
                v = qs[s]
                q_var.append(v), q_off.append(end + 2 * np.arange(v.size, dtype=np.int64))
                end += 2 * int(v.size)
                end = off + -(-(end - off) // n) * n                        # pad the block to whole rows
                counts.append(end - off)
                off = end
            cat = lambda xs, dt: (np.concatenate(xs) if xs else np.zeros(0)).astype(dt)
            out[side] = dict(cont_edge=cat(c_edge, np.int32), cont_off=cat(c_off, np.int64),
                             row_edge=cat(row_edge, np.int32), row_off=cat(row_off, np.int64), row_width=cat(row_w, np.int32),
                             q_var=cat(q_var, np.int32), q_off=cat(q_off, np.int64), counts=counts, size=off)
        return out

    @staticmethod
    def rows_of(side_layout, n):
        """(edge, offset, width) of EVERY row of one side of ``layout``: the continuous rows, then the discrete ones"""
        L = side_layout
        return (np.concatenate([L['cont_edge'], L['row_edge']]).astype(np.int64), np.concatenate([L['cont_off'], L['row_off']]),
                np.concatenate([np.full(L['cont_edge'].size, n, dtype=np.int64), L['row_width'].astype(np.int64)]))

    def ghost_rows(self, n):
        """``edge_canon`` of the local graph with every continuous ghost edge pointing at the row of the v -> f array its message
        arrives in: row E + (offset in the receive buffer) / n (``layout``).  (A ghost edge's own row is never written: its
        v -> f message is its owner's to compute.)  Returns (edge_canon, rows of the receive buffer)."""
        lay = self.layout(n)['recv']
        canon = np.arange(self.flat.E, dtype=np.int32)
        canon[lay['cont_edge']] = (self.flat.E + lay['cont_off'] // n).astype(np.int32)
        return canon, -(-lay['size'] // n)


class OwnerRunner:
    """One rank's part of the owner-computes particle sweep (EPBP semantics), ``bench.py --exchange ownercompute``.

    Per sweep: v -> f and the proposal update of the OWNED variables (every one of them is swept once, over all its factors, in
    the single-GPU order) -> the v -> f rows of the cut edges and the proposals of the variables that are ghosts elsewhere are
    packed (``lhvi_pbp_halo_pack``) -> ONE all_to_all, started asynchronously -> while it is in flight: new particles of the owned
    variables, f -> v of the factors without a ghost -> wait -> ``lhvi_pbp_halo_unpack`` puts the peers' rows into the v -> f
    rows of the ghost edges and their proposals into q -> new particles of the ghosts (the sampler is keyed by the global id: the
    same bits as at their owner) -> f -> v of the cut factors towards the owned variables.  No kernel reads boundary rows, no
    per-variable work is replicated, and every array equals the single-GPU run's bit for bit."""

    def __init__(self, flat, n, seed, rank, world, proposal_approximation='simple', group=None, overlap=True, var_owner=None):
        import torch
        from .pbp import EPBP
        self.plan = plan = OwnerPlan(flat, rank, world, var_owner=var_owner)
        self.rank, self.world, self.group = rank, world, group
        # the peers' continuous rows are read where they arrive: the receive buffer lies behind the E rows of the v -> f array and
        # the ghost edges' canonical rows point into it (OwnerPlan.ghost_rows) -- set before the work lists are described
        lay = self.lay = plan.layout(n)
        canon, halo_rows = plan.ghost_rows(n)
        plan.flat.edge_canon = canon
        bp = EPBP(None, n=n, proposal_approximation=proposal_approximation, sampler='device', seed=seed)
        # (the v -> f half and the proposal update address the OWNED variables through the same per-variable lists as a single-GPU
        # run -- sixteen binary variables per wavefront, one record per continuous one -- not through a range of all local rows)
        bp._setup(None, flat=plan.flat, edge_key=plan.edge_key, edge_skip=plan.edge_skip, owned=plan.n_owned)
        self.bp = bp
        bp.var_gid = _abi.to_dev(plan.var_gid)
        self.n = n
        self.n_owned, self.n_ghost = plan.n_owned, plan.n_ghost
        self.overlap = bool(overlap) and world > 1
        dev = bp.dg.device
        E = int(plan.flat.E)
        bp.v2f = torch.zeros(E + max(halo_rows, 1), n, dtype=torch.float64, device=dev)
        self.recv = bp.v2f.view(-1)[E * n:E * n + max(lay['recv']['size'], 1)]
        self.send = torch.zeros(max(lay['send']['size'], 1), dtype=torch.float64, device=dev)
        up = {}
        for side in ('send', 'recv'):
            for k in ('row_edge', 'row_off', 'row_width', 'q_var', 'q_off'):
                a = lay[side][k]
                up[side + '_' + k] = a if a.size else np.zeros(1, dtype=a.dtype)
        # where the v -> f half writes a copy of a cut edge's row as it forms it (lhvi_pbp_t.halo_off): its place in the send buffer.
        # An edge listed for more than one peer (a cut factor of three or more variables with ghosts of several owners) keeps its
        # first place there; the others are filled by the pack pass (`send_more_*`), like the proposals
        S = lay['send']
        all_edge = np.concatenate([S['cont_edge'], S['row_edge']]).astype(np.int64)
        all_off = np.concatenate([S['cont_off'], S['row_off']])
        all_w = np.concatenate([np.full(S['cont_edge'].size, n, dtype=np.int32), S['row_width']])
        halo = np.full(max(E, 1), -1, dtype=np.int64)
        first = np.unique(all_edge, return_index=True)[1]
        halo[all_edge[first]] = all_off[first]
        more = np.setdiff1d(np.arange(all_edge.size), first)
        up['halo_off'] = halo
        for k, a in (('send_more_edge', all_edge[more].astype(np.int32)), ('send_more_off', all_off[more]), ('send_more_width', all_w[more])):
            up[k] = a if a.size else np.zeros(1, dtype=a.dtype)
        self.n_send_more = int(more.size)
        # the sampler's lists (lhvi_pbp_t.resample_vars): the owned and the ghost hidden continuous variables, two per wavefront
        rr = bp.resample_vars.cpu().numpy() if bp.resample_vars is not None else np.zeros((0, 8), dtype=np.int32)
        self.res_lists = None
        if n <= 64 and rr.shape[0]:
            own = rr[:, 0] < plan.n_owned
            up['res_owned'] = rr[own] if own.any() else np.zeros((1, 8), dtype=np.int32)
            up['res_ghost'] = rr[~own] if (~own).any() else np.zeros((1, 8), dtype=np.int32)
            self.res_lists = (int(own.sum()), int((~own).sum()))
        self.idx = _abi.upload(up)
        self.counts = list(lay['send']['counts'])              # what a loopback group cuts this rank's send buffer by
        self.record_phases = False
        self._phase_events = []
        self.f2v_extra = []
        # Two f -> v launches per sweep (factors without a ghost while the rows are in flight, the cut factors after them) pay when
        # the first one hides a good part of the exchange.  With many ranks nearly every factor of an expander is cut: the first
        # launch then hides next to nothing and costs a second ramp-up and tail of the persistent kernels (0.35 of 1.75 ms at 8
        # ranks) -- below `min_interior` of the f -> v work ONE launch after the exchange serves both lists.
        pc = bp.part_counts
        total = bp.n_heavy + bp.n_pair + int(bp.generic_edges.numel()) + int(bp.fast_edges.numel()) + bp.n_small16 + bp.n_small32 + bp.n_cq
        inner = pc['heavy'] + pc['pair'] + pc['generic'] + pc['fast'] + pc['small16'] + pc['small32'] + pc['cq']
        self.interior_fraction = inner / float(max(total, 1))
        self.split_f2v = self.interior_fraction >= self.min_interior

    min_interior = 0.25         # least share of the f -> v work lists that must need no ghost for the two-launch schedule

    def message_rows(self):
        """the v -> f array with every edge's row at its own index ([E, n]): the continuous ghost edges' rows gathered from where
        they arrived (tests compare this with the single-GPU array)"""
        import torch
        E = int(self.plan.flat.E)
        canon = _abi.to_dev(self.plan.flat.edge_canon.astype(np.int64))
        return self.bp.v2f.index_select(0, canon)[:E]

    # ---- structs ------------------------------------------------------------------------------------------------------------
    def _struct(self, lo=0, hi=0, leave_room=False):
        s = self.bp._struct()
        if self.overlap and leave_room:                      # only the launches that run beside the collective leave it room
            s.flags |= _abi.PBP_LEAVE_ROOM
        s.prop_desc, s.n_prop_desc = None, 0                 # variables are addressed by range: owned [0, n_owned), ghosts after them
        s.prop_hub, s.n_prop_hub, s.prop_partial = None, 0, None
        s.v2f_wide, s.n_v2f_wide, s.v2f_narrow, s.n_v2f_narrow, s.v2f_hub, s.n_v2f_hub = None, 0, None, 0, None, 0
        s.v2f_mid16, s.n_v2f_mid16, s.v2f_mid32, s.n_v2f_mid32 = None, 0, None, 0
        s.var_lo, s.var_hi = int(lo), int(hi)
        return s

    _edge_part = ShardedRunner._edge_part

    def init(self):
        bp = self.bp
        _abi.check(_abi.lib().lhvi_pbp_init(bp.dg.g, self._struct(), _abi.ptr(bp.eta), _abi.ptr(bp.q_dev), _abi.ptr(bp.f2v),
                                            _abi.ptr(bp.v2f), _abi.stream_ptr()))
        bp._generate_sample()

    # ---- the phases of a sweep ----------------------------------------------------------------------------------------------
    def owned_half(self):
        """v -> f and the proposal update of the owned variables.  The rows the peers need go into the send buffer as the v -> f
        half forms them (``halo_off``); a small pass adds the proposals of the owned variables that are ghosts elsewhere"""
        bp, l, st = self.bp, _abi.lib(), _abi.stream_ptr()
        if self.n_owned:
            s = bp._struct()                                  # the owned variables' lists (EPBP._setup(owned=...))
            if bp.v2f_lists is None:
                s.var_lo, s.var_hi = 0, int(self.n_owned)
            s.halo_off, s.halo_buf = _abi.ptr(self.idx['halo_off']), _abi.ptr(self.send)
            _abi.check(l.lhvi_pbp_v2f(bp.dg.g, s, _abi.ptr(bp.f2v), _abi.ptr(bp.v2f), st))
            s.halo_off = s.halo_buf = None
            s.var_lo, s.var_hi = (0, 0) if s.prop_desc else (0, int(self.n_owned))
            _abi.check(l.lhvi_pbp_proposal(bp.dg.g, s, _abi.ptr(bp.f2v), _abi.ptr(bp.eta), _abi.ptr(bp.q_dev), st))
        L, ix = self.lay['send'], self.idx
        _abi.check(l.lhvi_pbp_halo_pack(_abi.ptr(bp.v2f), self.n, self.n_send_more, _abi.ptr(ix['send_more_edge']),
                                        _abi.ptr(ix['send_more_off']), _abi.ptr(ix['send_more_width']), _abi.ptr(bp.q_dev),
                                        int(L['q_var'].size), _abi.ptr(ix['send_q_var']), _abi.ptr(ix['send_q_off']), _abi.ptr(self.send), st))
        return self.send[:L['size']]

    def _resample(self, lo, hi):
        bp = self.bp
        if hi <= lo:
            return
        s = self._struct(lo, hi)
        if self.res_lists is not None:                       # the listed form: hidden continuous variables only (bit-identical draws)
            which = 0 if lo == 0 else 1
            if self.res_lists[which] == 0:
                return
            s.var_lo, s.var_hi = 0, 0
            s.resample_vars, s.n_resample_vars = _abi.ptr(self.idx['res_owned' if which == 0 else 'res_ghost']), self.res_lists[which]
            if which == 1:
                s.flags |= _abi.PBP_NO_UNIQ                    # a ghost's first-occurrence mask is read by nobody here (no v -> f of its own)
        _abi.check(_abi.lib().lhvi_pbp_resample_uniq(bp.dg.g, s, _abi.ptr(bp.var_gid), int(bp.seed), int(bp._draws - 1),
                                                     _abi.ptr(bp.particles), _abi.ptr(bp.uniq), _abi.stream_ptr()))

    def interior(self, f2v_events=None):
        """what needs nothing from the peers: the owned variables' new particles, f -> v of the factors without a ghost"""
        bp = self.bp
        bp.old_particles, bp.particles = bp.particles, bp.old_particles
        bp._draws += 1
        bp._views = {}
        self._resample(0, self.n_owned)
        if self.split_f2v:
            bp._launch_f2v(self._edge_part(self._struct(leave_room=True), 0), f2v_events)

    def boundary(self, recv, f2v_events=None):
        """the rest, once the peers' rows have arrived (after `interior`)"""
        bp, l, st = self.bp, _abi.lib(), _abi.stream_ptr()
        L, ix = self.lay['recv'], self.idx
        if recv.data_ptr() != self.recv.data_ptr():            # (a loopback group's copy; a real collective wrote the rows in place)
            self.recv[:recv.shape[0]].copy_(recv)
        # the continuous rows are read where they are; the short rows of discrete ghosts and the ghosts' proposals are scattered
        _abi.check(l.lhvi_pbp_halo_unpack(_abi.ptr(self.recv), self.n, int(L['row_edge'].size), _abi.ptr(ix['recv_row_edge']),
                                          _abi.ptr(ix['recv_row_off']), _abi.ptr(ix['recv_row_width']), _abi.ptr(bp.v2f),
                                          int(L['q_var'].size), _abi.ptr(ix['recv_q_var']), _abi.ptr(ix['recv_q_off']), _abi.ptr(bp.q_dev), st))
        self._resample(self.n_owned, self.n_owned + self.n_ghost)
        bp._launch_f2v(self._edge_part(self._struct(), 1) if self.split_f2v else self._struct(), f2v_events)

    def exchange(self, send, async_op=False):
        """the one collective of a sweep"""
        import torch.distributed as td
        L = self.lay
        recv = self.recv[:L['recv']['size']]
        work = None
        if td.get_backend(self.group) == 'nccl':
            work = td.all_to_all_single(recv, send, output_split_sizes=L['recv']['counts'], input_split_sizes=L['send']['counts'],
                                        async_op=async_op, group=self.group)
        else:                                               # rehearsal backend (gloo): same collective on host copies
            h_send = send.cpu()
            h_recv = h_send.new_empty(L['recv']['size'])
            td.all_to_all_single(h_recv, h_send, output_split_sizes=L['recv']['counts'], input_split_sizes=L['send']['counts'], group=self.group)
            recv.copy_(h_recv)
        return (recv, work) if async_op else recv

    def sweep(self, f2v_events=None):
        import torch
        marks = []

        def mark():
            if self.record_phases:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                marks.append(ev)
        mark()
        send = self.owned_half()
        mark()
        if self.world > 1:
            recv, work = self.exchange(send, async_op=self.overlap)
        else:
            recv, work = self.recv[:0], None
        extra = None
        if f2v_events and self.split_f2v:
            extra = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self.f2v_extra.append(extra)
        self.interior(extra)
        mark()
        if work is not None:
            work.wait()
        mark()
        self.boundary(recv, f2v_events)
        mark()
        if marks:
            self._phase_events.append(marks)

    def phase_ms(self):
        if not self._phase_events:
            return None
        names = ('owned_v2f_proposal_pack', 'interior', 'exchange_wait', 'boundary')
        acc = {k: 0.0 for k in names}
        for m in self._phase_events:
            for k, a, b in zip(names, m[:-1], m[1:]):
                acc[k] += a.elapsed_time(b)
        n = len(self._phase_events)
        out = {k: v / n for k, v in acc.items()}
        out.update(sweeps=n, owned_variables=int(self.n_owned), ghost_variables=int(self.n_ghost),
                   interior_fraction_of_f2v_lists=round(self.interior_fraction, 4), f2v_launches_per_sweep=2 if self.split_f2v else 1,
                   cut_edge_rows_sent=int(self.lay['send']['row_edge'].size + self.lay['send']['cont_edge'].size), exchange='owner computes: v->f rows of cut edges + ghost proposals',
                   exchanged_MB_per_sweep=8e-6 * self.lay['send']['size'])
        return out

    def local_edges(self):
        return int((~self.plan.edge_skip).sum())

    def work_fraction(self):
        return 1.0

    def f2v_joint_terms(self):
        return joint_terms(self.plan.flat, self.bp.np_host, ~self.plan.edge_skip)

    def heavy_stats(self):
        return self.bp.n_heavy, self.bp.heavy_terms

    def heavy_grid_terms(self):
        return self.bp.heavy_grid_terms
