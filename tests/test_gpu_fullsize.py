"""GPU, BASELINE.json's full size (10M-edge hybrid MRF, n=64, T=32): size-independent properties of the particle sweep
instead of an element-wise oracle comparison (the CPU oracle needs ~15 minutes per sweep at this size)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

E_FULL = 10_000_000


def _run(flat, sweeps, seed=1):
    from lhvi import _abi, dist
    from lhvi.pbp import EPBP
    bp = EPBP(None, n=64, proposal_approximation='simple', sampler='device', seed=seed)
    bp._setup(None, flat=flat)
    r = dist.SingleRunner(bp)
    r.init()
    for _ in range(sweeps):
        r.sweep()
    return bp


def test_full_size_sweep_invariants():
    import torch
    from lhvi import _abi, synth
    _abi.require_gpu()
    flat = synth.hybrid_mrf_flat(V=E_FULL // 4, deg=4, seed=0)
    assert flat.E == E_FULL
    bp = _run(flat, 2)
    dev = bp.dg.device
    hid_v = torch.from_numpy(flat.var_hidden).to(dev)
    edge_var = bp.dg.t['edge_var'].long()
    hid_e = hid_v[edge_var]
    npv = bp.np_dev.long()[edge_var]
    n = bp.n
    # (1) log_message_balance: every v->f message has zero mean over the variable's distinct particles (EPBP:204-215),
    #     unless the max-700 branch fired; checked on ALL 9M hidden edges on the device
    col = torch.arange(n, device=dev)[None, :]
    # the v->f messages of the last sweep live on the particles that are now `old_particles`: rebuild their mask
    uniq_old = torch.empty_like(bp.uniq)
    _abi.check(_abi.lib().lhvi_pbp_uniq(bp.dg.g, n, _abi.ptr(bp.old_particles), _abi.ptr(bp.np_dev), _abi.ptr(uniq_old),
                                        _abi.stream_ptr()))
    uq = uniq_old[edge_var].bool() & (col < npv[:, None])
    v2f = bp.v2f
    cnt = uq.sum(1).clamp_min(1)
    mean = (v2f * uq).sum(1) / cnt
    mx = torch.where(uq, v2f, torch.full_like(v2f, -1e300)).max(1).values
    max_branch = hid_e & ((mx - 700.0).abs() < 1e-9)          # shift = max - 700 (a particle sits on a domain bound: log 1e-200)
    by_mean = hid_e & ~max_branch
    assert float(mean[by_mean].abs().max()) < 1e-9 and float(mx[by_mean].max()) < 700.0
    assert bool((mean[max_branch] < -1e-9).all())             # the branch fires only when max - mean > 700
    # (2) f->v tables are finite, floored at -700 only where the sum underflowed, and untouched for observed targets
    f2v = bp.f2v
    assert bool(torch.isfinite(f2v[hid_e]).all())
    assert float(f2v[hid_e].min()) >= -800.0
    assert float(f2v[~hid_e].abs().max()) == 0.0
    # (3) proposals: positive finite variances, clamped from below by deg * var_threshold (EPBP:93-95)
    cont = torch.from_numpy(flat.var_hidden & flat.var_cont).to(dev)
    q = bp.q_dev[cont]
    assert bool(torch.isfinite(q).all()) and float(q[:, 1].min()) > 0
    deg = torch.from_numpy(np.diff(flat.var_ptr)).to(dev)[edge_var][torch.from_numpy(flat.var_cont[flat.edge_var] & flat.var_hidden[flat.edge_var]).to(dev)]
    eta = bp.eta[torch.from_numpy(flat.var_cont[flat.edge_var] & flat.var_hidden[flat.edge_var]).to(dev)]
    assert bool((eta[:, 1] >= 3.0 * deg - 1e-9).all())
    # (4) determinism: the same seed reproduces the run bit for bit (counter-based sampler, no atomics in the sweep)
    bp2 = _run(flat, 2)
    assert bool((bp2.q_dev == bp.q_dev).all()) and bool((bp2.f2v == bp.f2v).all())
    checksum = float(bp.q_dev[cont].sum())
    del bp2
    # (5) relabelling invariance: listing the factors in another order (new edge ids, new rv.nb order) changes the
    #     summation order only -- proposals agree to rounding
    rng = np.random.default_rng(5)
    perm = rng.permutation(flat.F)
    from lhvi.flat import build_flat
    a, b = flat.edge_var[0::2][perm], flat.edge_var[1::2][perm]
    specs = [(int(k), flat.pot_param[flat.pot_off[i]:flat.pot_off[i + 1]].tolist()) for i, k in enumerate(flat.pot_kind)]
    flat_p = build_flat(flat.fac_ptr, np.stack([a, b], axis=1).ravel(), flat.fac_pot[perm], specs, flat.var_value, flat.var_dom,
                        flat.domains)
    del bp
    torch.cuda.empty_cache()
    bp3 = _run(flat_p, 2)
    q3 = bp3.q_dev[cont]
    assert abs(float(q3.sum()) - checksum) <= 1e-9 * abs(checksum)


def _independent_stability_check(flat, rvc, fc):
    """Is (rvc, fc) a fixed point of colour passing?  Checked on the device with torch, independently of csrc/color.hip:
    factors exactly (cluster id and the scope's cluster ids packed into one int64 -- as many distinct keys as clusters means
    every cluster is uniform), variables through two 64-bit multiset hashes of the incident factor colours built with other
    mixing constants than the kernels' fingerprints (a wrong merge would need a simultaneous 128-bit collision here too)"""
    import torch
    dev = 'cuda'
    ev = torch.from_numpy(flat.edge_var.astype(np.int64)).to(dev)
    ef = torch.from_numpy(flat.edge_fac.astype(np.int64)).to(dev)
    r = torch.from_numpy(np.asarray(rvc, dtype=np.int64)).to(dev)
    f = torch.from_numpy(np.asarray(fc, dtype=np.int64)).to(dev)
    n_rv, n_f = int(r.max()) + 1, int(f.max()) + 1
    assert n_rv < (1 << 20) and n_f < (1 << 22) and bool((torch.from_numpy(np.diff(flat.fac_ptr)) == 2).all())
    key = (f << 40) | (r[ev[0::2]] << 20) | r[ev[1::2]]
    assert int(torch.unique(key).numel()) == n_f

    def mix(x, c1, c2):                       # splitmix-style, wrapping int64 arithmetic
        x = (x ^ (x >> 31)) * c1
        x = (x ^ (x >> 29)) * c2
        return x ^ (x >> 32)
    fe = f[ef]
    h1 = torch.zeros(flat.V, dtype=torch.int64, device=dev).index_add_(0, ev, mix(fe + 0x1F3D5B79, -0x61C8864680B583EB, 0x2545F4914F6CDD1D))
    h2 = torch.zeros(flat.V, dtype=torch.int64, device=dev).index_add_(0, ev, mix(fe * 3 + 0x7ED55D16, 0x369DEA0F31A53F85, -0x4B47D0C5B9E0A4C7))
    rows = torch.stack([r, h1, h2], dim=1)
    assert int(torch.unique(rows, dim=0).shape[0]) == n_rv
    return n_rv, n_f


def test_cfg5_rgm_at_10m_ground_edges_lifts_to_10k_clusters():
    """BASELINE.json cfg 5 at its stated shape: the RGM template at 10.0 M ground edges whose evidence makes colour passing
    converge to ~10 k rv clusters; refinement on the device, then lifted VI (K = 2, T = 3).
    Size-independent checks: the partition is a fixed point (independent re-hash), cluster sizes and counts add up to the
    ground degrees, the 1/25-scale twin (same evidence classes) has the same number of clusters and its partition equals
    the exact Python colour passing bit for bit, the lifted free energy equals the ground one on the twin (1e-8) and
    decreases under ADAM at full size."""
    import torch
    from lhvi import _abi, lifting, synth
    from lhvi.vi import VarInference
    from oracle import oracle
    _abi.require_gpu()
    flat, sym, rv0, f0 = synth.rgm_structured_flat()
    assert flat.E == 10_004_000
    dg = _abi.DeviceGraph(flat)
    st = {}
    rvc, fc = lifting.refine_flat(flat, sym, rv0, f0, dg=dg, stats=st)
    n_rv, n_f = _independent_stability_check(flat, rvc, fc)
    assert (n_rv, n_f) == (9956, 19630) and st['rounds'] <= 8
    # one more round changes nothing
    rvc2, fc2 = lifting.refine_flat(flat, sym, rvc, fc, dg=dg)       # (a round renumbers the colours: compare the partitions)
    assert int(rvc2.max()) + 1 == n_rv and np.unique(np.stack([rvc, rvc2], 1), axis=0).shape[0] == n_rv
    assert int(fc2.max()) + 1 == n_f and np.unique(np.stack([fc, fc2], 1), axis=0).shape[0] == n_f
    del dg
    # clusters never mix initial colours (hidden / evidence value, potential)
    assert np.unique(np.stack([rvc, rv0], 1), axis=0).shape[0] == n_rv and np.unique(np.stack([fc, f0], 1), axis=0).shape[0] == n_f
    lflat = lifting.lift_flat(flat, rvc, fc)
    assert lflat.V == n_rv and lflat.F == n_f
    # the same lifted graph when the O(V + F) reductions run on the device from device-resident colours
    dg = _abi.DeviceGraph(flat)
    rvc_d, fc_d = lifting.refine_flat(flat, sym, rv0, f0, dg=dg, device_out=True)
    ld = lifting.lift_flat(flat, rvc_d, fc_d, dg=dg)
    del dg
    for name in ('fac_ptr', 'edge_var', 'edge_canon', 'var_ptr', 'var_edge', 'edge_count', 'var_value', 'var_dom', 'var_mult',
                 'fac_mult', 'fac_pot'):
        np.testing.assert_array_equal(getattr(ld, name), getattr(lflat, name), err_msg=name)
    # count / N bookkeeping: a cluster's counts add up to its representative's ground degree, and over all clusters to E
    deg = np.diff(flat.var_ptr)
    rep = np.full(n_rv, flat.V, dtype=np.int64)
    np.minimum.at(rep, rvc, np.arange(flat.V))
    N = np.add.reduceat(lflat.edge_count[lflat.var_edge], lflat.var_ptr[:-1])
    np.testing.assert_array_equal(N, deg[rep])
    assert float((lflat.var_mult * N).sum()) == flat.E and lflat.var_mult.sum() == flat.V and lflat.fac_mult.sum() == flat.F
    assert (deg == deg[rep][rvc]).all()                           # members of a cluster have the same degree
    # lifted VI at full size: the free energy goes down under ADAM
    vi = VarInference(None, 2, 3)
    vi._setup_flat(lflat)
    np.random.seed(0)
    vi.init_param()
    fe0 = vi.free_energy()
    vi.is_log, vi.log_fe = True, True
    vi.time_log, vi.total_time = [], 0
    vi.alpha, vi.b1, vi.b2, vi.eps, vi.t = 0.1, 0.9, 0.999, 1e-8, 0
    vi.ADAM_update(20)
    fes = [fe for _, fe in vi.time_log]
    assert np.isfinite(fes).all() and fes[-1] < fe0 and fes[-1] < fes[4]
    # ---- the 1/25-scale twin (one market per class, one revenue per class)
    small, sym_s, rv0_s, f0_s = synth.rgm_structured_flat(400, 250)
    rs, fs = lifting.refine_flat(small, sym_s, rv0_s, f0_s)
    assert (int(rs.max()) + 1, int(fs.max()) + 1) == (n_rv, n_f)
    ro, fo = oracle.color_passing(small, sym_s, rv0_s, f0_s)                  # exact Python restatement (CGWO:264-271)
    assert oracle.canonical_labels(rs) == oracle.canonical_labels(ro) and oracle.canonical_labels(fs) == oracle.canonical_labels(fo)
    lsmall = lifting.lift_flat(small, rs, fs)
    lv = VarInference(None, 2, 3)
    lv._setup_flat(lsmall)
    np.random.seed(1)
    lv.init_param()
    gv = VarInference(None, 2, 3)
    gv._setup_flat(small)
    gv._upload_params(lv._dev['w_tau'].cpu().numpy(), lv._dev['eta_c'].cpu().numpy()[rs], lv._dev['tau_d'].cpu().numpy()[rs])
    fl, fg = lv.free_energy(), gv.free_energy()
    assert fl == pytest.approx(fg, rel=1e-8)
    # and stays equal along the optimisation: lifted ADAM steps, parameters broadcast to the ground graph
    lv.is_log = False
    lv.alpha, lv.b1, lv.b2, lv.eps, lv.t = 0.1, 0.9, 0.999, 1e-8, 0
    lv.ADAM_update(5)
    gv._upload_params(lv._dev['w_tau'].cpu().numpy(), lv._dev['eta_c'].cpu().numpy()[rs], lv._dev['tau_d'].cpu().numpy()[rs])
    assert lv.free_energy() == pytest.approx(gv.free_energy(), rel=1e-8) and lv.free_energy() < fl


def _lifted_sweeps(lflat, n, sweeps, sampler='device', seed=3):
    from lhvi import _abi
    from lhvi.pbp import HybridLBP
    bp = HybridLBP.on_flat(lflat, n=n, proposal_approximation='simple', sampler=sampler, seed=seed)
    bp._setup(None, flat=lflat)
    _abi.check(_abi.lib().lhvi_pbp_init(bp.dg.g, bp._struct(), _abi.ptr(bp.eta), _abi.ptr(bp.q_dev), _abi.ptr(bp.f2v),
                                        _abi.ptr(bp.v2f), _abi.stream_ptr()))
    bp._generate_sample()
    for _ in range(sweeps):
        bp.sweep(last=False)
    return bp


def test_lifted_particle_sweep_at_scale():
    """The counted sweep of HybridLBP (HLBP:182-215: count-weighted sums, the own factor with count - 1; Demo/RGM/demo.py:19-20
    runs it with n = 10) on array-lifted graphs:
    (a) the 1/25 twin of cfg 5 (400 k ground edges -> 39 260 lifted): with the same particles per cluster the lifted run equals the
        GROUND run of the same semantics (every variable its own cluster, all counts 1) member for member -- the partition is a
        fixed point of colour refinement, so lifting is exact;
    (b) the cfg-5 graph itself (10.0 M ground edges -> the same 39 260 lifted edges with 25 x larger multiplicities, counts up to
        2 000) and a 10 M-edge graph that lifts to ~1 M edges: finite tables, balanced messages, clamped sites, determinism."""
    import torch
    from lhvi import _abi, lifting, synth
    _abi.require_gpu()
    n, sweeps = 10, 3
    # ---- (a) lifted == ground on the twin
    small, sym, rv0, f0 = synth.rgm_structured_flat(400, 250)
    rvc, fc = lifting.refine_flat(small, sym, rv0, f0)
    lflat = lifting.lift_flat(small, rvc, fc)
    assert lflat.E == 39260 and lflat.edge_count.max() > 1
    ground = lifting.lift_flat(small, np.arange(small.V), np.arange(small.F))       # HLBP semantics, nothing merged
    rng = np.random.default_rng(4)
    draws = []

    def lifted_sampler(k, flat, q):
        cont = flat.var_hidden & flat.var_cont
        out = np.zeros((flat.V, n))
        lo, hi = flat.dom_lo[flat.var_dom], flat.dom_hi[flat.var_dom]
        out[cont] = np.clip(rng.standard_normal((int(cont.sum()), n)) * np.sqrt(q[cont, 1:2]) + q[cont, 0:1], lo[cont, None], hi[cont, None])
        draws.append(out)
        return out
    bl = _lifted_sweeps(lflat, n, sweeps, sampler=lifted_sampler)
    assert bl.n_heavy_class > 0 and bl.T == 100                       # the RGM's 100 integral points: n + T = 110 output points per edge
    bg = _lifted_sweeps(ground, n, sweeps, sampler=lambda k, flat, q: draws[k][rvc])
    hid = small.var_hidden
    ql, qg = bl.q_dev.cpu().numpy(), bg.q_dev.cpu().numpy()
    np.testing.assert_allclose(qg[hid], ql[rvc][hid], rtol=1e-9, atol=1e-11)
    # log-beliefs at the particles: sum over a ground variable's factors == count-weighted sum over its cluster's edges
    bg._stable_partition = bl._stable_partition = True
    gb = bg.belief_rv_all(bg.particles).cpu().numpy()
    lb = bl.belief_rv_all(bl.particles).cpu().numpy()
    np.testing.assert_allclose(gb[hid], lb[rvc][hid], rtol=1e-9, atol=1e-7)
    del bg, bl
    # ---- (b) invariants at scale
    for name, args, lifted_edges in (('cfg5', (2000, 1250, 400, 250), 39260), ('1M lifted edges', (2000, 1250, 1000, 250, True), None)):
        flat, sym, rv0, f0 = synth.rgm_structured_flat(*args)
        dg = _abi.DeviceGraph(flat)
        rd, fd = lifting.refine_flat(flat, sym, rv0, f0, dg=dg, device_out=True)
        lf = lifting.lift_flat(flat, rd, fd, dg=dg)
        del dg
        assert lifted_edges is None or lf.E == lifted_edges
        assert lf.E < flat.E / 5 and (lifted_edges is not None or lf.E > 500_000) and float((lf.edge_count[lf.var_edge] * lf.var_mult[np.repeat(np.arange(lf.V), np.diff(lf.var_ptr))]).sum()) == flat.E
        bp = _lifted_sweeps(lf, n, sweeps)
        dev = bp.dg.device
        edge_var = bp.dg.t['edge_var'].long()
        hid_e = torch.from_numpy(lf.var_hidden).to(dev)[edge_var]
        assert bool(torch.isfinite(bp.f2v[hid_e]).all()) and bool(torch.isfinite(bp.v2f[hid_e]).all())
        cont = torch.from_numpy(lf.var_hidden & lf.var_cont).to(dev)
        q = bp.q_dev[cont]
        assert bool(torch.isfinite(q).all()) and float(q[:, 1].min()) > 0
        # sites clamped from below by var_threshold * (number of ground factors of a member) (HLBP:104)
        N = torch.from_numpy(np.add.reduceat(lf.edge_count[lf.var_edge], lf.var_ptr[:-1])).to(dev)
        ce = cont[edge_var]
        assert bool((bp.eta[ce][:, 1] >= 5.0 * N[edge_var][ce] - 1e-9).all())
        # v -> f messages balanced over the distinct particles (HLBP:225-236) unless the max - 700 branch fired
        uq_old = torch.empty_like(bp.uniq)
        _abi.check(_abi.lib().lhvi_pbp_uniq(bp.dg.g, n, _abi.ptr(bp.old_particles), _abi.ptr(bp.np_dev), _abi.ptr(uq_old), _abi.stream_ptr()))
        uq = uq_old[edge_var].bool() & (torch.arange(n, device=dev)[None, :] < bp.np_dev.long()[edge_var][:, None])
        mean = (bp.v2f * uq).sum(1) / uq.sum(1).clamp_min(1)
        mx = torch.where(uq, bp.v2f, torch.full_like(bp.v2f, -1e300)).max(1).values
        by_mean = hid_e & ~((mx - 700.0).abs() < 1e-9)
        assert float(mean[by_mean].abs().max()) < 1e-7
        bp2 = _lifted_sweeps(lf, n, sweeps)
        assert torch.equal(bp2.q_dev, bp.q_dev) and torch.equal(bp2.f2v, bp.f2v), name
        del bp, bp2
