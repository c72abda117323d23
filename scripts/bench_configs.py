"""Secondary measurements for the other BASELINE.json configurations (cfg 1-5) and the Gaussian sweep roofline.
Not the contract benchmark (that is bench.py); prints one JSON line per measurement."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'lifted-hybrid-variational-inference_amd')]
import numpy as np, torch
from lhvi import _abi, synth, lifting
from lhvi.flat import flatten

which = sys.argv[1:] or ['gauss', 'cfg2', 'cfg3', 'cfg5', 'vi_ground']


def ev_time(fn, reps=5):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    fn()
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


def out(**kw):
    print(json.dumps(kw), flush=True)


def gabp_plan(flat, records=True):
    """(plan struct, device arrays to keep alive) of the pull form; `records`: with the 16-byte slot records of round 4"""
    from lhvi.gabp import pull_plan
    host = pull_plan(flat)
    dev = {k: (_abi.to_dev(a) if a is not None else None) for k, a in host.items()}
    plan = _abi.GabpPlanStruct()
    plan.pslot, plan.info, plan.count = (_abi.ptr(dev[k]) for k in ('pslot', 'info', 'count'))
    plan.rec = _abi.ptr(dev['rec']) if records and os.environ.get('GABP_NO_RECORDS') is None else None
    plan.pot_words = _abi.ptr(dev['pot_words'])
    plan.seg, plan.n_seg = _abi.ptr(dev['seg']), int(host['seg'].shape[0])
    plan.n_hub_rows = int((np.diff(flat.var_ptr) > 512).sum())
    return plan, dev


if 'gauss' in which:
    # Gaussian sweep roofline: random pairwise Gaussian MRF, E = 10M edges (+ unary priors)
    flat = synth.random_gaussian_mrf(V=2_000_000, deg=4, seed=0)
    dg = _abi.DeviceGraph(flat)
    f2v, v2f, mv = dg.empty(flat.E, 2), dg.empty(flat.E, 2), dg.empty(flat.V, 2)
    l, st = _abi.lib(), _abi.stream_ptr()
    _abi.check(l.lhvi_gabp_init(dg.g, _abi.ptr(f2v), _abi.ptr(v2f), st))
    t_v = ev_time(lambda: _abi.check(l.lhvi_gabp_v2f(dg.g, _abi.ptr(f2v), _abi.ptr(v2f), st)))
    t_f = ev_time(lambda: _abi.check(l.lhvi_gabp_f2v(dg.g, dg.p, _abi.ptr(v2f), _abi.ptr(f2v), st)))
    bytes_sweep = 76.0 * flat.E
    out(config='gaussian sweep, random pairwise MRF, v2f + f2v kernel pair', edges=flat.E, v2f_ms=t_v, f2v_ms=t_f,
        sweeps_per_s=1e3 / (t_v + t_f), algorithmic_GBs=bytes_sweep / ((t_v + t_f) * 1e-3) / 1e9,
        hbm_frac=bytes_sweep / ((t_v + t_f) * 1e-3) / 8e12)
    # pull form: one launch per sweep, messages in slot order
    nnz = int(flat.var_edge.size)
    va, vb = dg.empty(nnz, 2), dg.empty(nnz, 2)
    for records in (False, True):
        plan, dev = gabp_plan(flat, records)
        _abi.check(l.lhvi_gabp_pull(dg.g, dg.p, plan, _abi.ptr(va), _abi.ptr(vb), 1, st))
        _abi.check(l.lhvi_gabp_pull(dg.g, dg.p, plan, _abi.ptr(vb), _abi.ptr(va), 0, st))
        t_p = ev_time(lambda: _abi.check(l.lhvi_gabp_pull(dg.g, dg.p, plan, _abi.ptr(va), _abi.ptr(vb), 0, st)))
        out(config='gaussian sweep, random pairwise MRF, pull form (lhvi_gabp_pull)' + (', slot records' if records else ', graph arrays (round 3)'),
            edges=flat.E, sweep_ms=t_p, sweeps_per_s=1e3 / t_p, algorithmic_GBs=bytes_sweep / (t_p * 1e-3) / 1e9, hbm_frac=bytes_sweep / (t_p * 1e-3) / 8e12)
    del dg, f2v, v2f, mv, va, vb

if 'cfg2' in which:
    # cfg 2: RGM template C=100, B=50 (E=20 200), GaBP ground and GaLBP lifted, 20 sweeps
    flat, sym, rv0, f0 = synth.rgm_flat(C=100, B=50, n_values=0, evidence_ratio=0.2, seed=0)
    dg = _abi.DeviceGraph(flat)
    f2v, v2f = dg.empty(flat.E, 2), dg.empty(flat.E, 2)
    l, st = _abi.lib(), _abi.stream_ptr()
    t = ev_time(lambda: _abi.check(l.lhvi_gabp_run(dg.g, dg.p, _abi.ptr(f2v), _abi.ptr(v2f), 20, st)))
    out(config='cfg2 RGM C=100 B=50 ground GaBP, kernel pair', edges=flat.E, ms_20_sweeps=t, sweeps_per_s=20e3 / t)
    plan, dev = gabp_plan(flat)
    nb = int(l.lhvi_gabp_pull_workspace_bytes(dg.g))
    ws = torch.empty(nb, dtype=torch.uint8, device=dg.device)
    t = ev_time(lambda: _abi.check(l.lhvi_gabp_run_pull(dg.g, dg.p, plan, _abi.ptr(f2v), _abi.ptr(v2f), 20, _abi.ptr(ws), nb, st)))
    out(config='cfg2 RGM C=100 B=50 ground GaBP, pull form', edges=flat.E, ms_20_sweeps=t, sweeps_per_s=20e3 / t)
    t0 = time.perf_counter()
    rvc, fc = lifting.refine_flat(flat, sym, rv0, f0)
    torch.cuda.synchronize()
    t_ref = time.perf_counter() - t0
    lflat = lifting.lift_flat(flat, rvc, fc)
    ldg = _abi.DeviceGraph(lflat)
    lf2v, lv2f, lmv, mv = ldg.empty(lflat.E, 2), ldg.empty(lflat.E, 2), ldg.empty(lflat.V, 2), dg.empty(flat.V, 2)
    lplan, ldev = gabp_plan(lflat)
    lnb = int(l.lhvi_gabp_pull_workspace_bytes(ldg.g))
    lws = torch.empty(lnb, dtype=torch.uint8, device=ldg.device)
    t_pair = ev_time(lambda: _abi.check(l.lhvi_gabp_run(ldg.g, ldg.p, _abi.ptr(lf2v), _abi.ptr(lv2f), 20, st)))
    t = ev_time(lambda: _abi.check(l.lhvi_gabp_run_pull(ldg.g, ldg.p, lplan, _abi.ptr(lf2v), _abi.ptr(lv2f), 20, _abi.ptr(lws), lnb, st)))
    _abi.check(l.lhvi_gabp_marginals(ldg.g, _abi.ptr(lf2v), _abi.ptr(lmv), st))
    _abi.check(l.lhvi_gabp_marginals(dg.g, _abi.ptr(f2v), _abi.ptr(mv), st))
    hid = flat.var_hidden
    err = float(np.abs(lmv.cpu().numpy()[rvc][hid, 0] - mv.cpu().numpy()[hid, 0]).max())
    out(config='cfg2 RGM lifted GaLBP', ground_edges=flat.E, rv_clusters=int(rvc.max()) + 1, f_clusters=int(fc.max()) + 1,
        lifted_edges=lflat.E, colour_passing_s=t_ref, ms_20_sweeps=t, ms_20_sweeps_kernel_pair=t_pair, max_abs_mu_diff_vs_ground=err)

if 'cfg2' in which:
    # cfg 2 through the solver API: the first run() builds the device state and records the run, the second replays it
    from lhvi.gabp import GaBP
    flat2, _, _, _ = synth.rgm_flat(C=100, B=50, n_values=0, evidence_ratio=0.2, seed=0)
    bp2 = GaBP(flat2)
    t0 = time.perf_counter(); bp2.run(20); torch.cuda.synchronize(); t_first = time.perf_counter() - t0
    st2 = bp2._state
    h2 = st2['graphs'][20]
    t_replay = ev_time(lambda: _abi.check(_abi.lib().lhvi_gabp_graph_launch(h2, _abi.stream_ptr())), reps=20)
    t0 = time.perf_counter(); bp2.run(20); torch.cuda.synchronize(); t_second = time.perf_counter() - t0
    out(config='cfg2 RGM C=100 B=50 ground GaBP.run(20): recorded run replayed', edges=int(flat2.E), first_run_s=t_first,
        second_run_wall_ms_incl_readback=1e3 * t_second, ms_20_sweeps_and_marginals_device=t_replay, sweeps_per_s=20e3 / t_replay)

if 'gauss_rel' in which:
    # the Gaussian sweep (pull form) on the reference's own structures at scale: the 10 M-edge RGM ground graph of cfg 5 and a
    # 22.8 M-edge Kalman-filter graph (KalmanFilter.grounded_flat, n = 30, T = 20 000)
    from lhvi.gabp import pull_plan
    from lhvi import kalman
    from lhvi.graph import Domain
    graphs = [('RGM 10M ground edges (cfg 5 before lifting)', synth.rgm_structured_flat()[0])]
    rng = np.random.default_rng(0)
    nk, Tk = 30, int(os.environ.get('KALMAN_T', 20000))
    A = rng.normal(size=(nk, nk)) * (rng.random((nk, nk)) < 0.4) * 0.2 + np.eye(nk) * 0.5
    if os.environ.get('KALMAN_CONST_A'):         # (diagnosis aid: two distinct transition coefficients -> the potential table fits the kernel's LDS copy)
        A = np.where(A != 0, 0.1, 0.0) + np.eye(nk) * 0.4
    data = rng.normal(size=(nk, Tk))
    data[rng.random(data.shape) < 0.3] = kalman.MISSING
    dom = Domain((-20, 20), continuous=True, integral_points=np.linspace(-20, 20, 8))
    graphs.append(('Kalman n=30 T=%d' % Tk, kalman.KalmanFilter(dom, A, 0.7, np.eye(nk), 0.4).grounded_flat(Tk, data)[0]))
    if os.environ.get('GAUSS_REL_ONLY'):         # one graph per process: a PMC pass then averages a kernel over ONE graph's launches
        graphs = [gr for gr in graphs if os.environ['GAUSS_REL_ONLY'].lower() in gr[0].lower()]
    for label, flat in graphs:
        dg = _abi.DeviceGraph(flat)
        l, st = _abi.lib(), _abi.stream_ptr()
        plan, dev = gabp_plan(flat)
        nnz = int(flat.var_edge.size)
        va, vb = dg.empty(nnz, 2), dg.empty(nnz, 2)
        _abi.check(l.lhvi_gabp_pull(dg.g, dg.p, plan, _abi.ptr(va), _abi.ptr(vb), 1, st))
        _abi.check(l.lhvi_gabp_pull(dg.g, dg.p, plan, _abi.ptr(vb), _abi.ptr(va), 0, st))
        t_p = ev_time(lambda: _abi.check(l.lhvi_gabp_pull(dg.g, dg.p, plan, _abi.ptr(va), _abi.ptr(vb), 0, st)))
        bytes_sweep = 76.0 * flat.E
        out(config='gaussian sweep, pull form, ' + label, edges=int(flat.E), hubs=int((np.diff(flat.var_ptr) > 512).sum()),
            max_degree=int(np.diff(flat.var_ptr).max()), sweep_ms=t_p, sweeps_per_s=1e3 / t_p,
            algorithmic_GBs=bytes_sweep / (t_p * 1e-3) / 1e9, hbm_frac=bytes_sweep / (t_p * 1e-3) / 8e12,
            finite=bool(torch.isfinite(vb[torch.from_numpy(flat.var_hidden[flat.edge_var[flat.var_edge]]).to(vb.device)]).all().item()))
        del dg, va, vb, dev

def cfg3_model(P_=300, T_=10, seed=0):
    """cfg 3: the paper-popularity HMLN, 300 papers x 10 topics, evidence per Generator `generate_data`, as Python objects"""
    from lhvi.graph import Domain
    from lhvi.relational import LV, Atom, ParamF, RelationalGraph
    from lhvi.mln import MLNPotential, eq_op
    rng = np.random.default_rng(seed)
    dom_b = Domain((0, 1))
    dom_r = Domain((-15, 15), continuous=True, integral_points=np.linspace(0, 10, 32))
    lvp, lvt = LV([f'p{i}' for i in range(P_)]), LV([f't{i}' for i in range(T_)])
    atoms = (Atom(dom_b, (lvt, lvt), 'SameSession'), Atom(dom_b, (lvp, lvt), 'PaperIn'), Atom(dom_r, (lvt,), 'TopicPopularity'),
             Atom(dom_r, (lvp,), 'PaperPopularity'))
    pfs = (ParamF(MLNPotential(lambda x: eq_op(x[0], 1), w=0.3), nb=['PaperPopularity(p)']),
           ParamF(MLNPotential(lambda x: x[0] * eq_op(x[1], x[2]), w=0.5), nb=['SameSession(t1,t2)', 'TopicPopularity(t1)', 'TopicPopularity(t2)'],
                  constrain=lambda s: s['t1'] != s['t2']),
           ParamF(MLNPotential(lambda x: x[0] * eq_op(x[1], x[2]), w=1), nb=['PaperIn(p,t)', 'PaperPopularity(p)', 'TopicPopularity(t)']))
    rel = RelationalGraph(atoms, pfs)
    g, table = rel.ground_graph()
    data = {}
    for i in rng.choice(P_, int(P_ * 0.7), replace=False):
        data[('PaperPopularity', f'p{i}')] = float(rng.uniform(0, 10))
    for i in rng.choice(T_, int(T_ * 0.7), replace=False):
        data[('TopicPopularity', f't{i}')] = float(rng.uniform(0, 10))
    for i in rng.choice(P_, int(P_ * 0.7), replace=False):
        for j in rng.choice(T_, int(rng.integers(T_)), replace=False):
            data[('PaperIn', f'p{i}', f't{j}')] = int(rng.integers(0, 2))
    rel.add_evidence(data)
    g.rvs, g.factors = sorted(g.rvs), sorted(g.factors)
    g.init_nb()
    return g, table


def cold_warm(make, call, warm=5):
    """wall seconds of the first call in this process (code-object loads, allocator growth, first flattening) and the median of
    `warm` later calls on fresh solver objects; `make()` builds the solver (untimed), `call(solver)` is what the reference's script times"""
    times = []
    for _ in range(warm + 1):
        obj = make()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        call(obj)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    if os.environ.get('REFSIZE_CPROFILE'):
        import cProfile, pstats
        o2 = make()
        pr = cProfile.Profile()
        pr.enable(); call(o2); torch.cuda.synchronize(); pr.disable()
        pstats.Stats(pr).sort_stats('cumulative').print_stats(40)
    return times[0], float(np.median(times[1:])), obj


if 'refsize' in which:
    # the calls the reference's own scripts make, at the reference's sizes, through the unchanged API: cold and warm wall time (the
    # device's share comes from the kernel trace of this same run: scripts/profile_refsize.sh), the C oracle beside them
    from lhvi.pbp import HybridLBP
    from lhvi.gabp import GaLBP
    only = os.environ.get('REFSIZE_ONLY', '')
    prof = os.environ.get('REFSIZE_CPROFILE')
    if not only or 'cfg3' in only:
        g, table = cfg3_model()
        E = sum(len(f.nb) for f in g.factors)

        def make3():
            np.random.seed(0)
            return HybridLBP(g, n=10, proposal_approximation='simple')
        cold, warmed, bp = cold_warm(make3, lambda b: b.run(10))
        out(config='refsize cfg3: paper-popularity HMLN 300 x 10, HybridLBP(g, n=10, simple).run(10) through the objects (host sampler = the reference\'s RNG stream)',
            edges=E, rv_clusters=bp.g.num_rv_clusters, cold_s=cold, warm_s=warmed, calls=6)

        def make3d():
            return HybridLBP(g, n=10, proposal_approximation='simple', sampler='device', seed=1)
        cold, warmed, bp = cold_warm(make3d, lambda b: b.run(10))
        # the C oracle on the same lifted graph, same number of sweeps (CPU baseline, labelled)
        from oracle import oracle
        lf = bp.flat
        o = oracle.PbpOracle(lf, 10, ep=False, epbp=False, var_threshold=5)
        draws = [bp.particles.cpu().numpy()] * 10            # (valid particles: what the timing needs)
        t0 = time.perf_counter(); o.run(10, draws); t_or = time.perf_counter() - t0
        out(config='refsize cfg3, device sampler', edges=E, lifted_edges=int(lf.E), cold_s=cold, warm_s=warmed, calls=6,
            cpu_oracle_10_sweeps_on_the_lifted_graph_s=t_or, cpu_cores_used=min(os.cpu_count() or 1, 16))
    if not only or 'cfg2' in only:
        # cfg 2: GaLBP(g).run(20) on the RGM template (C = 100, B = 50: 20 200 edges) through the objects
        from lhvi import generators
        rel = generators.rgm(100, 50)
        rel.ground_graph()
        rng2 = np.random.default_rng(0)
        keys = sorted(rel.rvs_dict)
        data2 = {keys[i]: float(np.round(rng2.uniform(-30, 30), 2)) for i in rng2.choice(len(keys), len(keys) // 5, replace=False)}
        g2, _ = rel.add_evidence(data2)
        g2.rvs, g2.factors = sorted(g2.rvs), sorted(g2.factors)
        g2.init_nb()
        cold, warmed, lbp = cold_warm(lambda: GaLBP(g2), lambda b: b.run(20))
        from oracle import oracle
        from lhvi.flat import flatten as _fl
        fo = _fl(lbp.g)
        t0 = time.perf_counter(); oracle.gabp_run(fo, 20); t_or = time.perf_counter() - t0
        out(config='refsize cfg2: RGM 100 x 50, 20 % observed, GaLBP(g).run(20) through the objects', edges=sum(len(f.nb) for f in g2.factors),
            lifted_edges=int(fo.E), cold_s=cold, warm_s=warmed, calls=6, cpu_oracle_20_sweeps_on_the_lifted_graph_s=t_or, cpu_cores_used=1)
    if not only or 'rgm' in only:
        # the reference's two RGM demo calls (Demo/RGM/demo.py:20-21; RGMKLDivergence.py:54-55) through the objects
        from lhvi import generators
        for label, n_, its_, nobs in (('Demo/RGM/demo.py: recession = 25 observed, HybridLBP(g, n=10, simple).run(10, c2f=0)', 10, 10, 0),
                                      ('Demo/RGM/RGMKLDivergence.py: 83 atoms observed, HybridLBP(g, n=20, simple).run(15, c2f=0)', 20, 15, 83)):
            rel = generators.rgm(100, 10)
            rel.ground_graph()
            rngr = np.random.default_rng(0)
            keys = sorted(rel.rvs_dict)
            datar = {('recession', 'all'): 25.0} if nobs == 0 else \
                {keys[i]: float(np.round(rngr.uniform(-30, 30), 2)) for i in rngr.choice(len(keys), nobs, replace=False)}
            gr, _ = rel.add_evidence(datar)
            gr.rvs, gr.factors = sorted(gr.rvs), sorted(gr.factors)
            gr.init_nb()
            cold, warmed, bpr = cold_warm(lambda: HybridLBP(gr, n=n_, proposal_approximation='simple', sampler='device', seed=1),
                                          lambda b: b.run(its_, c2f=0))
            out(config='refsize ' + label, edges=sum(len(f.nb) for f in gr.factors), rv_clusters_final=int(bpr.flat.V), cold_s=cold,
                warm_s=warmed, calls=6)

if 'cfg3' in which:
    # cfg 3: paper-popularity HMLN (300 papers x 10 topics) through the object API, HybridLBP n=10, 10 sweeps
    from lhvi.graph import Domain
    from lhvi.relational import LV, Atom, ParamF, RelationalGraph
    from lhvi.mln import MLNPotential, eq_op
    from lhvi.pbp import HybridLBP
    rng = np.random.default_rng(0)
    P_, T_ = 300, 10
    dom_b = Domain((0, 1))
    dom_r = Domain((-15, 15), continuous=True, integral_points=np.linspace(0, 10, 32))
    lvp, lvt = LV([f'p{i}' for i in range(P_)]), LV([f't{i}' for i in range(T_)])
    atoms = (Atom(dom_b, (lvt, lvt), 'SameSession'), Atom(dom_b, (lvp, lvt), 'PaperIn'), Atom(dom_r, (lvt,), 'TopicPopularity'),
             Atom(dom_r, (lvp,), 'PaperPopularity'))
    pfs = (ParamF(MLNPotential(lambda x: eq_op(x[0], 1), w=0.3), nb=['PaperPopularity(p)']),
           ParamF(MLNPotential(lambda x: x[0] * eq_op(x[1], x[2]), w=0.5), nb=['SameSession(t1,t2)', 'TopicPopularity(t1)', 'TopicPopularity(t2)'],
                  constrain=lambda s: s['t1'] != s['t2']),
           ParamF(MLNPotential(lambda x: x[0] * eq_op(x[1], x[2]), w=1), nb=['PaperIn(p,t)', 'PaperPopularity(p)', 'TopicPopularity(t)']))
    rel = RelationalGraph(atoms, pfs)
    g, table = rel.ground_graph()
    data = {}
    for i in rng.choice(P_, int(P_ * 0.7), replace=False):
        data[('PaperPopularity', f'p{i}')] = float(rng.uniform(0, 10))
    for i in rng.choice(T_, int(T_ * 0.7), replace=False):
        data[('TopicPopularity', f't{i}')] = float(rng.uniform(0, 10))
    for i in rng.choice(P_, int(P_ * 0.7), replace=False):
        for j in rng.choice(T_, int(rng.integers(T_)), replace=False):
            data[('PaperIn', f'p{i}', f't{j}')] = int(rng.integers(0, 2))
    rel.add_evidence(data)
    g.rvs, g.factors = sorted(g.rvs), sorted(g.factors)
    g.init_nb()
    E = sum(len(f.nb) for f in g.factors)
    np.random.seed(0)
    bp = HybridLBP(g, n=10, proposal_approximation='simple')
    t0 = time.perf_counter()
    bp.run(10)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out(config='cfg3 paper-popularity HMLN HybridLBP n=10 T=32 c2f=-1', rvs=len(g.rvs), factors=len(g.factors), edges=E,
        rv_clusters=bp.g.num_rv_clusters, f_clusters=bp.g.num_factor_clusters, seconds_10_sweeps_incl_lifting=dt,
        example_map=float(bp.map(table[('TopicPopularity', 't0')])))

if 'cfg3s' in which:
    # cfg 3 scaled (SURVEY 8(d) cfg 3 template): CFG3_COPIES independent groundings of the reference-size model (300 papers x 10
    # topics, own evidence each) -- or, with CFG3_P / CFG3_T, one grounding of that size -- straight into arrays; EPBP semantics on
    # the ground graph, n = 64 particles, 32 integral points, 'simple' proposals, device sampler.  Device time per sweep (HIP events,
    # after set-up) with the conditional-quadratic routing and with every MLN edge on the generic kernel.
    from lhvi.pbp import EPBP
    P_, T_, K_ = int(os.environ.get('CFG3_P', 300)), int(os.environ.get('CFG3_T', 10)), int(os.environ.get('CFG3_COPIES', 286))
    t0 = time.perf_counter()
    flat = synth.paper_popularity_copies(K_, P_, T_, seed=0)
    t_ground = time.perf_counter() - t0
    res = {}
    for routed in ((True,) if os.environ.get('CFG3_ROUTED_ONLY') else (True, False)):
        bp = EPBP(None, n=64, proposal_approximation='simple', sampler='device', seed=1)
        bp.cq_routing = routed
        t0 = time.perf_counter()
        bp._setup(None, flat=flat)
        torch.cuda.synchronize()
        t_setup = time.perf_counter() - t0
        _abi.check(_abi.lib().lhvi_pbp_init(bp.dg.g, bp._struct(), _abi.ptr(bp.eta), _abi.ptr(bp.q_dev), _abi.ptr(bp.f2v),
                                            _abi.ptr(bp.v2f), _abi.stream_ptr()))
        bp._generate_sample()
        for _ in range(2):
            bp.sweep()
        torch.cuda.synchronize()
        q2 = bp.q_dev.clone()                  # proposals after the two warm-up sweeps: compared between the routings
        reps = 5 if routed else 2
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        fev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for (a, b), (fa, fb) in zip(ev, fev):
            a.record()
            _abi.check(_abi.lib().lhvi_pbp_v2f(bp.dg.g, bp._struct(), _abi.ptr(bp.f2v), _abi.ptr(bp.v2f), _abi.stream_ptr()))
            _abi.check(_abi.lib().lhvi_pbp_proposal(bp.dg.g, bp._struct(), _abi.ptr(bp.f2v), _abi.ptr(bp.eta), _abi.ptr(bp.q_dev), _abi.stream_ptr()))
            bp._generate_sample()
            fa.record()
            bp._launch_f2v(bp._struct())
            fb.record()
            b.record()
        torch.cuda.synchronize()
        sweep_ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
        f2v_ms = float(np.median([a.elapsed_time(b) for a, b in fev]))
        res[routed] = dict(sweep_ms=sweep_ms, f2v_ms=f2v_ms, setup_s=t_setup, heavy=bp.n_heavy, heavy_terms=bp.heavy_terms, pair=bp.n_pair,
                           light=bp.n_light, cq=bp.n_cq, cq_terms=bp.cq_terms, generic=int(bp.generic_edges.numel()),
                           finite=bool(torch.isfinite(bp.f2v).all().item()))
        if routed:
            ref, qdiff = q2, None
        else:
            qdiff = float((q2 - ref).abs().max().item())
        del bp
    r, u = res[True], res.get(False, dict(sweep_ms=None, f2v_ms=None, generic=None, finite=None))
    terms = r['heavy_terms'] + r['cq_terms']
    out(config='cfg3 scaled: %d x paper-popularity HMLN %d papers x %d topics, ground EPBP n=64 T=32 simple' % (K_, P_, T_), rvs=int(flat.V),
        factors=int(flat.F), edges=int(flat.E), hidden=int(flat.var_hidden.sum()), max_degree=int(np.diff(flat.var_ptr).max()),
        grounding_host_s=t_ground, setup_s=r['setup_s'], sweep_ms=r['sweep_ms'], f2v_ms=r['f2v_ms'], sweeps_per_s=1e3 / r['sweep_ms'],
        edge_messages_per_s=2e3 * flat.E / r['sweep_ms'], heavy_edges=r['heavy'], pair_records=r['pair'], cq_edges=r['cq'],
        generic_edges=r['generic'], heavy_terms=r['heavy_terms'], cq_terms=r['cq_terms'],
        f2v_fp64_TFLOPs_at_16_flop_per_term=16.0 * terms / (r['f2v_ms'] * 1e-3) / 1e12, finite=r['finite'],
        generic_routing_sweep_ms=u['sweep_ms'], generic_routing_f2v_ms=u['f2v_ms'], generic_routing_generic_edges=u['generic'], generic_routing_finite=u['finite'],
        generic_routing_f2v_TFLOPs_same_terms=(16.0 * terms / (u['f2v_ms'] * 1e-3) / 1e12 if u['f2v_ms'] else None),
        max_abs_q_diff_routed_vs_generic_after_2_sweeps=qdiff)

if 'cfg5' in which:
    # cfg 5: RGM template at 10M ground edges, structured evidence; colour refinement on the device, then lifted VI
    flat, sym, rv0, f0 = synth.rgm_structured_flat()          # C = 2000, B = 1250: 10.0 M ground edges, ~10 k rv clusters
    dg5 = _abi.DeviceGraph(flat)                     # graph resident in HBM before the timed region
    lifting.refine_flat(flat, sym, rv0, f0, dg=dg5)  # warm-up (rocPRIM temporary storage, code objects)
    torch.cuda.synchronize()
    st5 = {}
    t0 = time.perf_counter()
    rvc_d, fc_d = lifting.refine_flat(flat, sym, rv0, f0, dg=dg5, stats=st5, device_out=True)
    torch.cuda.synchronize()
    t_ref = time.perf_counter() - t0
    t0 = time.perf_counter()
    rvc_s, fc_s = lifting.refine_flat(flat, sym, rv0, f0, dg=dg5, method=_abi.COLOR_SORT, device_out=True)
    torch.cuda.synchronize()
    t_ref_sort = time.perf_counter() - t0
    assert bool((rvc_s == rvc_d).all()) and bool((fc_s == fc_d).all())
    lifting.lift_flat(flat, rvc_d, fc_d, dg=dg5)     # warm-up of the torch kernels it uses
    t0 = time.perf_counter()
    lflat = lifting.lift_flat(flat, rvc_d, fc_d, dg=dg5)
    t_lift_dev = time.perf_counter() - t0
    rvc, fc = rvc_d.cpu().numpy(), fc_d.cpu().numpy()
    del dg5
    t0 = time.perf_counter()
    lflat_h = lifting.lift_flat(flat, rvc, fc)
    t_lift = time.perf_counter() - t0
    assert all(np.array_equal(getattr(lflat, k), getattr(lflat_h, k), equal_nan=True) for k in
               ('fac_ptr', 'edge_var', 'var_ptr', 'var_edge', 'edge_count', 'var_value', 'var_mult', 'fac_mult', 'fac_pot'))
    from lhvi.vi import VarInference
    vi = VarInference(None, 2, 3)
    vi._setup_flat(lflat)
    np.random.seed(0)
    vi.init_param()
    fe0 = vi.free_energy()
    t0 = time.perf_counter()
    vi.is_log, vi.log_fe = False, True
    vi.alpha, vi.b1, vi.b2, vi.eps, vi.t = 0.1, 0.9, 0.999, 1e-8, 0
    vi.ADAM_update(20)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out(config='cfg5 RGM 10M ground edges -> colour refinement -> LVI K=2 T=3', ground_edges=flat.E,
        rv_clusters=int(rvc.max()) + 1, f_clusters=int(fc.max()) + 1, lifted_edges=lflat.E, colour_refinement_s=t_ref, colour_rounds=st5.get('rounds'),
        colour_ms_per_round=1e3 * t_ref / max(st5.get('rounds', 1), 1), colour_GBs_at_120B_per_edge_round=120.0 * flat.E * st5.get('rounds', 0) / t_ref / 1e9,
        colour_refinement_by_radix_sort_s=t_ref_sort, lift_flat_device_assisted_s=t_lift_dev, lift_flat_host_s=t_lift, adam_iterations_per_s=20 / dt, fe_start=fe0, fe_after_20=vi.free_energy())

if 'full_size' in which:
    # size-independent parity properties at BASELINE's full size (cfg 5: the RGM template at 10.0 M ground edges, structured evidence),
    # where no oracle finishes: (1) lifting is exact for Gaussian BP -- the counted sweep on the lifted graph (39 k edges) gives every
    # ground variable the marginal the ground sweep over 10 M edges gives it; (2) the lifted free energy and gradient of the
    # variational step equal the ground ones at parameters tied per cluster (multiplicities len(rv.rvs), len(f.factors), rv.count[f]:
    # LVI:59-199 against VI:57-195); (3) hash-table and radix-sort refinement give the same colour arrays; (4) the device
    # reductions of lift_flat equal the host ones
    from lhvi.gabp import GaBP
    from lhvi.vi import VarInference
    flat, sym, rv0, f0 = synth.rgm_structured_flat()
    dgf = _abi.DeviceGraph(flat)
    rvc_d, fc_d = lifting.refine_flat(flat, sym, rv0, f0, dg=dgf, device_out=True)
    rvc_s, fc_s = lifting.refine_flat(flat, sym, rv0, f0, dg=dgf, method=_abi.COLOR_SORT, device_out=True)
    same_colours = bool((rvc_s == rvc_d).all()) and bool((fc_s == fc_d).all())
    lflat = lifting.lift_flat(flat, rvc_d, fc_d, dg=dgf)
    rvc, fc = rvc_d.cpu().numpy(), fc_d.cpu().numpy()
    lflat_h = lifting.lift_flat(flat, rvc, fc)
    same_lift = all(np.array_equal(getattr(lflat, k), getattr(lflat_h, k), equal_nan=True) for k in
                    ('fac_ptr', 'edge_var', 'var_ptr', 'var_edge', 'edge_count', 'var_value', 'var_mult', 'fac_mult', 'fac_pot'))
    del dgf
    its = 10
    ground, lifted = GaBP(flat), GaBP(lflat)
    ground.run(its)
    lifted.run(its)
    hid = flat.var_hidden
    mg, ml = ground._mu_var[hid], lifted._mu_var[rvc[hid]]
    d_mu = float(np.abs(mg[:, 0] - ml[:, 0]).max())
    d_var = float(np.abs(mg[:, 1] / ml[:, 1] - 1).max())
    del ground, lifted
    K_, T_ = 2, 3
    lv = VarInference(None, K_, T_)
    lv._setup_flat(lflat)
    np.random.seed(0)
    lv.init_param()
    lv._grad()
    gv = VarInference(None, K_, T_)
    gv._setup_flat(flat)
    d = lv._dev
    gv._upload_params(d['w_tau'].cpu().numpy(), d['eta_c'].cpu().numpy()[rvc], d['tau_d'].cpu().numpy()[rvc])
    gv._grad()
    fe_l, fe_g = float(lv._dev['fe'].cpu().numpy()[0]), float(gv._dev['fe'].cpu().numpy()[0])
    gw_l, gw_g = lv._dev['g_w'].cpu().numpy(), gv._dev['g_w'].cpu().numpy()
    # the gradient of a cluster's parameters is the ground gradient of any one of its members (LVI:94-134 walks the factor clusters
    # of ONE variable with rv.count[f]; the cluster size enters the free energy, not this gradient): every member against its cluster
    gc_g, gc_l = gv._dev['g_c'].cpu().numpy(), lv._dev['g_c'].cpu().numpy()
    contg = flat.var_hidden & flat.var_cont
    d_gc = float(np.abs(gc_g[contg] - gc_l[rvc][contg]).max() / max(np.abs(gc_l).max(), 1e-300))
    out(config='full-size properties, cfg 5 RGM (10.0 M ground edges -> %d lifted)' % lflat.E, ground_edges=int(flat.E), lifted_edges=int(lflat.E),
        hash_and_sort_refinement_same_colours=same_colours, device_and_host_lift_same_graph=same_lift,
        gabp_sweeps=its, max_abs_mu_lifted_vs_ground=d_mu, max_rel_var_lifted_vs_ground=d_var,
        free_energy_lifted=fe_l, free_energy_ground=fe_g, rel_diff_free_energy=abs(fe_l - fe_g) / abs(fe_g),
        max_rel_diff_g_w=float(np.abs(gw_l - gw_g).max() / np.abs(gw_g).max()), max_rel_diff_member_gradient_vs_cluster_gradient=d_gc)

if 'c2fvi' in which:
    # C2FVarInference on the 10 M-edge RGM of cfg 5, on arrays: coarse start, evidence split by k-means under a shrinking threshold,
    # re-lift every 10 ADAM updates (C2FVarInference.py:301-352).  Reported: what a round spends re-lifting (evidence split, colour
    # refinement to the fixed point, lift_flat, Gaussian-observation variances) and what it spends in its 10 ADAM updates.
    from lhvi import c2fvi
    flat, sym, rv0, f0 = synth.rgm_structured_flat()
    dgc = _abi.DeviceGraph(flat)
    owner = c2fvi.VarInference.__new__(c2fvi.VarInference)
    owner._init_common(2, 3)
    np.random.seed(0)
    opts = dict(k_mean_k=2, k_mean_its=10, update_obs_its=10, output_its=0, min_obs_var=0, gaussian_obs=True)
    seen = []
    t0 = time.perf_counter()
    res = c2fvi.run_c2fvi_flat(flat, c2fvi._DeviceEngine(owner), 2, 30, 0.2, opts, dg=dgc,
                               observer=lambda r, st: seen.append((int(st['rvc'].max()) + 1, int((st['obs_var'] > 0).sum()))))
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    out(config='C2FVarInference K=2 T=3, 30 updates in 3 rounds, RGM 10M ground edges (arrays)', ground_edges=int(flat.E),
        rv_clusters_per_round=[a for a, _ in seen], gaussian_observation_clusters_per_round=[b for _, b in seen],
        relift_ms_per_round=[1e3 * x for x in res['relift_s']], total_s=total, fe_first=res['fe_log'][0], fe_last=res['fe_log'][-1])
    del dgc

if 'lifted_pbp' in which:
    # the counted particle sweep (HybridLBP semantics) on array-lifted graphs: the cfg-5 graph (10 M ground edges -> 39 260 lifted)
    # and a 10 M-edge graph that lifts to ~1 M edges; n = 10 as in Demo/RGM/demo.py:19-20, the RGM's 100 integral points
    from lhvi.pbp import HybridLBP
    for label, args in (('cfg5 lifted', (2000, 1250, 400, 250)), ('10M ground -> ~1M lifted edges', (2000, 1250, 1000, 250, True))):
        flat, sym, rv0, f0 = synth.rgm_structured_flat(*args)
        dgl = _abi.DeviceGraph(flat)
        rd, fd = lifting.refine_flat(flat, sym, rv0, f0, dg=dgl, device_out=True)
        lf = lifting.lift_flat(flat, rd, fd, dg=dgl)
        del dgl
        for n_ in (10, 64):
            bp = HybridLBP.on_flat(lf, n=n_, proposal_approximation='simple', sampler='device', seed=1)
            bp._setup(None, flat=lf)
            _abi.check(_abi.lib().lhvi_pbp_init(bp.dg.g, bp._struct(), _abi.ptr(bp.eta), _abi.ptr(bp.q_dev), _abi.ptr(bp.f2v),
                                                _abi.ptr(bp.v2f), _abi.stream_ptr()))
            bp._generate_sample()
            t = ev_time(lambda: bp.sweep(last=False))
            out(config='lifted particle sweep (HybridLBP semantics, simple proposals), %s, n=%d T=%d' % (label, n_, bp.T), ground_edges=int(flat.E),
                lifted_edges=int(lf.E), rv_clusters=int(lf.V), max_count=float(lf.edge_count.max()), max_lifted_degree=int(np.diff(lf.var_ptr).max()),
                sweep_ms=t, sweeps_per_s=1e3 / t, ground_edge_messages_per_s=2e3 * flat.E / t, heavy_edges=bp.n_heavy,
                fast_edges=int(bp.fast_edges.numel()), generic_edges=int(bp.generic_edges.numel()), finite=bool(torch.isfinite(bp.f2v).all().item()))
            del bp

if 'grid100' in which:
    # the headline generator with the 100 integral points of the reference's RGM domain (Demo/Data/RGM/Generator.py:16) instead of
    # BASELINE's 32: n + T = 164 output points, more than the heavy kernel's two rounds -- its list takes these edges since round 3
    # because the recurrence tabulates up to 128 grid points; before, the general kernel served them with one exponential per term
    from lhvi.pbp import EPBP
    flat = synth.hybrid_mrf_flat(V=500_000, deg=4, seed=0, T=100)
    for label, min_edges in (('integral points by recurrence (heavy kernel)', 0), ('general kernel, direct form (round 2)', 1 << 30)):
        bp = EPBP(None, n=64, proposal_approximation='simple', sampler='device', seed=1)
        bp.long_grid_min_edges = min_edges
        bp._setup(None, flat=flat)
        _abi.check(_abi.lib().lhvi_pbp_init(bp.dg.g, bp._struct(), _abi.ptr(bp.eta), _abi.ptr(bp.q_dev), _abi.ptr(bp.f2v),
                                            _abi.ptr(bp.v2f), _abi.stream_ptr()))
        bp._generate_sample()
        for _ in range(2):
            bp.sweep(last=False)
        t = ev_time(lambda: bp.sweep(last=False))
        out(config='cfg4 generator, 2 M edges, n=64, T=100 (the grid of the reference\'s RGM domain): ' + label, edges=int(flat.E), sweep_ms=t,
            sweeps_per_s=1e3 / t, heavy_edges=bp.n_heavy, fast_edges=int(bp.fast_edges.numel()), q_checksum=float(bp.q_dev.nan_to_num().abs().sum().item()),
            finite=bool(torch.isfinite(bp.f2v).all().item()))
        del bp

if 'vi_ground' in which:
    # the variational step on a GROUND graph: RGM template C=1000, B=500 (1.0 M pairwise Gaussian factors), K=2, T=3
    from lhvi.vi import VarInference
    flat, sym, rv0, f0 = synth.rgm_flat(C=1000, B=500, n_values=0, evidence_ratio=0.1, seed=0)
    vi = VarInference(None, 2, 3)
    vi._setup_flat(flat)
    np.random.seed(0)
    vi.init_param()
    t = ev_time(vi._grad)
    out(config='ground VI gradient + free energy, RGM C=1000 B=500, K=2 T=3', factors=int(flat.F), edges=int(flat.E), grad_ms=t,
        factors_per_s=flat.F / (t * 1e-3), quadrature_nodes_per_s=flat.F * 2 * 9 / (t * 1e-3))

if 'c2f_pbp' in which:
    # the particle coarse-to-fine run the reference's demos make (Demo/RGM/demo.py:20-21: HybridLBP(g, n=10, 'simple').run(10, c2f=0)
    # with recession = 25 observed; Demo/RGM/RGMKLDivergence.py:54-55: n=20, run(15, c2f=0) on Demo/Data/RGM/0..4: 83 of 1 111 atoms
    # observed ~U(-30, 30)), on arrays (lhvi.c2f.run_c2f_flat).  Per sweep: seconds in the two re-liftings (refinement half rounds on
    # the device + building both lifted graphs), in building the two solver states (host), and in the message kernels; the C oracle
    # driving the same schedule on the host beside it (CPU baseline, labelled; reference-size graphs only).
    from lhvi.pbp import HybridLBP
    from lhvi import c2f as _c2f

    def demo_flat():
        flat, _, _, _ = synth.rgm_flat(C=100, B=10, n_values=0, evidence_ratio=0.0, seed=0)
        val = np.full(flat.V, np.nan)
        val[0] = 25.0
        flat.var_value = val
        return flat

    def kl_flat():
        flat, _, _, _ = synth.rgm_flat(C=100, B=10, n_values=0, evidence_ratio=0.0, seed=0)
        rng = np.random.default_rng(0)
        val = np.full(flat.V, np.nan)
        ev = rng.choice(np.arange(1, flat.V), 83, replace=False)
        val[ev] = rng.uniform(-30, 30, 83)
        flat.var_value = val
        return flat

    def cpu_baseline_c2f(flat, n, its):
        """CPU baseline (labelled): the same schedule driven by the C oracle and the exact CPU refinement on this box's host cores"""
        from oracle.engines import OracleEngine, OracleTensorRefiner
        rvc0, fc0, sym = lifting.initial_colors_flat(flat, False)
        rng = np.random.default_rng(0)

        def draw(k, lf, q):
            lo, hi = lf.dom_lo[lf.var_dom][:, None], lf.dom_hi[lf.var_dom][:, None]
            return np.clip(rng.standard_normal((lf.V, n)) * np.sqrt(np.nan_to_num(q[:, 1:2], nan=1.0)) + np.nan_to_num(q[:, 0:1]), lo, hi)
        t0 = time.perf_counter()
        _c2f.run_c2f_flat(flat, lifting.TensorGraph(flat), OracleEngine(n, False), OracleTensorRefiner(flat, sym), its, 0, 2, 10, draw, rvc0, fc0)
        return time.perf_counter() - t0

    cases = [('RGM 100 x 10, recession = 25 observed (Demo/RGM/demo.py), n=10, run(10, c2f=0)', demo_flat, 10, 10, True),
             ('RGM 100 x 10, 83 atoms observed ~U(-30,30) (Demo/Data/RGM/0 shape), n=20, run(15, c2f=0)', kl_flat, 20, 15, True),
             ('RGM 2000 x 1250 (10 M ground edges, cfg 5 evidence), n=10, run(10, c2f=0)', lambda: synth.rgm_structured_flat()[0], 10, 10, False)]
    if os.environ.get('C2F_SMALL_ONLY'):
        cases = cases[:2]
    if os.environ.get('C2F_CASE'):
        cases = [cases[int(os.environ['C2F_CASE'])]]
    for label, make, n_, its, with_cpu in cases:
        flat = make()
        wall, timing = None, None
        for rep in range(2):                                    # the first pass pays code-object loads and allocator growth
            bp = HybridLBP.on_flat(flat, n=n_, proposal_approximation='simple', sampler='device', seed=1)
            timing = {}
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            bp.run_flat(its, c2f=0, timing=timing)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
        untimed = None
        bp2 = HybridLBP.on_flat(flat, n=n_, proposal_approximation='simple', sampler='device', seed=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bp2.run_flat(its, c2f=0)
        torch.cuda.synchronize()
        untimed = time.perf_counter() - t0
        ps = timing['per_sweep']
        med = lambda name: float(np.median([s[name] for s in ps[1:]])) * 1e3
        out(config='particle coarse-to-fine on arrays: ' + label, ground_edges=int(flat.E), rv_clusters_final=int(bp.flat.V),
            lifted_edges_final=int(bp.flat.E), wall_s_with_phase_syncs=wall, wall_s=untimed,
            per_sweep_ms_median=dict(relift=med('lift'), build_states=med('setup'), message_kernels=med('sweep')),
            totals_s=timing['total'], finite=bool(torch.isfinite(bp.v2f).all().item()),
            cpu_oracle_same_schedule_s=(cpu_baseline_c2f(flat, n_, its) if with_cpu else None), cpu_cores_used=1)
        del bp, bp2

if 'vi_models' in which:
    # the variational step on the models the reference published timings for (VI / LVI / C2FVI seconds per ADAM
    # update, K = 2, T = 3, 100 updates; Demo/HMLN/HMLNTimeLog.py:48-58: paper-popularity 2.93 / 1.98 / 1.68 s/it, robot-mapping
    # 6.37 / 6.10 / 1.90 s/it on an unrecorded CPU): seconds per update end to end (set-up, lifting, loop, read-back) and device
    # only (HIP events around the loop), with the C oracle (CPU baseline, labelled, bounded sample) on the same graph beside them.
    import gzip
    from lhvi import c2fvi, generators
    from lhvi.vi import VarInference

    def robot_flat():
        rec = json.load(gzip.open(os.path.join(ROOT, 'tests', 'golden', 'grounding.json.gz'), 'rt'))['robot_mapping']
        data = {tuple(k): v for k, v in rec['evidence']}
        return generators.robot_mapping().ground_flat(data)[0]

    def cpu_baseline_vi(flat, updates, obs_var=None, K=2):
        """CPU baseline (labelled): the C oracle's ADAM updates on the same graph, one host core; seconds per update"""
        from oracle import oracle
        o = oracle.ViOracle(flat, K, 3, obs_var=obs_var)
        rng = np.random.default_rng(0)
        eta_c = np.ones((flat.V, K, 2)); eta_c[:, :, 0] = rng.random((flat.V, K)) * 3 - 1.5
        o.set_params(np.zeros(K), eta_c, rng.random((flat.V, K, o.Dmax)) * 10)
        t0 = time.perf_counter()
        o.run(updates, lr=0.2)
        return (time.perf_counter() - t0) / updates

    def timed_loop(vi, updates):
        np.random.seed(0)
        vi.init_param()
        vi.is_log, vi.log_fe = True, True
        vi.alpha, vi.b1, vi.b2, vi.eps, vi.t = 0.2, 0.9, 0.999, 1e-8, 0
        vi.time_log, vi.total_time = [], 0
        vi.ADAM_update(2)                              # warm-up (code objects)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        vi.time_log = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        a.record(); vi.ADAM_update(updates); b.record()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, a.elapsed_time(b) * 1e-3, vi.time_log[-1][1]

    # (the RGM instances of Demo/RGM/RGMTimeLog.py:14-38: 100 categories x 5 banks = 606 atoms, 121 / 30 of them observed, K = 1,
    # T = 3, 200 updates; Demo/Data/RGM/time_log_20_result, time_log_5_result)
    models = [('paper-popularity HMLN 300 papers x 10 topics (Demo/Data/HMLN/0 evidence pattern)', lambda: synth.paper_popularity_flat(300, 10, seed=0, points=20)[0],
               dict(VI=2.93, LVI=1.98, C2FVI=1.68), 2, 100),
              ('robot-mapping HMLN (Demo/Data/HMLN/robot-map evidence + closed world)', robot_flat, dict(VI=6.37, LVI=6.10, C2FVI=1.90), 2, 100),
              ('RGM 100 x 5, 20 % of the atoms observed (Demo/RGM/RGMTimeLog.py)', lambda: synth.rgm_flat(C=100, B=5, n_values=0, evidence_ratio=0.2, seed=0)[0],
               dict(VI=1.65, LVI=1.08, C2FVI=0.99), 1, 200),
              ('RGM 100 x 5, 5 % of the atoms observed (Demo/RGM/RGMTimeLog.py)', lambda: synth.rgm_flat(C=100, B=5, n_values=0, evidence_ratio=0.05, seed=0)[0],
               dict(VI=2.17, LVI=0.50, C2FVI=0.46), 1, 200)]
    if os.environ.get('VI_MODEL'):
        models = [models[int(os.environ['VI_MODEL'])]]
    for label, make, published, K_, UPD in models:
        flat = make()
        base = dict(model=label, rvs=int(flat.V), factors=int(flat.F), edges=int(flat.E), hidden=int(flat.var_hidden.sum()),
                    max_arity=int(np.diff(flat.fac_ptr).max()), K=K_, T=3, updates=UPD)
        # ---- VI on the ground graph (set-up timed twice: the first one in a process also pays the library's code-object loads)
        t_first = None
        for rep in range(2):
            flat_ = make()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            vi = VarInference(None, K_, 3)
            vi._setup_flat(flat_)
            torch.cuda.synchronize()
            t_setup = time.perf_counter() - t0
            t_first = t_setup if t_first is None else t_first
        wall, dev, fe = timed_loop(vi, UPD)
        out(config='VI (ground) ' + label, **base, setup_s=t_setup, setup_s_first_call_in_process=t_first, s_per_update_end_to_end=(t_setup + wall) / UPD, s_per_update_loop_wall=wall / UPD,
            s_per_update_device=dev / UPD, fe_last=fe, reference_published_s_per_update_unknown_cpu=published['VI'],
            cpu_oracle_s_per_update_1_core=cpu_baseline_vi(flat, 3, K=K_))
        del vi
        # ---- LVI: colour passing to the stable partition, then the same loop on the lifted graph
        t0 = time.perf_counter()
        rv0, f0, sym = lifting.initial_colors_flat(flat, True)
        dgm = _abi.DeviceGraph(flat)
        rvc, fc = lifting.refine_flat(flat, sym, rv0, f0, dg=dgm, device_out=True)
        lflat = lifting.lift_flat(flat, rvc, fc, dg=dgm)
        lvi = VarInference(None, K_, 3)
        lvi._setup_flat(lflat)
        torch.cuda.synchronize()
        t_setup = time.perf_counter() - t0
        wall, dev, fe = timed_loop(lvi, UPD)
        out(config='LVI (lifted) ' + label, **base, rv_clusters=int(lflat.V), factor_clusters=int(lflat.F), lifted_edges=int(lflat.E),
            lifting_and_setup_s=t_setup, s_per_update_end_to_end=(t_setup + wall) / UPD, s_per_update_loop_wall=wall / UPD, s_per_update_device=dev / UPD,
            fe_last=fe, reference_published_s_per_update_unknown_cpu=published['LVI'], cpu_oracle_s_per_update_1_core=cpu_baseline_vi(lflat, 3, K=K_))
        del lvi, dgm
        # ---- C2FVI: coarse start, re-lift every 10 updates (C2FVarInference.py:301-352), on arrays
        owner = c2fvi.VarInference.__new__(c2fvi.VarInference)
        owner._init_common(K_, 3)
        opts = dict(k_mean_k=2, k_mean_its=10, update_obs_its=10, output_its=0, min_obs_var=0, gaussian_obs=True)
        for rep in range(2):
            np.random.seed(0)
            seen = []
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = c2fvi.run_c2fvi_flat(flat, c2fvi._DeviceEngine(owner), K_, UPD, 0.2, opts,
                                       observer=lambda r, st: seen.append(int(st['rvc'].max()) + 1))
            torch.cuda.synchronize()
            total = time.perf_counter() - t0
        out(config='C2FVI (coarse to fine, arrays) ' + label, **base, rv_clusters_per_round=seen, total_s=total, s_per_update_end_to_end=total / UPD,
            relift_ms_per_round=[round(1e3 * x, 2) for x in res['relift_s']], fe_last=res['fe_log'][-1],
            reference_published_s_per_update_unknown_cpu=published['C2FVI'],
            cpu_oracle_s_per_update_1_core_final_lifted_graph=cpu_baseline_vi(res['flat'], 3, obs_var=res['obs_var'], K=K_))

if 'vi_scaled' in which:
    # ground VI on the scaled cfg 3 (286 groundings of the 300 x 10 paper-popularity HMLN: ~970 k factors, discrete axes, ternary
    # formulas -- the general vi_factor_kernel<3>) and on the Gaussian RGM (vi_factor_cc_kernel): gradient + free energy per launch,
    # roofline at SURVEY 8(d)'s K * T^a * (K * a * 25 + c_phi) flop per factor
    from lhvi.vi import VarInference
    K_, T_ = 2, 3
    for label, flat in (('scaled cfg 3: %d x paper-popularity 300 x 10' % int(os.environ.get('CFG3_COPIES', 286)),
                         synth.paper_popularity_copies(int(os.environ.get('CFG3_COPIES', 286)), 300, 10, seed=0, points=20)),
                        ('RGM C=1000 B=500 ground (Gaussian pairwise)', synth.rgm_flat(C=1000, B=500, n_values=0, evidence_ratio=0.1, seed=0)[0])):
        vi = VarInference(None, K_, T_)
        vi.tiny_kernel = {'1': True, '0': False, 'always': 'always'}[os.environ.get('VI_TINY', '1')]
        vi._setup_flat(flat)
        np.random.seed(0)
        vi.init_param()
        t = ev_time(vi._grad)
        # algorithmic flops: per factor K * G * (K * a_h * 25 + 30), G = product of the axis lengths (T for a hidden continuous or
        # Gaussian-observed argument, #states for a hidden discrete one, 1 for an observed one), a_h = hidden arguments
        hid, cont, nst = flat.var_hidden, flat.var_cont, flat.var_nstates
        axis = np.where(hid[flat.edge_var], np.where(cont[flat.edge_var], T_, nst[flat.edge_var]), 1).astype(np.float64)
        G = np.multiply.reduceat(axis, flat.fac_ptr[:-1])
        ah = np.add.reduceat(hid[flat.edge_var].astype(np.float64), flat.fac_ptr[:-1])
        flop = float((K_ * G * (K_ * ah * 25 + 30)).sum())
        out(config='ground VI gradient + free energy, ' + label + ', K=2 T=3', factors=int(flat.F), edges=int(flat.E), grad_ms=t,
            kernel_split=dict(zip(('cc', 'tiny', 'grp3', 'grp6', 'rest3', 'rest6'), vi._fac_counts)),
            factors_per_s=flat.F / (t * 1e-3), quadrature_nodes=float((K_ * G).sum()), algorithmic_flop=flop,
            fp64_TFLOPs=flop / (t * 1e-3) / 1e12, fp64_frac_of_78_6=flop / (t * 1e-3) / 78.6e12)
        del vi

if 'gauss_probe' in which:
    # where the pull kernel's bytes go: the same launch with the partner slots replaced by the slot itself (no random access left)
    # and by a random permutation (every gather its own sector), next to the real plan; traffic from rocprofv3 --pmc on this command
    from lhvi.gabp import pull_plan
    flat = synth.random_gaussian_mrf(V=2_000_000, deg=4, seed=0)
    dg = _abi.DeviceGraph(flat)
    l, st = _abi.lib(), _abi.stream_ptr()
    host = pull_plan(flat)
    nnz = int(flat.var_edge.size)
    va, vb = dg.empty(nnz, 2), dg.empty(nnz, 2)
    rng = np.random.default_rng(0)
    for label, ps in (('real plan', host['pslot']), ('identity (no gather)', np.where(host['pslot'] >= 0, np.arange(nnz), host['pslot']).astype(np.int32)),
                      ('random permutation', np.where(host['pslot'] >= 0, rng.permutation(nnz), host['pslot']).astype(np.int32))):
        dev = dict(pslot=_abi.to_dev(ps), info=_abi.to_dev(host['info']))
        plan = _abi.GabpPlanStruct()
        plan.pslot, plan.info, plan.count = _abi.ptr(dev['pslot']), _abi.ptr(dev['info']), None
        plan.n_hub_rows = 0
        _abi.check(l.lhvi_gabp_pull(dg.g, dg.p, plan, _abi.ptr(va), _abi.ptr(vb), 1, st))
        va.fill_(1.0)
        t_p = ev_time(lambda: _abi.check(l.lhvi_gabp_pull(dg.g, dg.p, plan, _abi.ptr(va), _abi.ptr(vb), 0, st)))
        out(config='gauss probe: ' + label, slots=nnz, sweep_ms=t_p, algorithmic_frac=76.0 * flat.E / (t_p * 1e-3) / 8e12)

if 'demo_loop' in which:
    # Demo/RGM/demo.py:11-35 as a caller writes it, through the object API: ground the RGM template (100 x 10), observe
    # recession = 25, HybridLBP(g, n=10, 'simple').run(10, c2f=0), then `infer.map(rv)` for every one of the 1 111 rvs.  The loop
    # of per-variable calls is answered from one batched pass over the ground variables (default) or by one fminbound on the device
    # function per call (exact_queries = True).
    from lhvi import generators
    from lhvi.pbp import HybridLBP
    rel = generators.rgm(100, 10)
    rel.ground_graph()
    g, table = rel.add_evidence({('recession', 'all'): 25})
    g.rvs, g.factors = sorted(g.rvs), sorted(g.factors)
    g.init_nb()
    for rep in range(2):
        np.random.seed(0)
        infer = HybridLBP(g, n=10, proposal_approximation='simple')
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        infer.run(10, c2f=0)
        torch.cuda.synchronize()
        t_run = time.perf_counter() - t0
    t0 = time.perf_counter()
    maps = {rv: infer.map(rv) for rv in table.values()}
    t_loop = time.perf_counter() - t0
    some = list(table.values())[:40]
    infer.exact_queries = True
    t0 = time.perf_counter()
    exact = {rv: infer.map(rv) for rv in some}
    t_exact = (time.perf_counter() - t0) / len(some)
    hid = [rv for rv in some if rv.value is None]
    worse = [rv for rv in hid if infer.belief_rv_query(maps[rv], rv) < infer.belief_rv_query(exact[rv], rv) - 1e-6]
    out(config='Demo/RGM/demo.py through the object API: HybridLBP(g, n=10, simple).run(10, c2f=0), then map(rv) for all rvs',
        rvs=len(table), run_s=t_run, map_loop_s_batched=t_loop, per_call_ms_exact_fminbound=1e3 * t_exact,
        map_loop_s_exact_extrapolated=t_exact * len(table), max_abs_diff_batched_vs_exact=float(max(abs(maps[rv] - exact[rv]) for rv in hid)),
        batched_maps_with_a_lower_belief_than_exact=len(worse))
