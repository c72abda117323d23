"""GPU edge cases through the C ABI: empty graph, everything observed, isolated / unary-only variables, high degree,
ragged particle counts, argument validation."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def api():
    from lhvi import _abi
    _abi.require_gpu()
    return _abi


def _chain(n, observed=()):
    from lhvi.graph import Domain, F, Graph, RV
    from lhvi.potentials import LinearGaussianPotential, X2Potential
    d = Domain((-10, 10), continuous=True, integral_points=np.linspace(-10, 10, 16))
    rvs = [RV(d, 0.5 * i if i in observed else None) for i in range(n)]
    fs = [F(LinearGaussianPotential(0.8, 1.0), [rvs[i], rvs[i + 1]]) for i in range(n - 1)]
    fs += [F(X2Potential(1.0, 3.0), [rv]) for rv in rvs]
    g = Graph()
    g.rvs, g.factors = rvs, fs
    g.init_nb()
    return g, rvs


def test_empty_graph_is_a_no_op(api):
    from lhvi.graph import Graph
    from lhvi.gabp import GaBP
    g = Graph()
    g.rvs, g.factors = [], []
    bp = GaBP(g)
    bp.run(3)
    assert bp.message == {}


def test_everything_observed(api):
    from lhvi.gabp import GaBP
    from lhvi.pbp import EPBP
    g, rvs = _chain(4, observed=range(4))
    bp = GaBP(g)
    bp.run(5)
    assert [bp.map(rv) for rv in rvs] == [rv.value for rv in rvs]
    ep = EPBP(g, n=8, proposal_approximation='simple')
    ep.run(3)
    assert ep.map(rvs[2]) == rvs[2].value and ep.belief(rvs[2].value, rvs[2]) == 1


def test_degree_one_hidden_variable_raises_like_the_reference(api):
    from lhvi.graph import Domain, F, Graph, RV
    from lhvi.potentials import LinearGaussianPotential
    from lhvi.gabp import GaBP
    d = Domain((-5, 5), continuous=True)
    a, b = RV(d), RV(d, 1.0)
    g = Graph()
    g.rvs, g.factors = [a, b], [F(LinearGaussianPotential(1.0, 1.0), [a, b])]
    g.init_nb()
    with pytest.raises(ZeroDivisionError):
        GaBP(g).run(3)


def test_isolated_hidden_variable_raises_like_the_reference(api):
    """a hidden continuous variable without factors: the reference divides by zero in gaussian_product (EPBP:30-41) on its
    first proposal update; the device path must not quietly produce NaN particles"""
    from lhvi.graph import Domain, RV
    from lhvi.pbp import EPBP
    g, rvs = _chain(3)
    g.rvs = rvs + [RV(Domain((-5, 5), continuous=True))]
    g.init_nb()
    with pytest.raises(ZeroDivisionError):
        EPBP(g, n=8, proposal_approximation='simple').run(3)


def test_high_degree_hub_matches_oracle(api):
    """star: one hub with 300 pairwise neighbours (exercises the O(deg) v2f path and long CSR rows)"""
    from lhvi import synth, _abi
    from lhvi.flat import build_flat
    from lhvi.graph import Domain
    from lhvi import potentials as P
    from lhvi.pbp import EPBP
    from oracle import oracle
    D = 300
    dom = Domain((-10, 10), continuous=True, integral_points=np.linspace(-10, 10, 32))
    edge_var = np.stack([np.zeros(D, dtype=np.int32), np.arange(1, D + 1, dtype=np.int32)], axis=1).ravel()
    edge_var = np.concatenate([edge_var, np.arange(D + 1, dtype=np.int32)])
    fac_ptr = np.concatenate([np.arange(0, 2 * D + 1, 2), 2 * D + np.arange(1, D + 2)]).astype(np.int32)
    specs = [(P.POT_LINEAR_GAUSSIAN, [0.7, 2.0]), (P.POT_X2, [1.0, 4.0])]
    fac_pot = np.concatenate([np.zeros(D), np.ones(D + 1)]).astype(np.int32)
    value = np.full(D + 1, np.nan)
    value[5::7] = 1.0
    flat = build_flat(fac_ptr, edge_var, fac_pot, specs, value, np.zeros(D + 1, dtype=np.int32), [dom])
    n = 32
    bp = EPBP(None, n=n, proposal_approximation='EP', sampler='device', seed=4)
    bp._setup(None, flat=flat)
    l, st = api.lib(), api.stream_ptr()
    api.check(l.lhvi_pbp_init(bp.dg.g, bp._struct(), api.ptr(bp.eta), api.ptr(bp.q_dev), api.ptr(bp.f2v), api.ptr(bp.v2f), st))
    bp._generate_sample()
    o = oracle.PbpOracle(flat, n, ep=True, epbp=True, var_threshold=3)
    o.init()
    o.set_particles(bp.particles.cpu().numpy())
    hid_e = flat.var_hidden[flat.edge_var]
    for _ in range(2):
        bp.sweep(last=False)
        o.step_v2f(); o.step_proposal(); o.set_particles(bp.particles.cpu().numpy()); o.step_f2v()
        np.testing.assert_allclose(bp.v2f.cpu().numpy()[hid_e][:, :n], o.v2f[hid_e], rtol=1e-9, atol=1e-7)
        np.testing.assert_allclose(bp.f2v.cpu().numpy()[hid_e], o.f2v[hid_e], rtol=1e-9, atol=1e-7)
        np.testing.assert_allclose(bp.q_dev.cpu().numpy()[flat.var_hidden], o.q[flat.var_hidden], rtol=1e-9, atol=1e-10)
    # Gaussian sweep on the same star
    dg = _abi.DeviceGraph(flat)
    f2v, v2f, mv = dg.empty(flat.E, 2), dg.empty(flat.E, 2), dg.empty(flat.V, 2)
    api.check(l.lhvi_gabp_run(dg.g, dg.p, api.ptr(f2v), api.ptr(v2f), 6, st))
    api.check(l.lhvi_gabp_marginals(dg.g, api.ptr(f2v), api.ptr(mv), st))
    _, _, omv = oracle.gabp_run(flat, 6)
    np.testing.assert_allclose(mv.cpu().numpy(), omv, rtol=1e-11, atol=1e-12)


@pytest.mark.parametrize('n,T', [(5, 20), (100, 20), (50, 32), (64, 64), (64, 70), (33, 7), (64, 0)])
def test_particle_counts_and_grid_sizes(api, n, T):
    """shapes around the kernels' fast paths, against the oracle: n = 5 (tiny), n = 100 (> one wavefront: chunked v2f,
    tiled f2v staging, generic uniq), n = 50 (the reference's default), n + T = 128 (two full rounds, the heavy kernel's
    limit), n + T = 134 (three rounds, general kernel), odd sizes, and no integral points at all"""
    from lhvi import synth
    from lhvi.pbp import EPBP
    from oracle import oracle
    flat = synth.hybrid_mrf_flat(V=400, deg=4, seed=9, T=T)
    bp = EPBP(None, n=n, proposal_approximation='simple', sampler='device', seed=n)
    bp._setup(None, flat=flat)
    l, st = api.lib(), api.stream_ptr()
    api.check(l.lhvi_pbp_init(bp.dg.g, bp._struct(), api.ptr(bp.eta), api.ptr(bp.q_dev), api.ptr(bp.f2v), api.ptr(bp.v2f), st))
    bp._generate_sample()
    o = oracle.PbpOracle(flat, n, ep=False, epbp=True, var_threshold=3)
    o.init()
    o.set_particles(bp.particles.cpu().numpy())
    assert (o.uniq == bp.uniq.cpu().numpy()).all()
    hid_e = flat.var_hidden[flat.edge_var]
    for _ in range(2):
        bp.sweep(last=False)
        o.step_v2f(); o.step_proposal(); o.set_particles(bp.particles.cpu().numpy()); o.step_f2v()
        np.testing.assert_allclose(bp.v2f.cpu().numpy()[hid_e], o.v2f[hid_e], rtol=1e-9, atol=1e-7)
        np.testing.assert_allclose(bp.f2v.cpu().numpy()[hid_e], o.f2v[hid_e], rtol=1e-9, atol=1e-7)


def test_argument_validation_returns_codes(api):
    l = api.lib()
    assert l.lhvi_gabp_v2f(None, None, None, None) == -1
    g = api.GraphStruct()
    g.V, g.E, g.F, g.nnz = 1, 2, 1, 2          # sizes without arrays
    assert l.lhvi_gabp_v2f(C.byref(g), None, None, None) == -1
    assert l.lhvi_adam_step(None, None, None, None, 10, 1, 0.1, 0.9, 0.999, 1e-8, 0, 0.0, None) == -1
    assert l.lhvi_adam_step(None, None, None, None, 0, 1, 0.1, 0.9, 0.999, 1e-8, 0, 0.0, None) == 0
    assert l.lhvi_softmax_rows(None, None, 4, 0, 0, None) == -1
    assert l.lhvi_strerror(-3) == b'unsupported configuration'


def test_gaussian_hub_above_the_direct_sum_threshold(api):
    """a star with 700 pairwise neighbours: the hub takes the wave-parallel total-minus-own path of the Gaussian sweep
    (beyond 512 edges the reference-order direct sum is quadratic in the degree); marginals against the oracle"""
    from lhvi.flat import build_flat
    from lhvi.graph import Domain
    from lhvi import potentials as P
    from oracle import oracle
    D = 700
    dom = Domain((-10, 10), continuous=True, integral_points=np.linspace(-10, 10, 8))
    edge_var = np.stack([np.zeros(D, dtype=np.int32), np.arange(1, D + 1, dtype=np.int32)], axis=1).ravel()
    edge_var = np.concatenate([edge_var, np.arange(D + 1, dtype=np.int32)])
    fac_ptr = np.concatenate([np.arange(0, 2 * D + 1, 2), 2 * D + np.arange(1, D + 2)]).astype(np.int32)
    specs = [(P.POT_LINEAR_GAUSSIAN, [0.7, 2.0]), (P.POT_X2, [1.0, 4.0])]
    fac_pot = np.concatenate([np.zeros(D), np.ones(D + 1)]).astype(np.int32)
    value = np.full(D + 1, np.nan)
    value[5::7] = 1.0
    flat = build_flat(fac_ptr, edge_var, fac_pot, specs, value, np.zeros(D + 1, dtype=np.int32), [dom])
    assert np.diff(flat.var_ptr).max() > 512
    dg = api.DeviceGraph(flat)
    assert dg.g.n_hubs == 1
    f2v, v2f, mv = dg.empty(flat.E, 2), dg.empty(flat.E, 2), dg.empty(flat.V, 2)
    l, st = api.lib(), api.stream_ptr()
    api.check(l.lhvi_gabp_run(dg.g, dg.p, api.ptr(f2v), api.ptr(v2f), 6, st))
    api.check(l.lhvi_gabp_marginals(dg.g, api.ptr(f2v), api.ptr(mv), st))
    _, _, omv = oracle.gabp_run(flat, 6)
    np.testing.assert_allclose(mv.cpu().numpy(), omv, rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize('frac_discrete,evidence', [(0.0, 0.0), (0.5, 0.5), (0.9, 0.2), (1.0, 0.1)])
def test_graph_compositions_against_oracle(api, frac_discrete, evidence):
    """all-continuous (heavy kernel only), half discrete with heavy evidence (light kernel, observed partners on both
    sides), almost all discrete and all discrete (generic table kernel only): two sweeps against the oracle"""
    from lhvi import synth
    from lhvi.pbp import EPBP
    from oracle import oracle
    flat = synth.hybrid_mrf_flat(V=600, deg=4, seed=21, frac_discrete=frac_discrete, evidence_ratio=evidence, T=16)
    n = 32
    bp = EPBP(None, n=n, proposal_approximation='EP', sampler='device', seed=3)
    bp._setup(None, flat=flat)
    l, st = api.lib(), api.stream_ptr()
    api.check(l.lhvi_pbp_init(bp.dg.g, bp._struct(), api.ptr(bp.eta), api.ptr(bp.q_dev), api.ptr(bp.f2v), api.ptr(bp.v2f), st))
    bp._generate_sample()
    o = oracle.PbpOracle(flat, n, ep=True, epbp=True, var_threshold=3)
    o.init()
    o.set_particles(bp.particles.cpu().numpy())
    hid_e = flat.var_hidden[flat.edge_var]
    for _ in range(2):
        bp.sweep(last=False)
        o.step_v2f(); o.step_proposal(); o.set_particles(bp.particles.cpu().numpy()); o.step_f2v()
        np.testing.assert_allclose(bp.v2f.cpu().numpy()[hid_e], o.v2f[hid_e], rtol=1e-9, atol=1e-7)
        np.testing.assert_allclose(bp.f2v.cpu().numpy()[hid_e], o.f2v[hid_e], rtol=1e-9, atol=1e-7)
        np.testing.assert_allclose(bp.q_dev.cpu().numpy()[flat.var_hidden], o.q[flat.var_hidden], rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize('seed', range(8))
def test_random_shapes_against_oracle(api, seed):
    """random mixtures of what the work lists are made of -- particle count, grid size, degree, share of discrete variables,
    evidence, proposal rule -- three sweeps each against the oracle with the device's particles"""
    from lhvi import synth
    from lhvi.pbp import EPBP
    from oracle import oracle
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([8, 16, 24, 33, 48, 64]))
    T = int(rng.choice([0, 9, 16, 32, 40]))
    deg = int(rng.choice([3, 4, 6]))
    V = int(rng.integers(300, 900)) // 2 * 2 * (2 if deg % 2 else 1)
    flat = synth.hybrid_mrf_flat(V=V, deg=deg, seed=int(rng.integers(1 << 20)), frac_discrete=float(rng.uniform(0, 0.7)),
                                 evidence_ratio=float(rng.uniform(0, 0.5)), T=T)
    ep = bool(rng.integers(2)) and T > 0
    bp = EPBP(None, n=n, proposal_approximation='EP' if ep else 'simple', sampler='device', seed=seed)
    bp._setup(None, flat=flat)
    l, st = api.lib(), api.stream_ptr()
    api.check(l.lhvi_pbp_init(bp.dg.g, bp._struct(), api.ptr(bp.eta), api.ptr(bp.q_dev), api.ptr(bp.f2v), api.ptr(bp.v2f), st))
    bp._generate_sample()
    o = oracle.PbpOracle(flat, n, ep=ep, epbp=True, var_threshold=3)
    o.init()
    o.set_particles(bp.particles.cpu().numpy())
    hid_e = flat.var_hidden[flat.edge_var]
    for _ in range(3):
        bp.sweep(last=False)
        o.step_v2f(); o.step_proposal(); o.set_particles(bp.particles.cpu().numpy()); o.step_f2v()
        np.testing.assert_allclose(bp.v2f.cpu().numpy()[hid_e][:, :n], o.v2f[hid_e], rtol=1e-9, atol=1e-7)
        np.testing.assert_allclose(bp.f2v.cpu().numpy()[hid_e], o.f2v[hid_e], rtol=1e-9, atol=1e-7)
        np.testing.assert_allclose(bp.q_dev.cpu().numpy()[flat.var_hidden], o.q[flat.var_hidden], rtol=1e-9, atol=1e-10)


def test_build_then_smoke_in_one_fresh_process():
    """``__graft_entry__.build()`` followed by ``smoke()`` in ONE process that has not imported torch before: ``build()`` loads
    liblhvi.so, and unless PyTorch's own HIP runtime is in the process first the library binds to the system's copy and the first
    launch fails with hipErrorNoDevice (seen in round 2; ``lhvi/_abi.py::lib`` imports torch before ``dlopen``)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; assert 'torch' not in sys.modules; import __graft_entry__ as g; g.build(); g.smoke(); print('BUILD+SMOKE OK')")
    out = subprocess.run([sys.executable, '-c', code], cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and 'BUILD+SMOKE OK' in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_first_members_of_colours():
    """``lhvi_color_first_members`` (wavefront-aggregated atomicMin): one giant colour, many small ones, colours without members,
    n not a multiple of the block"""
    import torch
    from lhvi import lifting
    rng = np.random.default_rng(0)
    for n, ncol, giant in ((1, 1, 0.0), (1000, 7, 0.9), (300_001, 5000, 0.6), (2_000_003, 3, 0.0), (70_000, 70_000, 0.0)):
        col = rng.integers(0, ncol, n)
        col[rng.random(n) < giant] = min(2, ncol - 1)
        if ncol > 10:
            col[col == 5] = 6                           # colour 5 has no member
        want = np.full(ncol, n, dtype=np.int64)
        np.minimum.at(want, col, np.arange(n))
        got = lifting.first_members(torch.from_numpy(col.astype(np.int32)).cuda(), ncol, n).cpu().numpy()
        np.testing.assert_array_equal(got, want)


def test_segment_sums_are_running_sums_in_index_order():
    """``lhvi_color_segment_sums``: every segment summed left to right, bit for bit what NumPy's sequential ``add.at`` gives (torch's
    segmented reduction on the GPU adds in a tree: a cluster's evidence value must not depend on where it was summed); empty
    segments, one huge segment, a million small ones"""
    import torch
    from lhvi import lifting
    rng = np.random.default_rng(0)
    for n, nseg in ((0, 3), (1000, 37), (250_000, 5), (1_000_000, 400_000)):
        vals = rng.uniform(-30, 30, n)
        seg = np.sort(rng.integers(0, nseg, n)) if n else np.zeros(0, dtype=np.int64)
        lengths = np.bincount(seg, minlength=nseg)
        want = np.zeros(nseg)
        np.add.at(want, seg, vals)
        got = lifting.segment_sums(torch.from_numpy(vals).cuda(), torch.from_numpy(lengths).cuda()).cpu().numpy()
        assert got.tobytes() == want.tobytes()


def test_initial_colours_on_the_device_equal_the_host_numbering():
    """``lifting.initial_colors_device``: same colour ids (numbered by first appearance) as ``initial_colors_flat``, with and without
    the split of continuous evidence by value; -0.0 and 0.0 are one evidence value; a graph with discrete evidence and equal
    potentials on different table rows"""
    from lhvi import _abi, lifting, synth
    cases = [synth.rgm_flat(C=30, B=20, n_values=5, evidence_ratio=0.3, seed=2)[0], synth.paper_popularity_flat(30, 4, seed=1, points=8)[0]]
    flat = synth.rgm_flat(C=12, B=6, n_values=0, evidence_ratio=0.4, seed=5)[0]
    val = flat.var_value.copy()
    obs = np.flatnonzero(~np.isnan(val))
    val[obs[:4]] = [0.0, -0.0, 0.0, -0.0]
    flat.var_value = val
    cases.append(flat)
    for flat in cases:
        dg = _abi.DeviceGraph(flat)
        for split in (True, False):
            want = lifting.initial_colors_flat(flat, split)
            got = lifting.initial_colors_device(flat, dg, split)
            for w, g_, what in zip(want, got, ('rv colours', 'factor colours', 'symmetric flags')):
                np.testing.assert_array_equal(g_.cpu().numpy(), w, err_msg='%s (split=%s)' % (what, split))


def test_image_potentials_on_a_denoising_grid_match_the_oracle():
    """``ImageNodePotential`` / ``ImageEdgePotential`` (Potential.py:400-424; the model of Demo/old/DenoisingDemo.py:20-60: a hidden
    pixel per cell tied to its noisy observation, a truncated-exponential smoothness prior between 4-neighbours, domain
    (-30, 130)) on a 32 x 32 grid: three EPBP sweeps through the C ABI against the C oracle with the same particles"""
    import torch
    from lhvi import _abi, graph as G, potentials as P
    from lhvi.flat import flatten
    from lhvi.pbp import EPBP
    from oracle import oracle
    _abi.require_gpu()
    rows = cols = 32
    rng = np.random.default_rng(0)
    img = np.where((np.arange(rows)[:, None] // 8 + np.arange(cols)[None, :] // 8) % 2 == 0, 20.0, 90.0) + rng.normal(0, 8, (rows, cols))
    dom = G.Domain((-30, 130), continuous=True, integral_points=np.linspace(-30, 130, 32))
    ev = [G.RV(dom, float(img[i, j])) for i in range(rows) for j in range(cols)]
    rvs = [G.RV(dom) for _ in range(rows * cols)]
    pxo, pxy = P.ImageNodePotential(0, 5), P.ImageEdgePotential(0, 3.5, 25)
    fs = [G.F(pxo, (rvs[k], ev[k])) for k in range(rows * cols)]
    fs += [G.F(pxy, (rvs[i * cols + j], rvs[i * cols + j + 1])) for i in range(rows) for j in range(cols - 1)]
    fs += [G.F(pxy, (rvs[i * cols + j], rvs[(i + 1) * cols + j])) for i in range(rows - 1) for j in range(cols)]
    g = G.Graph()
    g.rvs, g.factors = rvs + ev, fs
    g.init_nb()
    flat = flatten(g, require_device_potentials=True)
    assert set(flat.pot_kind.tolist()) == {P.POT_IMAGE_NODE, P.POT_IMAGE_EDGE}
    n = 16
    bp = EPBP(g, n=n, proposal_approximation='simple', sampler='device', seed=4)
    bp._setup(g)
    l = _abi.lib()
    _abi.check(l.lhvi_pbp_init(bp.dg.g, bp._struct(), _abi.ptr(bp.eta), _abi.ptr(bp.q_dev), _abi.ptr(bp.f2v), _abi.ptr(bp.v2f), _abi.stream_ptr()))
    # a first sample around the observations (the initial proposal N(0, 5) would put every particle at one end of the domain)
    obs = np.concatenate([img.ravel(), img.ravel()])
    first = np.clip(obs[:, None] + rng.normal(0, 10, (flat.V, n)), -30, 130)
    bp.sampler = lambda k, f, q: first
    bp._generate_sample()
    bp.sampler = 'device'
    o = oracle.PbpOracle(flat, n, ep=False, epbp=True, var_threshold=3)
    o.init()
    o.set_particles(bp.particles.cpu().numpy())
    hid_e = flat.var_hidden[flat.edge_var]
    hid = flat.var_hidden
    for it in range(3):
        bp.sweep(last=False)
        o.step_v2f()
        o.step_proposal()
        o.set_particles(bp.particles.cpu().numpy())
        o.step_f2v()
        np.testing.assert_allclose(bp.v2f.cpu().numpy()[hid_e], o.v2f[hid_e], rtol=1e-9, atol=1e-8)
        np.testing.assert_allclose(bp.q_dev.cpu().numpy()[hid], o.q[hid], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(bp.f2v.cpu().numpy()[hid_e], o.f2v[hid_e], rtol=1e-9, atol=1e-8)
    assert int(bp.generic_edges.numel()) == int(hid_e.sum())            # the image kinds have no closed family: generic kernel
    # the denoised image: MAP of every pixel from one batched pass, inside the domain and closer to the clean image than the noise
    mp = np.array([bp.map(rv) for rv in rvs]).reshape(rows, cols)
    clean = np.where((np.arange(rows)[:, None] // 8 + np.arange(cols)[None, :] // 8) % 2 == 0, 20.0, 90.0)
    assert np.isfinite(mp).all() and mp.min() >= -30 and mp.max() <= 130
    assert np.abs(mp - clean).mean() < np.abs(img - clean).mean()


def test_full_size_properties_of_cfg5():
    """BASELINE cfg 5 at its full size (10.0 M ground edges, where no oracle finishes), through size-independent properties
    (`scripts/bench_configs.py full_size`): hash-table and radix-sort refinement give the same colours; device and host lifting the
    same graph; the counted Gaussian sweep on the 39 k lifted edges gives every ground variable the ground sweep's marginal; the
    lifted free energy, mixture-weight gradient and per-cluster gradients equal the ground ones at tied parameters"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'bench_configs.py'), 'full_size'], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert d['ground_edges'] >= 10_000_000 and d['lifted_edges'] < 50_000
    assert d['hash_and_sort_refinement_same_colours'] and d['device_and_host_lift_same_graph']
    assert d['max_abs_mu_lifted_vs_ground'] < 1e-12 and d['max_rel_var_lifted_vs_ground'] < 1e-12
    assert d['rel_diff_free_energy'] < 1e-12 and d['max_rel_diff_g_w'] < 1e-11 and d['max_rel_diff_member_gradient_vs_cluster_gradient'] < 1e-12
