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
