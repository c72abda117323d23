"""CPU suite: the oracle reproduces the golden vectors captured from the reference (pins the oracle)."""
import json
import os

import numpy as np
import pytest

import modelio
from lhvi import graph as G, potentials as P, mln as M, lifting
from lhvi.flat import flatten
from oracle import oracle


class API:
    pass


for mod in (G, P, M):
    for k, v in vars(mod).items():
        if not k.startswith('_'):
            setattr(API, k, v)


def load(golden_dir, name):
    with open(os.path.join(golden_dir, name + '.json')) as fh:
        return json.load(fh)


def nan_equal(a, b, rtol=0.0, atol=0.0):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    assert a.shape == b.shape
    nan_a, nan_b = np.isnan(a), np.isnan(b)
    assert (nan_a == nan_b).all()
    inf = np.isinf(a) | np.isinf(b)
    assert (a[inf] == b[inf]).all()
    ok = ~nan_a & ~inf
    np.testing.assert_allclose(a[ok], b[ok], rtol=rtol, atol=atol)


@pytest.mark.parametrize('name', ['gauss_g1_chain', 'gauss_g2_kalman', 'gauss_g3_rgm0'])
def test_gabp_oracle_matches_reference(golden_dir, name):
    rec = load(golden_dir, name)
    g, rvs, factors = modelio.load_model(rec['model'], API)
    flat = flatten(g)
    hidden_edge = flat.var_hidden[flat.edge_var]
    for k, want in rec['sweeps'].items():
        f2v, v2f, mv = oracle.gabp_run(flat, int(k))
        # fp64, same operation order: the oracle is expected to be bit-identical to CPython here
        nan_equal(v2f, want['v2f'], rtol=1e-15)
        nan_equal(f2v[hidden_edge], np.asarray(want['f2v'], dtype=float)[hidden_edge], rtol=1e-15)
        nan_equal(mv, want['mu_var'], rtol=1e-15)


def _initial(flat, g):
    rv_color, f_color = lifting.initial_colors(g)
    sym = np.array([1 if getattr(f.potential, 'symmetric', False) else 0 for f in flat.factors])
    return sym, rv_color, f_color


def test_color_refinement_oracle_matches_reference(golden_dir):
    rec = load(golden_dir, 'color_partitions')
    for name, entry in rec.items():
        g, rvs, factors = modelio.load_model(entry['model'], API)
        flat = flatten(g)
        sym, rv0, f0 = _initial(flat, g)
        rv_color, f_color = oracle.color_passing(flat, sym, rv0, f0)
        assert oracle.canonical_labels(rv_color) == entry['rv_label'], name
        assert oracle.canonical_labels(f_color) == entry['f_label'], name
        assert int(rv_color.max()) + 1 == entry['n_rv'] and int(f_color.max()) + 1 == entry['n_f']
        rv_c, _ = lifting.initial_colors(g, is_split_cont_evidence=False)
        assert oracle.canonical_labels(rv_c) == entry['coarse_init_rv_label'], name
        if 'sorted_rv_label' in entry:   # CompressedGraphSorted gives the same partition without evidence
            assert oracle.canonical_labels(rv_color) == entry['sorted_rv_label']
            assert oracle.canonical_labels(f_color) == entry['sorted_f_label']


@pytest.mark.parametrize('name', ['gauss_g1_chain', 'gauss_g2_kalman', 'gauss_g3_rgm0'])
def test_galbp_oracle_matches_reference(golden_dir, name):
    """lifted sweep = colour passing + counted sweep; compare the per-ground-rv MAP with the reference's GaLBP"""
    rec = load(golden_dir, name)
    g, rvs, factors = modelio.load_model(rec['model'], API)
    flat = flatten(g)
    sym, rv0, f0 = _initial(flat, g)
    rv_color, f_color = oracle.color_passing(flat, sym, rv0, f0)
    assert oracle.canonical_labels(rv_color) == rec['galbp']['rv_label']
    assert oracle.canonical_labels(f_color) == rec['galbp']['f_label']
    cg = lifting.CompressedGraph(g)
    cg.set_colors(rv_color, f_color)
    lflat = flatten(cg)
    _, _, mv = oracle.gabp_run(lflat, rec['galbp']['iterations'])
    got = np.array([mv[lflat.var_index[rv.cluster], 0] for rv in rvs])
    # summation order inside a lifted cluster follows the representative's rv.nb, which is a set-order
    # artefact in the reference: allow a few ulp
    np.testing.assert_allclose(got, rec['galbp']['map'], rtol=1e-12, atol=1e-13)
    # the vectorised lifter must describe the same lifted graph
    lf2 = lifting.lift_flat(flat, rv_color, f_color)
    _, _, mv2 = oracle.gabp_run(lf2, rec['galbp']['iterations'])
    np.testing.assert_allclose(mv2[rv_color, 0], rec['galbp']['map'], rtol=1e-12, atol=1e-13)
