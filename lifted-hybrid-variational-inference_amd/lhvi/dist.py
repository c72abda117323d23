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


def joint_terms(flat, np_host):
    """number of (output point, joint partner particle) terms one f2v launch evaluates"""
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
    return int(terms.sum())


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
