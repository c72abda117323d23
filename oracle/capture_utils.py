#!/usr/bin/env python3
"""Golden values of the reference's evaluation helpers (``utils.py``: log_likelihood, KL, kl_discrete, kl_continuous*,
kl_normal) -> tests/golden/utils.json.

TEST INFRASTRUCTURE, build container only (imports /root/reference, read-only; see capture_golden.py).  Only data is
written: the serialised models, the assignments / densities' parameters and the reference's results.
usage: python oracle/capture_utils.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import capture_golden as cg  # noqa: E402  (installs the aliases, puts the reference on sys.path)
from capture_pbp import model_hybrid_small, model_hmln_small  # noqa: E402


def norm_pdf(x, mu, sig):
    u = (x - mu) / sig
    return np.exp(-u * u * 0.5) / (2.506628274631 * sig)


def main():
    import utils as RU
    out = {'log_likelihood': [], 'kl': []}
    rng = np.random.RandomState(3)
    for name, g in (('chain', cg.model_chain()), ('kalman', cg.model_kalman(3, 4, 1)), ('hmln', model_hmln_small(cg)),
                    ('rgm0', cg.model_rgm(0))):
        rvs = list(g.rvs)
        for rep in range(2):
            x = []
            for rv in rvs:
                if rv.value is not None:
                    x.append(float(rv.value))
                elif rv.domain.continuous:
                    x.append(float(rng.uniform(rv.domain.values[0], rv.domain.values[1]) * 0.3))
                else:
                    x.append(float(rv.domain.values[rng.randint(len(rv.domain.values))]))
            asg = {rv: (v if rv.domain.continuous else int(v)) for rv, v in zip(rvs, x)}
            out['log_likelihood'].append({'name': name, 'model': cg.modelio.dump_model(g), 'x': x,
                                          'value': float(RU.log_likelihood(g, asg))})
    # (the reference's log_likelihood passes a *list* to potential.get, which TablePotential cannot index: models with
    # tables are not evaluable there.)  A vanishing factor: the reference returns -inf
    RG, RP, RM = cg.RG, cg.RP, cg.RM
    d = RG.Domain((0, 1))
    a, b = RG.RV(d), RG.RV(d)
    g = RG.Graph()
    g.rvs = [a, b]
    g.factors = [RG.F(RM.MLNHardPotential(cg.modelio.FORMULAS['x0']), [a]), RG.F(RM.MLNPotential(cg.modelio.FORMULAS['nand'], 0.7), [a, b])]
    g.init_nb()
    out['log_likelihood'].append({'name': 'zero', 'model': cg.modelio.dump_model(g), 'x': [0.0, 1.0],
                                  'value': float(RU.log_likelihood(g, {a: 0, b: 1}))})
    dom_c = RG.Domain((-6, 6), continuous=True, integral_points=np.linspace(-6, 6, 41))
    dom_d = RG.Domain((0, 1, 2))
    for mu1, s1, mu2, s2 in ((0.3, 0.8, -0.4, 1.1), (1.0, 0.5, 1.0, 0.5), (-1.2, 1.4, 0.9, 0.7)):
        p = lambda x: norm_pdf(x, mu1, s1)
        q = lambda x: norm_pdf(x, mu2, s2)
        lp = lambda x: -((x - mu1) / s1) ** 2 * 0.5 - np.log(2.506628274631 * s1)
        lq = lambda x: -((x - mu2) / s2) ** 2 * 0.5 - np.log(2.506628274631 * s2)
        out['kl'].append({'mu1': mu1, 's1': s1, 'mu2': mu2, 's2': s2, 'lo': -6, 'hi': 6, 'points': 41,
                          'KL': float(RU.KL(p, q, dom_c)),
                          'kl_continuous': float(RU.kl_continuous(p, q, -6, 6)),
                          'kl_continuous_no_add_const': float(RU.kl_continuous_no_add_const(p, q, -6, 6)),
                          'kl_continuous_logpdf': float(RU.kl_continuous_logpdf(lp, lq, -6, 6)),
                          'kl_normal': float(RU.kl_normal(mu1, mu2, s1, s2))})
    tp, tq = np.array([0.2, 0.5, 0.3]), np.array([0.4, 0.4, 0.2])
    out['discrete'] = {'p': tp.tolist(), 'q': tq.tolist(), 'kl_discrete': float(RU.kl_discrete(tp, tq)),
                       'KL': float(RU.KL(lambda x: tp[x], lambda x: tq[x], dom_d))}
    cg.save('utils', out)


if __name__ == '__main__':
    os.chdir(cg.REF)
    main()
