#!/usr/bin/env python3
"""Golden vectors of the reference's third demo family, the relational Kalman filters on well data
(Demo/RKF/LRKFDemoTree.py:16-60, Demo/RKF/LRKFDemoCycle.py:16-66) -> tests/golden/rkf.npz, tests/golden/vi_c2f_rkf_{tree,cycle}_k1.npz.

TEST INFRASTRUCTURE, build container only (imports /root/reference, read-only; see capture_golden.py).  Only data is written:
the arrays the demos derive from their .mat inputs (the observations of the selected wells, the three filter parameters of
parameter set 0, the MATLAB answers), and what the reference computes on the graph its own KalmanFilter builds from them --
GaBP(20) marginals, the GaLBP partition and marginals, C2FVarInference(g, 1, 3) for three rounds (through capture_vi.py).
usage: python oracle/capture_rkf.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import capture_golden as cg  # noqa: E402  (installs the aliases, puts the reference on sys.path)
import capture_vi  # noqa: E402

REF_DATA = os.path.join(cg.REF, 'Demo', 'Data', 'RKF')
T_STEPS = 20


def demo_inputs(which):
    """the arrays LRKFDemoTree.py:16-38 / LRKFDemoCycle.py:16-41 derive from the .mat files"""
    import scipy.io
    cluster_mat = scipy.io.loadmat(os.path.join(REF_DATA, 'cluster_NcutDiscrete.mat'))['NcutDiscrete'].copy()
    well_t = scipy.io.loadmat(os.path.join(REF_DATA, 'well_t.mat'))['well_t'].astype(np.float64)
    mat = scipy.io.loadmat(os.path.join(REF_DATA, 'LRKF_%s.mat' % which))
    ans, param = mat['res'], mat['param']
    if which == 'cycle':
        idx = np.where(cluster_mat[:, 1] == 1)[0]
        cluster_mat[idx[3:], 1] = 0
        idx = np.where(cluster_mat[:, 2] == 1)[0]
        cluster_mat[idx[:49], 2] = 0
        cluster_mat[idx[52:], 2] = 0
    well_t = well_t[:, 199:]
    well_t[well_t[:, 0] == 5000, 0] = 0
    well_t[well_t == 5000] = 1
    cluster_id = [1] if which == 'tree' else [1, 2]
    rvs_id = np.concatenate([np.where(cluster_mat[:, i] == 1)[0] for i in cluster_id], axis=None)
    data = well_t[rvs_id, :T_STEPS]
    return data, np.asarray(param, dtype=np.float64), np.asarray(ans, dtype=np.float64)


def build(which, data, param, i=0):
    import KalmanFilter as RK
    n = data.shape[0]
    domain = cg.RG.Domain((-4, 4), continuous=True, integral_points=np.linspace(-4, 4, 30))
    A = np.eye(n) * param[2, i] + (0.01 if which == 'cycle' else 0.0)
    kmf = RK.KalmanFilter(domain, A, param[0, i], np.eye(n), param[1, i])
    g, table = kmf.grounded_graph(T_STEPS, data)
    return g, table


def main():
    import GaBP as RGaBP
    import GaLBP as RGaLBP
    rec = {}
    for which in ('tree', 'cycle'):
        data, param, ans = demo_inputs(which)
        g, table = build(which, data, param)
        rvs = list(g.rvs)
        print(which, 'rvs', len(rvs), 'factors', len(g.factors), 'edges', sum(len(f.nb) for f in g.factors),
              'observed', sum(rv.value is not None for rv in rvs))
        bp = RGaBP.GaBP(g)
        with cg.quiet():
            bp.run(20)
        marg = np.array([list(bp.get_belief_params(rv)) if rv.value is None else [rv.value, 0.0] for rv in rvs], dtype=np.float64)
        lbp = RGaLBP.GaLBP(g)
        with cg.quiet():
            lbp.run(20)
        rv_label = np.array(cg.partition_labels(rvs, lbp.g.rvs, 'rvs'))
        f_label = np.array(cg.partition_labels(list(g.factors), lbp.g.factors, 'factors'))
        lmap = np.array([float(lbp.map(rv)) for rv in rvs])
        print('   GaLBP clusters', len(set(rv_label.tolist())), len(set(f_label.tolist())), 'max |GaLBP - GaBP| mu', float(np.abs(lmap - marg[:, 0]).max()))
        last = np.array([rvs.index(rv) for rv in table[T_STEPS - 1]])
        rec.update({which + '_data': data, which + '_param': param, which + '_res': ans, which + '_gabp': marg,
                    which + '_galbp_rv_label': rv_label, which + '_galbp_f_label': f_label, which + '_galbp_map': lmap,
                    which + '_last_step': last, which + '_sizes': np.array([len(rvs), len(g.factors), sum(len(f.nb) for f in g.factors)])})
        # C2FVarInference(g, 1, 3), three rounds of ten updates (the demo runs 200 updates at lr = 0.1)
        g2, _ = build(which, data, param)
        capture_vi.capture_c2fvi(cg, 'c2f_rkf_%s_k1' % which, g2, 1, 3, 50 + len(which), 30, 0.1)
    path = os.path.join(cg.OUT, 'rkf.npz')
    np.savez_compressed(path, **rec)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
