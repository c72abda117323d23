"""Batched device queries on the headline graph: log-beliefs of every variable at n points, and MAP of every variable,
against the per-variable query path the reference API maps to (one launch per variable).
usage: python scripts/bench_queries.py [edges] [out.json]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'lifted-hybrid-variational-inference_amd')]
import numpy as np, torch
from lhvi import _abi, synth, dist
from lhvi.pbp import EPBP

E = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
n = 64
flat = synth.hybrid_mrf_flat(V=E // 4, deg=4, seed=0)
bp = EPBP(None, n=n, proposal_approximation='simple', sampler='device', seed=1)
bp._setup(None, flat=flat)
run = dist.SingleRunner(bp)
run.init()
for _ in range(3):
    run.sweep()
bp.sweep(last=True)
torch.cuda.synchronize()
hidden = int(flat.var_hidden.sum())
x = bp.particles.clone()
bp.belief_rv_all(x); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    b = bp.belief_rv_all(x)
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / 3
t0 = time.perf_counter()
mp, mv = bp.map_all(steps=5)
torch.cuda.synchronize()
t_map = time.perf_counter() - t0
# normalised beliefs at 8 points per variable (20-point trapezoid normaliser) and interval probabilities
lo = flat.dom_lo[flat.var_dom]
bp.probability_all(lo + 1.0, lo + 3.0); torch.cuda.synchronize()
t0 = time.perf_counter()
ball = bp.belief_all(x[:, :8].contiguous())
torch.cuda.synchronize()
t_bel = time.perf_counter() - t0
t0 = time.perf_counter()
pall = bp.probability_all(lo + 1.0, lo + 3.0)
torch.cuda.synchronize()
t_prob = time.perf_counter() - t0
cont = torch.from_numpy(flat.var_hidden & flat.var_cont).to(pall.device)
assert bool(torch.isfinite(pall[cont]).all()) and bool(((pall[cont] >= 0) & (pall[cont] <= 1.0 + 1e-9)).all())
# per-variable path (what EPBP.belief_rv / map do for one rv): 64 points of one variable per launch
vs = np.flatnonzero(flat.var_hidden & flat.var_cont)[:200]
t0 = time.perf_counter()
for v in vs:
    bp._belief_rv_points(int(v), x[v].cpu().numpy())
t_one = (time.perf_counter() - t0) / len(vs)
out = {'edges': flat.E, 'variables': flat.V, 'hidden_variables': hidden, 'points_per_variable': n,
       'belief_rv_all_ms': round(1e3 * t_all, 2), 'log_belief_points_per_s': round(hidden * n / t_all),
       'map_all_5_steps_ms': round(1e3 * t_map, 1), 'maps_per_s': round(hidden / t_map),
       'belief_all_8_points_ms': round(1e3 * t_bel, 1), 'probability_all_ms': round(1e3 * t_prob, 1),
       'per_variable_query_ms': round(1e3 * t_one, 3), 'per_variable_path_for_all_s': round(t_one * hidden, 1),
       'speedup_vs_per_variable': round(t_one * hidden / t_all)}
print(json.dumps(out))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], 'w'), indent=1)
