#!/bin/bash
# tuning aid: the f -> v half sweep as one call on one stream (LHVI_PBP_OVERLAP=0) or with its short kernels (pair / light / cq /
# generic) on a second stream beside the long one, which then leaves a workgroup per CU free (1); bench.py at n = 64 / 16 / 10, and a
# check that both give the same bits
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
for O in 0 1; do
  echo "=== LHVI_PBP_OVERLAP=$O"
  for n in 64 16 10; do
    LHVI_PBP_OVERLAP=$O python3 $R/bench.py --particles $n --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print($n, round(d['ms_per_step'],3), 'ms per sweep; long f2v kernel', round(d['roofline']['kernel_ms'],3), 'ms')"
  done
done
python3 - <<'PY'
import os, sys
R = os.environ.get('GRAFT_REPO_ROOT', os.getcwd())
sys.path[:0] = [R, os.path.join(R, 'lifted-hybrid-variational-inference_amd')]
import torch
from lhvi import synth
from lhvi.pbp import EPBP
flat = synth.hybrid_mrf_flat(V=40000, deg=4, seed=3)
outs = []
for ov in ('0', '1'):
    os.environ['LHVI_PBP_OVERLAP'] = ov
    for n in (64, 12):
        bp = EPBP(None, n=n, proposal_approximation='simple', sampler='device', seed=5)
        bp._setup(None, flat=flat)
        bp._run_sweeps(4)
        torch.cuda.synchronize()
        outs.append((ov, n, bp.f2v.clone(), bp.q_dev.clone()))
for n in (64, 12):
    a = [o for o in outs if o[1] == n]
    print('n', n, 'same bits with and without the second stream:', bool(torch.equal(a[0][2], a[1][2]) and torch.equal(a[0][3], a[1][3])))
PY
