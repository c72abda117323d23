#!/bin/bash
# pair list in lane groups of 10 / 12 / 20 lanes against 16 / 32 (LHVI_PBP_POW2_GROUPS=1 also widens the f->v few-particle kernel: compare
# the pair kernel's line of the trace, not the sweep): tests, then kernel trace lines at n = 10, 12, 20
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
timeout -k 10 600 python3 -m pytest $R/tests/test_gpu_pbp.py -q -m gpu -k "packed_pair or smoke or golden or fused" > $O/pair_tests.log 2>&1; tail -3 $O/pair_tests.log
bash $R/scripts/diag/trace_n.sh 10 12 20 2>&1 | grep -E "== n|pair_small"
for n in 10 12 20; do python3 $R/bench.py --particles $n --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('n', d['config']['particles'], round(d['ms_per_step'],3),'ms')"; done
