#!/bin/bash
# f->v few-particle kernel with narrow lane groups (10 / 12 lanes; two particles per lane for 17-32 particles) against the 16- / 32-lane
# groups (LHVI_PBP_POW2_GROUPS=1): the tests of both, then ms per sweep on the headline graph.  Run through gpurun from the repo root.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
timeout -k 10 600 python3 -m pytest $R/tests/test_gpu_pbp.py -q -m gpu -k "few_particle or small_particle or fused_variable or smoke" > $O/narrow_tests.log 2>&1
tail -5 $O/narrow_tests.log
[ -n "$1" ] && timeout -k 10 300 python3 $R/scripts/diag/narrow_diag.py > $O/narrow_diag.log 2>&1
: > $O/narrow_groups.log
for n in ${NS:-10 12 20 24 32}; do
  for pow2 in 0 1; do
    echo "n=$n pow2=$pow2" >> $O/narrow_groups.log
    LHVI_PBP_POW2_GROUPS=$pow2 timeout -k 10 300 python3 $R/bench.py --particles $n --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3),'ms; f2v kernel', round(d['roofline']['kernel_ms'],3))" >> $O/narrow_groups.log || exit 1
  done
done
cat $O/narrow_groups.log
