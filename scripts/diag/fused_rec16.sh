#!/bin/bash
# the fused per-variable kernel with sixteen-word records (np, var_ptr and the first six edges in the record) against eight-word
# records (LHVI_PBP_FUSED_REC16=0): its tests, then ms per sweep on the headline graph at n = 10, 16, 20
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
timeout -k 10 600 python3 -m pytest $R/tests/test_gpu_pbp.py -q -m gpu -k "fused or few_particle or smoke or golden" > $O/fused_tests.log 2>&1
tail -3 $O/fused_tests.log
: > $O/fused_rec16.log
for n in 10 16 20; do
  for r in 1 0 1 0; do
    echo "n=$n rec16=$r" >> $O/fused_rec16.log
    LHVI_PBP_FUSED_REC16=$r timeout -k 10 300 python3 $R/bench.py --particles $n --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3),'ms')" >> $O/fused_rec16.log || exit 1
  done
done
cat $O/fused_rec16.log
