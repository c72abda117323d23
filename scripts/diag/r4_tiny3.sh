#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
python -m pytest tests/test_gpu_vi.py tests/test_gpu_pbp.py -x -q -m gpu > $O/r4_tiny_tests.log 2>&1 || { tail -30 $O/r4_tiny_tests.log; exit 1; }
tail -2 $O/r4_tiny_tests.log
for v in "" scripts/ubench/liblhvi_s0w2.so scripts/ubench/liblhvi_s1w3.so; do
  echo "lib=$v"
  if [ -n "$v" ]; then export LHVI_LIB=$R/$v; fi
  python scripts/bench_configs.py vi_scaled 2> $O/r4_tiny_v.log | cut -c1-200 | head -1
done
unset LHVI_LIB
python scripts/bench_configs.py vi_models 2>> $O/r4_tiny_v.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   ', d['config'][:40], d.get('s_per_update_device'))"
