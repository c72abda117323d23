#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
python -m pytest tests/test_gpu_vi.py -x -q -m gpu > $O/r4_tiny_tests.log 2>&1 || { tail -30 $O/r4_tiny_tests.log; exit 1; }
tail -2 $O/r4_tiny_tests.log
for t in 1 0; do
VI_TINY=$t python scripts/bench_configs.py vi_scaled 2> $O/r4_tiny_v.log | cut -c1-200 | head -1
done
python scripts/bench_configs.py vi_models 2>> $O/r4_tiny_v.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   ', d['config'][:40], d.get('s_per_update_device'), d.get('total_s'))"
