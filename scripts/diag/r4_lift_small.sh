#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
python -m pytest tests -x -q -m gpu > $O/r4_ls_tests.log 2>&1 || { tail -30 $O/r4_ls_tests.log; exit 1; }
tail -2 $O/r4_ls_tests.log
C2F_SMALL_ONLY=1 python scripts/bench_configs.py c2f_pbp 2> $O/r4_ls.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'][36:90], d['wall_s'], d['per_sweep_ms_median'])"
python scripts/bench_configs.py vi_models 2>> $O/r4_ls.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   ', d['config'][:50], d.get('s_per_update_device'), d.get('s_per_update_end_to_end'), d.get('relift_ms_per_round', '')[:4])"
