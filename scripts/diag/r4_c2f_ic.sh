#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_pbp.py tests/test_gpu_gabp.py tests/test_rkf.py -x -q -m gpu > $O/r4_ic_tests.log 2>&1 || { tail -30 $O/r4_ic_tests.log; exit 1; }
tail -2 $O/r4_ic_tests.log
python scripts/bench_configs.py c2f_pbp 2> $O/r4_ic.log | cut -c1-420
python scripts/diag/r4_prof_c2f.py > $O/r4_prof_c2f.log 2>&1
head -40 $O/r4_prof_c2f.log | cut -c1-150
