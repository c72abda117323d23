#!/bin/bash
# tiny-grid VI kernel: parity tests, scaled cfg 3 with and without it, kernel trace; then the host profile of the 10 M-edge c2f run
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
python -m pytest tests/test_gpu_vi.py -x -q -m gpu > $O/r4_tiny_tests.log 2>&1 || { tail -30 $O/r4_tiny_tests.log; exit 1; }
tail -2 $O/r4_tiny_tests.log
VI_TINY=1 python scripts/bench_configs.py vi_scaled > $O/r4_tiny_on.jsonl 2> $O/r4_tiny_on.log
VI_TINY=0 python scripts/bench_configs.py vi_scaled > $O/r4_tiny_off.jsonl 2> $O/r4_tiny_off.log
cut -c1-330 $O/r4_tiny_on.jsonl $O/r4_tiny_off.jsonl
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4_tiny_prof -o s -- python3 $R/scripts/bench_configs.py vi_scaled > /dev/null 2> $O/r4_tiny_prof.log
head -8 $O/r4_tiny_prof/s_kernel_stats.csv | cut -c1-160
cd $R
python scripts/diag/r4_prof_c2f.py > $O/r4_prof_c2f.log 2>&1
head -60 $O/r4_prof_c2f.log | cut -c1-180
