#!/bin/bash
# rocprofv3 summaries of the secondary configurations (Gaussian sweep at 10M edges, colour refinement + lifted VI of cfg 5,
# ground variational step): kernel stats and, in separate passes, HBM read / write counters.  Run through gpurun.
set -e
tag=${1:-r01_secondary}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_stats -o s -- python3 $R/scripts/bench_configs.py gauss cfg5 vi_ground > $O/${tag}_configs.jsonl 2> $O/${tag}.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${tag}_fetch -- python3 $R/scripts/bench_configs.py gauss vi_ground > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${tag}_write -- python3 $R/scripts/bench_configs.py gauss vi_ground > /dev/null 2>&1
cd $R
cat $O/${tag}_configs.jsonl | cut -c1-160
