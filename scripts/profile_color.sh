#!/bin/bash
# rocprofv3 summaries of colour refinement + lifted VI on the 10 M-edge cfg-5 graph: kernel stats, then HBM counters in
# separate passes.  Run through gpurun from the repo root.
set -e
tag=${1:-r03_color}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_stats -o s -- python3 $R/scripts/bench_configs.py cfg5 > $O/${tag}_configs.jsonl 2> $O/${tag}.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${tag}_fetch -- python3 $R/scripts/bench_configs.py cfg5 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${tag}_write -- python3 $R/scripts/bench_configs.py cfg5 > /dev/null 2>&1
cd $R
cat $O/${tag}_configs.jsonl | cut -c1-400
