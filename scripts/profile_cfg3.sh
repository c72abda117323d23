#!/bin/bash
# rocprofv3 summaries of the scaled cfg 3 (paper-popularity HMLN copies, conditional-quadratic routing): kernel stats and,
# in separate passes, the HBM read / write counters.  Run through gpurun from the repo root.
set -e
tag=${1:-r03_cfg3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export CFG3_ROUTED_ONLY=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_stats -o s -- python3 $R/scripts/bench_configs.py cfg3s > $O/${tag}_configs.jsonl 2> $O/${tag}.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${tag}_fetch -- python3 $R/scripts/bench_configs.py cfg3s > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${tag}_write -- python3 $R/scripts/bench_configs.py cfg3s > /dev/null 2>&1
cd $R
cat $O/${tag}_configs.jsonl | cut -c1-400
