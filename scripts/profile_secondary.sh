#!/bin/bash
# rocprofv3 summaries of the secondary configurations (Gaussian sweep at 10M edges on the random expander, the 10 M-edge RGM and a
# Kalman graph; colour refinement + lifted VI of cfg 5; ground variational step; lifted particle sweep): kernel stats and, in
# separate passes, HBM read / write counters.  Run through gpurun.
set -e
tag=${1:-r04_secondary}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export KALMAN_T=${KALMAN_T:-12000}
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_stats -o s -- python3 $R/scripts/bench_configs.py gauss gauss_rel cfg2 cfg5 vi_ground lifted_pbp c2fvi > $O/${tag}_configs_under_rocprof.jsonl 2> $O/${tag}.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${tag}_fetch -- python3 $R/scripts/bench_configs.py gauss gauss_rel cfg5 vi_ground > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${tag}_write -- python3 $R/scripts/bench_configs.py gauss gauss_rel cfg5 vi_ground > /dev/null 2>&1
cd $R
python3 scripts/bench_configs.py gauss gauss_rel cfg2 cfg3 cfg5 vi_ground lifted_pbp c2fvi > $O/${tag}_configs.jsonl 2> $O/${tag}_plain.log
cat $O/${tag}_configs.jsonl | cut -c1-200
