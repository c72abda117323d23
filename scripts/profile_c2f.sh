#!/bin/bash
# rocprofv3 kernel trace of the particle coarse-to-fine runs (scripts/bench_configs.py c2f_pbp): the sum of all kernel durations is
# the DEVICE time of a run; the wall time minus that is host time.  Run through gpurun.
set -e
tag=${1:-r04_c2f}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in 0 1 2; do
  C2F_CASE=$c rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_case${c} -o s -- python3 $R/scripts/bench_configs.py c2f_pbp > $O/${tag}_case${c}_under_rocprof.jsonl 2> $O/${tag}_case${c}.log
done
cd $R
python3 scripts/bench_configs.py c2f_pbp > $O/${tag}_configs.jsonl 2> $O/${tag}_plain.log
cut -c1-300 $O/${tag}_configs.jsonl
