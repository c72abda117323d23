#!/bin/bash
# kernel trace of the one-GPU rehearsal of the sharded sweep (scripts/sim_sharded.py): per-kernel device time of every rank's phases,
# without the host's launch latencies that the rehearsal's wall-clock phase times include.  Run through gpurun from the repo root.
# usage: scripts/profile_sim_sharded.sh <tag> [world] [mode]   -> gpurun_out/<tag>_kernel_stats.csv, gpurun_out/<tag>.log
set -e
tag=${1:-r05_sim_oc8}
world=${2:-8}
mode=${3:-ownercompute}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_stats -o s -- python3 $R/scripts/sim_sharded.py 10000000 $world $mode > $O/${tag}.log 2>&1
cd $R
cp $(ls $O/${tag}_stats/*/s_kernel_stats.csv $O/${tag}_stats/s_kernel_stats.csv 2>/dev/null | head -1) $O/${tag}_kernel_stats.csv
head -14 $O/${tag}_kernel_stats.csv | cut -c1-160
