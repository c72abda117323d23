#!/bin/bash
# Collect the judged profiles of bench.py on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats, then separate --pmc passes (never combined with other trace domains), then the summaries.
# usage: scripts/profile_round.sh <tag>      -> gpurun_out/<tag>_*  (+ profiles/<tag>_* written by summarize_pmc.py)
set -e
tag=${1:-r01_final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_stats -o s -- python3 $R/bench.py --no-cpu-baseline > $O/${tag}_bench_under_rocprof.json 2> $O/${tag}_stats.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${tag}_fetch -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${tag}_write -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/${tag}_sq1 -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/${tag}_sq2 -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
cd $R
tail -1 $O/${tag}_bench_under_rocprof.json | cut -c1-200
