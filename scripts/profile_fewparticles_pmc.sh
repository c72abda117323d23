#!/bin/bash
# PMC passes of bench.py --particles <n> (separate passes, never combined with a trace domain) and their summary:
#   scripts/profile_fewparticles_pmc.sh <tag> <n>      -> profiles/<tag>_traffic.json, profiles/<tag>_pmc.md
set -e
tag=${1:-r05_n16}
n=${2:-16}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
A="--particles $n --no-cpu-baseline --steps 3 --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_stats -o s -- python3 $R/bench.py --particles $n --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${tag}_fetch -- python3 $R/bench.py $A > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${tag}_write -- python3 $R/bench.py $A > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/${tag}_sq1 -- python3 $R/bench.py $A > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/${tag}_sq2 -- python3 $R/bench.py $A > /dev/null 2>&1
cd $R
python3 scripts/summarize_pmc.py $tag $O/${tag}_stats $O/${tag}_fetch $O/${tag}_write $O/${tag}_sq1 $O/${tag}_sq2
cp profiles/${tag}_pmc.md profiles/${tag}_traffic.json profiles/${tag}_kernel_stats.csv $O/ 2>/dev/null || true
cat profiles/${tag}_pmc.md
