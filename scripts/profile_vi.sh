#!/bin/bash
# rocprofv3 summaries of the variational step: kernel stats of the reference's two HMLN models (VI / LVI / C2FVI, 100 updates) and of
# the scaled ground graphs (general kernel on the scaled cfg 3, Gaussian fast path on the RGM); HBM counters in separate passes.
set -e
tag=${1:-r04_vi}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_scaled_stats -o s -- python3 $R/scripts/bench_configs.py vi_scaled > $O/${tag}_scaled_under_rocprof.jsonl 2> $O/${tag}_scaled.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_models_stats -o s -- python3 $R/scripts/bench_configs.py vi_models > $O/${tag}_models_under_rocprof.jsonl 2> $O/${tag}_models.log
if [ -z "$VI_NO_PMC" ]; then
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${tag}_fetch -- python3 $R/scripts/bench_configs.py vi_scaled > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${tag}_write -- python3 $R/scripts/bench_configs.py vi_scaled > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d $O/${tag}_sq -- python3 $R/scripts/bench_configs.py vi_scaled > /dev/null 2>&1
fi
cd $R
python3 scripts/bench_configs.py vi_models vi_scaled > $O/${tag}_configs.jsonl 2> $O/${tag}_plain.log
cut -c1-200 $O/${tag}_configs.jsonl
