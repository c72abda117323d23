#!/bin/bash
# the headline workload at the particle counts the reference's demos use (n = 10, 16, 20; Demo/RGM/demo.py:19-20,
# RGMKLDivergence.py:54) and with the reference's default proposal rule (EP): one bench.py line each, plus kernel stats of the n = 16
# and the EP run.  Run through gpurun from the repo root.
set -e
tag=${1:-r04_particles}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
: > $O/${tag}.jsonl
for n in 10 16 20 64; do
  python3 $R/bench.py --particles $n --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | grep '^{' >> $O/${tag}.jsonl
done
python3 $R/bench.py --proposal EP --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | grep '^{' >> $O/${tag}.jsonl
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_n16_stats -o s -- python3 $R/bench.py --particles 16 --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_ep_stats -o s -- python3 $R/bench.py --proposal EP --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2>&1
cd $R
python3 - $O/${tag}.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print(d['config']['particles'], d['config']['proposal'], round(d['value'], 2), 'sweeps/s', round(d['ms_per_step'], 3), 'ms')
PY
