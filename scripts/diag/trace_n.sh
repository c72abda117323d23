#!/bin/bash
# kernel trace of the headline graph at a few particle counts: per-kernel average ms (diagnosis aid).  usage: trace_n.sh 10 20
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_n$n -o s -- python3 $R/bench.py --particles $n --no-cpu-baseline --steps 10 --warmup 2 > $O/trace_n$n.json 2>/dev/null
  f=$(ls $O/trace_n$n/*/s_kernel_stats.csv $O/trace_n$n/s_kernel_stats.csv 2>/dev/null | head -1)
  echo "== n=$n"; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:12]:
    print('%-60s calls %4s avg %.3f ms total %.2f ms' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e6, float(r['TotalDurationNs']) / 1e6))
PY
done
