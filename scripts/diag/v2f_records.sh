#!/bin/bash
# tuning aid: the wide v -> f kernel reading its variables as ids (the graph walked: four dependent loads before the rows) or as
# records (LHVI_PBP_V2F_RECORDS: one scalar load, then the rows); headline graph, kernel trace of bench.py
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for mode in 1 0; do
  LHVI_PBP_V2F_REC=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $O/v2frec_$mode -o s -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 2 > $O/v2frec_$mode.json 2>/dev/null
  echo "=== records=$mode"
  python3 - $O/v2frec_$mode/s_kernel_stats.csv $O/v2frec_$mode.json <<'PY'
import csv, json, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'lhvi::pbp_' in r['Name'] and int(r['Calls']) >= 10:
        print('  %-60s %8.3f ms' % (r['Name'].split('(')[0][:60], float(r['AverageNs']) / 1e6))
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print('  sweep', round(d['ms_per_step'], 3), 'ms')
PY
done
