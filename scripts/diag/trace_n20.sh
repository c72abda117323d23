cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/n20 -o s -- python3 $GRAFT_REPO_ROOT/bench.py --particles 20 --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2>&1
python3 - $GRAFT_REPO_ROOT/gpurun_out/n20/s_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print(r['Name'][:75].ljust(75), r['Calls'], round(float(r['AverageNs'])/1e6,3))
PY
