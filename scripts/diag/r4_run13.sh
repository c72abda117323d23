set -e
R=$GRAFT_REPO_ROOT
for v in base scripts/ubench/liblhvi_proptab.so; do
  if [ "$v" != base ]; then export LHVI_LIB=$R/$v; fi
  tag=$(basename $v .so)
  for prop in simple EP; do
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_prop_${tag}_$prop -o s -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --proposal $prop > $R/gpurun_out/r04_prop_${tag}_$prop.json 2>/dev/null
    cd $R
    python - $tag $prop <<'PY'
import csv, json, sys
tag, prop = sys.argv[1:3]
d = json.loads([l for l in open("gpurun_out/r04_prop_%s_%s.json" % (tag, prop)) if l.startswith("{")][0])
rows = [r for r in csv.DictReader(open("gpurun_out/r04_prop_%s_%s/s_kernel_stats.csv" % (tag, prop))) if "proposal" in r["Name"]]
print(tag, prop, "sweeps/s %.2f" % d["value"], [(r["Name"][:44], round(float(r["AverageNs"]) / 1e3, 1)) for r in rows])
PY
  done
done
