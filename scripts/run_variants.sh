#!/bin/bash
# tuning aid: time the sweep's kernels with several builds of liblhvi.so (build_variants/liblhvi_<name>.so, selected
# through LHVI_LIB) on one box, the default build first and last.  usage: bash scripts/run_variants.sh name1 name2 ...
set -e
for v in default "$@" default; do
  if [ $v = default ]; then unset LHVI_LIB; else export LHVI_LIB=$PWD/build_variants/liblhvi_$v.so; fi
  echo "=== $v"; python scripts/time_kernels.py 10000000 2>&1 | grep -E "f2v heavy|f2v light|sum|v2f |proposal|resample"
done
