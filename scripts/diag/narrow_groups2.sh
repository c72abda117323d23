#!/bin/bash
# second pass: the 10-lane groups at 5 waves / SIMD (default build) against 6 waves with 48 B of scratch (variants_tmp/liblhvi_w10six.so),
# and the Kalman-filter Gaussian sweep with 360 / 2 distinct potentials (does the potential table bound it?)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
timeout -k 10 600 python3 -m pytest $R/tests/test_gpu_pbp.py -q -m gpu -k "few_particle or small_particle" > $O/narrow_tests.log 2>&1
tail -3 $O/narrow_tests.log
: > $O/narrow_groups2.log
for lib in default w10six default w10six; do
  if [ $lib = default ]; then unset LHVI_LIB; else export LHVI_LIB=$R/variants_tmp/liblhvi_$lib.so; fi
  echo "n=10 lib=$lib" >> $O/narrow_groups2.log
  timeout -k 10 300 python3 $R/bench.py --particles 10 --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3),'ms; f2v kernel', round(d['roofline']['kernel_ms'],3))" >> $O/narrow_groups2.log || exit 1
done
unset LHVI_LIB
cat $O/narrow_groups2.log
export KALMAN_T=12000 GAUSS_REL_ONLY=kalman
timeout -k 10 300 python3 $R/scripts/bench_configs.py gauss_rel 2>/dev/null | cut -c1-300
KALMAN_CONST_A=1 timeout -k 10 300 python3 $R/scripts/bench_configs.py gauss_rel 2>/dev/null | cut -c1-300
