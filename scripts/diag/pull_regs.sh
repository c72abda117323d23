#!/bin/bash
# Gaussian pull kernel with the slot records kept in registers and the gathers batched: tests, then the three 10 M-edge-class
# graphs with the kernel capped at 6 (default build), 4 and 8 waves per SIMD
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
timeout -k 10 900 python3 -m pytest $R/tests/test_gpu_gabp.py $R/tests/test_rkf.py -q -m gpu -x > $O/pull_tests.log 2>&1 || { tail -30 $O/pull_tests.log; exit 1; }
tail -2 $O/pull_tests.log
timeout -k 10 600 python3 -m pytest $R/tests/test_gpu_pbp.py -q -m gpu -k "few_particle or small_particle" > $O/narrow_tests.log 2>&1
tail -2 $O/narrow_tests.log
export KALMAN_T=12000
: > $O/pull_regs.log
for lib in default pull4 pull8 default; do
  if [ $lib = default ]; then unset LHVI_LIB; else export LHVI_LIB=$R/variants_tmp/liblhvi_$lib.so; fi
  echo "lib=$lib" >> $O/pull_regs.log
  timeout -k 10 300 python3 $R/scripts/bench_configs.py gauss gauss_rel 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l)
    if 'sweep_ms' in d: print(round(d['sweep_ms'],4), round(d['hbm_frac'],3), d['config'][-60:])" >> $O/pull_regs.log || exit 1
done
cat $O/pull_regs.log
