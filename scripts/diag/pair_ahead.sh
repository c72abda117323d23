#!/bin/bash
# tuning aid: the pair f->v kernel with its descriptors one or two entries ahead (LHVI_PAIR_AHEAD); built on the GPU box, kernel trace
# of bench.py at n = 64 and n = 16
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/lifted-hybrid-variational-inference_amd/csrc
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for A in 1 2; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DLHVI_PAIR_AHEAD=$A -c $C/pbp.hip -o /tmp/pbp_a$A.o 2>/dev/null
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/liblhvi_a$A.so $C/abi.o $C/color.o $C/gabp.o $C/halo.o $C/vi.o /tmp/pbp_a$A.o
  for n in 64 16; do
    LHVI_LIB=/tmp/liblhvi_a$A.so rocprofv3 --kernel-trace --stats --output-format csv -d $O/pair_a${A}_$n -o s -- python3 $R/bench.py --particles $n --no-cpu-baseline --steps 10 --warmup 2 > $O/pair_a${A}_$n.json 2>/dev/null
    python3 - $O/pair_a${A}_$n/s_kernel_stats.csv $O/pair_a${A}_$n.json $A $n <<'PY'
import csv, json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
for r in csv.DictReader(open(sys.argv[1])):
    if 'pair_kernel' in r['Name']:
        print('ahead %s n %s: pair kernel %.3f ms, sweep %.3f ms' % (sys.argv[3], sys.argv[4], float(r['AverageNs']) / 1e6, d['ms_per_step']))
PY
  done
done
