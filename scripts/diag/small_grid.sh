#!/bin/bash
# tuning aid: the few-particle f->v kernel with its integral points computed directly (LHVI_SMALL_GRID=0) or by the recurrence along
# the uniform grid inside the lane group (1), at several occupancy targets; built on the GPU box, timed through bench.py --particles n
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/lifted-hybrid-variational-inference_amd/csrc
mkdir -p $R/gpurun_out
for GW in "0 7" "1 7" "1 6" "1 5"; do
  set -- $GW; H=$1$2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DLHVI_SMALL_GRID=$1 -DLHVI_SMALL_WAVES=$2 -c $C/pbp.hip -o /tmp/pbp_g$H.o 2>/dev/null
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/liblhvi_g$H.so $C/abi.o $C/color.o $C/gabp.o $C/halo.o $C/vi.o /tmp/pbp_g$H.o
  echo "=== LHVI_SMALL_GRID=$1 LHVI_SMALL_WAVES=$2"
  for n in 10 16 20 32; do
    LHVI_LIB=/tmp/liblhvi_g$H.so python3 $R/bench.py --particles $n --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print($n, round(d['ms_per_step'],3), 'ms per sweep; f2v kernel', round(d['roofline']['kernel_ms'],3), 'ms')"
  done
done
