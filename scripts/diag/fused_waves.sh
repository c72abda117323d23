#!/bin/bash
# tuning aid: the fused per-variable kernel at several occupancy targets (built on the GPU box), n = 16 and n = 20
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/lifted-hybrid-variational-inference_amd/csrc
for W in 4 6 8; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DLHVI_FUSED_WAVES=$W -c $C/pbp.hip -o /tmp/pbp_f$W.o 2>/dev/null
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/liblhvi_f$W.so $C/abi.o $C/color.o $C/gabp.o $C/halo.o $C/vi.o /tmp/pbp_f$W.o
  echo "=== fused kernel at $W waves/SIMD"
  for n in 16 20; do
    LHVI_LIB=/tmp/liblhvi_f$W.so python3 $R/bench.py --particles $n --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print($n, round(d['ms_per_step'],3), 'ms')"
  done
done
