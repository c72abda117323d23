#!/bin/bash
# tuning aid: the fused per-variable kernel with the loads of LHVI_FUSED_NB proposal passes in flight together, at several occupancy
# targets (round 5 first measured the edge-by-edge walk of round 4 against the batched loads: 5.65 -> 5.05 ms at n = 10, 5.83 -> 5.18 at n = 16); n = 10 / 16 (16-lane form) and n = 20 / 32 with the 32-lane forms switched on
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/lifted-hybrid-variational-inference_amd/csrc
for BW in "4 4" "4 3" "2 4" "2 5"; do
  set -- $BW; H=$1$2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DLHVI_FUSED_NB=$1 -DLHVI_FUSED_WAVES=$2 -c $C/pbp.hip -o /tmp/pbp_b$H.o 2>/dev/null
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/liblhvi_b$H.so $C/abi.o $C/color.o $C/gabp.o $C/halo.o $C/vi.o /tmp/pbp_b$H.o
  echo "=== LHVI_FUSED_NB=$1 LHVI_FUSED_WAVES=$2"
  for n in 10 16; do
    LHVI_LIB=/tmp/liblhvi_b$H.so python3 $R/bench.py --particles $n --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print($n, round(d['ms_per_step'],3), 'ms per sweep')"
  done
  for n in 20 32; do
    LHVI_PBP_FUSED_MAX=32 LHVI_LIB=/tmp/liblhvi_b$H.so python3 $R/bench.py --particles $n --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print($n, round(d['ms_per_step'],3), 'ms per sweep (32-lane fused forms on)')"
  done
done
