#!/bin/bash
# tuning aid: the 8-rank rehearsal's kernel times with the heavy kernel's tail zone at 0 / 2 / 8 entries per wave (built on the box)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/lifted-hybrid-variational-inference_amd/csrc
for W in 0 2 8; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DLHVI_PBP_TAIL_PER_WAVE=$W -c $C/pbp.hip -o /tmp/pbp_$W.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/liblhvi_t$W.so $C/abi.o $C/color.o $C/gabp.o $C/halo.o $C/vi.o /tmp/pbp_$W.o
  echo "=== tail zone $W entries per wave"
  export LHVI_LIB=/tmp/liblhvi_t$W.so
  $R/scripts/profile_sim_sharded.sh r05_tail$W 8 ownercompute | grep -i "heavy" | cut -c1-170
  tail -1 $R/gpurun_out/r05_tail$W.log
  unset LHVI_LIB
done
