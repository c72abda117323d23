#!/bin/bash
# tuning aid: the interpreter-free tiny VI kernel at 2 / 3 / 4 waves per SIMD (LHVI_VI_TINY_SLIM_WAVES), built on the GPU box
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/lifted-hybrid-variational-inference_amd/csrc
for W in 2 3 4; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DLHVI_VI_TINY_SLIM_WAVES=$W -c $C/vi.hip -o /tmp/vi_$W.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/liblhvi_w$W.so $C/abi.o $C/color.o $C/gabp.o $C/halo.o $C/pbp.o /tmp/vi_$W.o
  echo "=== slim tiny kernel at $W waves/SIMD"
  LHVI_LIB=/tmp/liblhvi_w$W.so python3 $R/scripts/bench_configs.py vi_scaled 2>/dev/null | cut -c1-400
done
