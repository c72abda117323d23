#!/bin/bash
# tuning aid: the interpreter-free tiny VI kernel with every component's means read inside the point loop (LHVI_VI_TINY_HOIST=0) or
# held in registers (1), at 2 / 3 waves per SIMD; built on the GPU box, scaled cfg 3 (970 k ground MLN factors)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/lifted-hybrid-variational-inference_amd/csrc
for HW in "0 3" "1 3" "1 2"; do
  set -- $HW
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DLHVI_VI_TINY_HOIST=$1 -DLHVI_VI_TINY_SLIM_WAVES=$2 -c $C/vi.hip -o /tmp/vi_$1$2.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/liblhvi_v$1$2.so $C/abi.o $C/color.o $C/gabp.o $C/halo.o $C/pbp.o /tmp/vi_$1$2.o
  echo "=== hoist $1, $2 waves/SIMD"
  LHVI_LIB=/tmp/liblhvi_v$1$2.so python3 $R/scripts/bench_configs.py vi_scaled 2>/dev/null | cut -c1-420
done
