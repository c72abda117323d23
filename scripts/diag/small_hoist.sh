#!/bin/bash
# tuning aid: the few-particle f->v kernel (pbp_f2v_small_kernel) with its loads as written in round 4 (0), with the descriptor
# and everything it points to fetched in two round trips (1), and with the next step's descriptor touched one step ahead (2);
# each with and without the padding between the lane groups' record blocks (LHVI_SMALL_PAD);
# built on the GPU box, timed through bench.py --particles n and the kernel trace
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/lifted-hybrid-variational-inference_amd/csrc
mkdir -p $R/gpurun_out
for HP in "0 0" "2 0" "1 1" "2 1"; do
  set -- $HP; H=$1$2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DLHVI_SMALL_HOIST=$1 -DLHVI_SMALL_PAD=$2 -c $C/pbp.hip -o /tmp/pbp_h$H.o 2>/dev/null
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/liblhvi_h$H.so $C/abi.o $C/color.o $C/gabp.o $C/halo.o $C/vi.o /tmp/pbp_h$H.o
  echo "=== LHVI_SMALL_HOIST=$1 LHVI_SMALL_PAD=$2"
  for n in 10 16 20; do
    LHVI_LIB=/tmp/liblhvi_h$H.so python3 $R/bench.py --particles $n --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print($n, round(d['ms_per_step'],3), 'ms per sweep; f2v kernel', round(d['roofline']['kernel_ms'],3), 'ms')"
  done
done
