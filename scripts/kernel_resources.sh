#!/bin/bash
# VGPR / SGPR / scratch / occupancy of every kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage), one line each
# usage: scripts/kernel_resources.sh lifted-hybrid-variational-inference_amd/csrc/pbp.hip [extra -D flags]
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function "$@" -c "$f" -o /dev/null \
    -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys
cur = {}
for line in sys.stdin:
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m: continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        if cur: print(cur)
        cur = {"fn": re.sub(r"^_ZN4lhvi\d+", "", t.split(": ")[1])[:44]}
    else:
        k, _, v = t.partition(": ")
        if k.strip() in ("VGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"):
            cur[k.strip().split(" ")[0]] = v
if cur: print(cur)
'
