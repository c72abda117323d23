#!/bin/bash
# fresh seed ranges for the kernels the second session of round 5 changed (few-particle f->v lane groups, fused-kernel records,
# Gaussian pull kernel): gpurun --timeout 1200 -- 'bash tests/soak/run_soaks_fresh2.sh > gpurun_out/soaks_fresh2.log 2>&1'
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
python tests/soak/soak_pbp_random.py 30000 800
SOAK_HUBS=1 python tests/soak/soak_pbp_random.py 36000 200
python tests/soak/soak_gabp_random.py 30000 800
SOAK_HUBS=1 python tests/soak/soak_gabp_random.py 38000 200
python tests/soak/soak_twins_random.py 30000 300
python tests/soak/soak_c2f_random.py 30000 300
python tests/soak/soak_dist_random.py 30000 150
