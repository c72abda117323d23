#!/bin/bash
# every random-instance soak of round 4 with the seed ranges recorded in profiles/r04_experiments.md (one MI355X, ~5 minutes).
# Run through gpurun from the repo root:  gpurun --timeout 1200 -- 'bash tests/soak/run_soaks.sh > gpurun_out/soaks.log 2>&1'
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python tests/soak/soak_pbp_random.py 0 200
python tests/soak/soak_pbp_random.py 200 400
SOAK_HUBS=1 python tests/soak/soak_pbp_random.py 6000 300
python tests/soak/soak_c2f_random.py 0 150
python tests/soak/soak_c2f_random.py 1000 150
python tests/soak/soak_c2f_random.py 2000 200
python tests/soak/soak_c2f_random.py 3000 300 mixed
SOAK_HUBS=1 python tests/soak/soak_c2f_random.py 7000 120
python tests/soak/soak_gabp_random.py 0 500
SOAK_HUBS=1 python tests/soak/soak_gabp_random.py 8000 150
python tests/soak/soak_vi_random.py 1000 400
python tests/soak/soak_lvi_random.py 0 500
python tests/soak/soak_c2fvi_random.py 0 300
python tests/soak/soak_dist_random.py 0 150
python tests/soak/soak_twins_random.py 0 120
python tests/soak/soak_queries_random.py 0 360
python scripts/soak.py 10000000 200
