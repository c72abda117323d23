set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
python tests/soak/soak_pbp_random.py 10000 1500
SOAK_HUBS=1 python tests/soak/soak_pbp_random.py 20000 500
python tests/soak/soak_c2f_random.py 10000 1500 mixed
SOAK_HUBS=1 python tests/soak/soak_c2f_random.py 20000 400
python tests/soak/soak_gabp_random.py 10000 1500
python tests/soak/soak_vi_random.py 10000 2000
python tests/soak/soak_lvi_random.py 10000 2000
python tests/soak/soak_c2fvi_random.py 10000 1000
python tests/soak/soak_dist_random.py 10000 600
python tests/soak/soak_twins_random.py 10000 600
