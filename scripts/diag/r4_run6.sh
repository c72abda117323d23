set -e
python -m pytest tests/test_dist.py -x -q -m gpu > gpurun_out/r4_t6.log 2>&1 || { tail -40 gpurun_out/r4_t6.log; exit 1; }
tail -3 gpurun_out/r4_t6.log
python scripts/sim_sharded.py 10000000 8 ownercompute > gpurun_out/r4_sim_oc_8.log 2>&1 || { tail -20 gpurun_out/r4_sim_oc_8.log; exit 1; }
tail -2 gpurun_out/r4_sim_oc_8.log
LHVI_DIST_BACKEND=gloo python bench.py --gpus 2 --edges 400000 --steps 3 --warmup 1 --no-cpu-baseline --exchange ownercompute > gpurun_out/r4_b_oc2.json 2> gpurun_out/r4_b_oc2.err || { tail -20 gpurun_out/r4_b_oc2.err; exit 1; }
python -c "
import json; d=json.loads([l for l in open('gpurun_out/r4_b_oc2.json') if l.startswith('{')][0]); print(d['value'], d['config']['sharding'], d['phases_ms'])"
