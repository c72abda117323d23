set -e
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r4_b7.json 2> gpurun_out/r4_b7.err || { tail -20 gpurun_out/r4_b7.err; exit 1; }
python -c "
import json; d=json.loads([l for l in open('gpurun_out/r4_b7.json') if l.startswith('{')][0]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
python scripts/sim_sharded.py 10000000 8 ownercompute > gpurun_out/r4_sim_oc_8.log 2>&1 || { tail -20 gpurun_out/r4_sim_oc_8.log; exit 1; }
tail -2 gpurun_out/r4_sim_oc_8.log
python scripts/sim_sharded.py 10000000 4 ownercompute > gpurun_out/r4_sim_oc_4.log 2>&1
tail -1 gpurun_out/r4_sim_oc_4.log
python scripts/sim_sharded.py 10000000 2 ownercompute > gpurun_out/r4_sim_oc_2.log 2>&1
tail -1 gpurun_out/r4_sim_oc_2.log
