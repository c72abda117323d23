set -e
python -m pytest tests/test_gpu_gabp.py -x -q -m gpu > gpurun_out/r4_t11.log 2>&1 || { tail -40 gpurun_out/r4_t11.log; exit 1; }
tail -2 gpurun_out/r4_t11.log
KALMAN_T=12000 python scripts/bench_configs.py gauss gauss_rel cfg2 > gpurun_out/r4_gauss.jsonl 2> gpurun_out/r4_gauss.err || { tail -20 gpurun_out/r4_gauss.err; exit 1; }
python - <<PY
import json
for l in open("gpurun_out/r4_gauss.jsonl"):
    d=json.loads(l); print(d["config"][:90], {k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k in ("sweep_ms","hbm_frac","ms_20_sweeps","ms_20_sweeps_and_marginals_device")})
PY
