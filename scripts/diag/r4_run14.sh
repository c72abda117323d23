set -e
python -m pytest tests/test_gpu_pbp.py tests/test_gpu_edge_cases.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r4_t14.log 2>&1 || { tail -40 gpurun_out/r4_t14.log; exit 1; }
tail -2 gpurun_out/r4_t14.log
python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
