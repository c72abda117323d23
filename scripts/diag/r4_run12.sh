set -e
python -m pytest tests/test_gpu_pbp.py tests/test_gpu_edge_cases.py -x -q -m gpu > gpurun_out/r4_t12.log 2>&1 || { tail -50 gpurun_out/r4_t12.log; exit 1; }
tail -2 gpurun_out/r4_t12.log
python scripts/bench_configs.py demo_loop > gpurun_out/r4_demo_loop.jsonl 2> gpurun_out/r4_demo_loop.err || { tail -20 gpurun_out/r4_demo_loop.err; exit 1; }
cat gpurun_out/r4_demo_loop.jsonl
