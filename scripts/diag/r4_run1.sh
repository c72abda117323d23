set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_pbp.py -x -q -m gpu -k "coarse or first_device_draw or sampler" > gpurun_out/r4_t1.log 2>&1 || { tail -40 gpurun_out/r4_t1.log; exit 1; }
tail -3 gpurun_out/r4_t1.log
C2F_SMALL_ONLY=1 python scripts/bench_configs.py c2f_pbp > gpurun_out/r4_c2f_small.jsonl 2> gpurun_out/r4_c2f_small.err || { tail -30 gpurun_out/r4_c2f_small.err; exit 1; }
cat gpurun_out/r4_c2f_small.jsonl
