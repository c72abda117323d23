set -e
python -m pytest tests -x -q -m gpu > gpurun_out/r04_gpu_tests.log 2>&1 || { tail -40 gpurun_out/r04_gpu_tests.log; exit 1; }
tail -2 gpurun_out/r04_gpu_tests.log
python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err || { tail -20 gpurun_out/r04_bench_default.err; exit 1; }
cut -c1-300 gpurun_out/r04_bench_default.json
bash scripts/profile_round.sh r04_final
