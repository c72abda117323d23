set -e
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r4_t3.log 2>&1 || { tail -60 gpurun_out/r4_t3.log; exit 1; }
tail -3 gpurun_out/r4_t3.log
python scripts/bench_configs.py c2f_pbp > gpurun_out/r4_c2f.jsonl 2> gpurun_out/r4_c2f.err || { tail -30 gpurun_out/r4_c2f.err; exit 1; }
cut -c1-700 gpurun_out/r4_c2f.jsonl
