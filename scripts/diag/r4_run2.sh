set -e
mkdir -p gpurun_out
python scripts/bench_configs.py c2f_pbp > gpurun_out/r4_c2f.jsonl 2> gpurun_out/r4_c2f.err || { tail -30 gpurun_out/r4_c2f.err; exit 1; }
tail -1 gpurun_out/r4_c2f.jsonl
