set -e
python scripts/bench_configs.py vi_models vi_scaled > gpurun_out/r4_vi_models.jsonl 2> gpurun_out/r4_vi_models.err || { tail -30 gpurun_out/r4_vi_models.err; exit 1; }
cut -c1-900 gpurun_out/r4_vi_models.jsonl
