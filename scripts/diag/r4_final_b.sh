set -e
bash scripts/profile_secondary.sh r04_secondary > gpurun_out/r04_secondary.out 2>&1 || { tail -20 gpurun_out/r04_secondary.out; exit 1; }
tail -3 gpurun_out/r04_secondary.out | cut -c1-200
bash scripts/profile_cfg3.sh r04_cfg3 > gpurun_out/r04_cfg3.out 2>&1 || { tail -20 gpurun_out/r04_cfg3.out; exit 1; }
python scripts/bench_configs.py cfg3s > gpurun_out/r04_cfg3_routed_vs_generic.jsonl 2> gpurun_out/r04_cfg3_plain.err
bash scripts/profile_color.sh r04_color > gpurun_out/r04_color.out 2>&1 || { tail -20 gpurun_out/r04_color.out; exit 1; }
echo done b
