set -e
bash scripts/profile_c2f.sh r04_c2f > gpurun_out/r04_c2f.out 2>&1 || { tail -20 gpurun_out/r04_c2f.out; exit 1; }
bash scripts/profile_vi.sh r04_vi > gpurun_out/r04_vi.out 2>&1 || { tail -20 gpurun_out/r04_vi.out; exit 1; }
bash scripts/profile_particles.sh r04_particles
python scripts/bench_configs.py demo_loop > gpurun_out/r04_demo_loop.jsonl 2> gpurun_out/r04_demo_loop.err
python scripts/bench_queries.py 10000000 gpurun_out/r04_queries.json > /dev/null 2> gpurun_out/r04_queries.err
for w in 2 4 8; do python scripts/sim_sharded.py 10000000 $w ownercompute > gpurun_out/r04_sim_ownercompute_$w.log 2>&1; python scripts/sim_sharded.py 10000000 $w pairs > gpurun_out/r04_sim_pairs_$w.log 2>&1; done
tail -1 gpurun_out/r04_sim_ownercompute_8.log gpurun_out/r04_sim_pairs_8.log
echo done c
python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err
cut -c1-200 gpurun_out/r04_bench_default.json
