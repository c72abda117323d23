set -e
for w in 2 4 8; do
  python scripts/sim_sharded.py 10000000 $w ownercompute > gpurun_out/r4_sim_oc_$w.log 2>&1 || { tail -20 gpurun_out/r4_sim_oc_$w.log; exit 1; }
  tail -4 gpurun_out/r4_sim_oc_$w.log
done
python scripts/sim_sharded.py 10000000 8 pairs > gpurun_out/r4_sim_pairs_8.log 2>&1 || { tail -20 gpurun_out/r4_sim_pairs_8.log; exit 1; }
tail -3 gpurun_out/r4_sim_pairs_8.log
