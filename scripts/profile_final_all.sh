#!/bin/bash
# the round's judged profiles in two gpurun calls (run from the repo root through gpurun):
#   part 1: bench.py's default line, its kernel trace and PMC passes (profile_round.sh), the few-particle lines (profile_particles.sh)
#   part 2: the secondary configurations (profile_secondary.sh), the Gaussian sweep's traffic per graph (profile_gauss.sh), PMC passes
#           of the n = 10 sweep (profile_fewparticles_pmc.sh)
# usage: scripts/profile_final_all.sh <round tag, e.g. r05> <1|2>
tag=${1:-r05}; part=${2:-1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
if [ $part = 1 ]; then
  timeout -k 10 400 python3 bench.py > $O/${tag}_bench_default.json 2> $O/${tag}_bench_default.log || exit 1
  cp $O/${tag}_bench_default.json profiles/${tag}_bench_default.json
  cut -c1-300 $O/${tag}_bench_default.json
  bash scripts/profile_round.sh ${tag}_final || exit 1
  python3 scripts/summarize_pmc.py ${tag}_final $O/${tag}_final_stats $O/${tag}_final_fetch $O/${tag}_final_write $O/${tag}_final_sq1 $O/${tag}_final_sq2 || exit 1
  cp $O/${tag}_final_bench_under_rocprof.json profiles/ 2>/dev/null
  cp profiles/${tag}_final_* $O/ 2>/dev/null
  bash scripts/profile_particles.sh ${tag}_particles || exit 1
elif [ $part = 2 ]; then
  bash scripts/profile_secondary.sh ${tag}_secondary || exit 1
  python3 scripts/summarize_pmc.py ${tag}_secondary $O/${tag}_secondary_stats $O/${tag}_secondary_fetch $O/${tag}_secondary_write || exit 1
  cp profiles/${tag}_secondary_* $O/ 2>/dev/null
  bash scripts/profile_gauss.sh ${tag}_gauss || exit 1
  bash scripts/profile_fewparticles_pmc.sh ${tag}_n10 10 || exit 1
fi
if [ $part = 3 ]; then        # PMC passes of the few-particle sweeps only
  for n in 16 20; do bash scripts/profile_fewparticles_pmc.sh ${tag}_n$n $n > $O/${tag}_n$n.log 2>&1 || exit 1; done
  grep -h -E "small_kernel|var_fused" $O/${tag}_n16.log $O/${tag}_n20.log | cut -c1-160
fi
