set -e
python -m pytest tests -x -q -m gpu > gpurun_out/r4_t10.log 2>&1 || { tail -40 gpurun_out/r4_t10.log; exit 1; }
tail -2 gpurun_out/r4_t10.log
VI_NO_PMC= bash scripts/profile_vi.sh r04_vi > gpurun_out/r4_profile_vi.out 2>&1 || { tail -20 gpurun_out/r4_profile_vi.out; exit 1; }
tail -3 gpurun_out/r4_profile_vi.out | cut -c1-200
