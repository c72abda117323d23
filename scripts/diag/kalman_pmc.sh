#!/bin/bash
# SQ counters of the Gaussian pull kernel on the Kalman-filter graph (rows of 66 entries, 360 distinct potentials): two separate
# --pmc passes (never combined with a trace domain)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
export KALMAN_T=${KALMAN_T:-12000} GAUSS_REL_ONLY=kalman
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/kalman_sq1 -- python3 $R/scripts/bench_configs.py gauss_rel > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM --output-format csv -d $O/kalman_sq2 -- python3 $R/scripts/bench_configs.py gauss_rel > /dev/null 2>&1
rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $O/kalman_ta -- python3 $R/scripts/bench_configs.py gauss_rel > /dev/null 2>&1 || true
cd $R
python3 - <<'PY'
import collections, csv, glob, os
for d in ('gpurun_out/kalman_sq1', 'gpurun_out/kalman_sq2', 'gpurun_out/kalman_ta'):
    fs = sorted(glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True), key=os.path.getmtime, reverse=True)
    if not fs:
        print(d, 'no counters'); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        if 'gabp' in r['Kernel_Name']:
            acc[r['Kernel_Name'].split('(')[0][-50:]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in acc.items():
        print(k, {c: '%.4g' % (sum(v) / len(v)) for c, v in cs.items()}, 'launches', len(next(iter(cs.values()))))
PY
