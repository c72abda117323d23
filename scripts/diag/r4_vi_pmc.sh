#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES --output-format csv -d $O/r4_vipmc1 -- python3 $R/scripts/bench_configs.py vi_scaled > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $O/r4_vipmc2 -- python3 $R/scripts/bench_configs.py vi_scaled > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections, os
for d in ('gpurun_out/r4_vipmc1', 'gpurun_out/r4_vipmc2'):
    f = sorted(glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True), key=os.path.getmtime)[-1]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if 'tiny' in r['Kernel_Name'] or 'cc_tab' in r['Kernel_Name']:
            acc[r['Kernel_Name'].split('(')[0][-30:]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in acc.items():
        print(k, {c: '%.3g' % (sum(v) / len(v)) for c, v in cs.items()})
PY
