"""profiles/<tag>_sim_sharded.json from the rehearsal logs and kernel traces of scripts/profile_sim_sharded.sh (gpurun_out/<tag>_sim_*):
per split and rank count the plan statistics, the wall-clock phase times of every sweep (host launch latencies and the
synchronisations between the phases included) and the DEVICE time per rank and sweep from the kernel trace (sweep kernels only).
usage: python scripts/summarize_sim.py r05"""
import csv, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
O = os.path.join(ROOT, 'gpurun_out')
setup = ('describe', 'classify', 'init_kernel', 'resample_uniq_kernel<false>')
out = {'note': 'one-GPU rehearsal (scripts/sim_sharded.py): every simulated rank runs its phases one after the other on the same device, the '
               'exchange itself is NOT included -- no multi-GPU node was available: none of this is a throughput figure.  wall = host clock '
               'around each phase with a synchronisation on both sides; device = sum of the sweep kernels of the rocprofv3 kernel trace '
               'of the same run / (ranks x sweeps).  Single GPU: 15.7-16.0 ms per sweep, an eighth is 1.96-2.0 ms.'}
for name in sorted(os.listdir(O)):
    m = re.match(tag + r'_sim_(oc|pairs)(\d+)\.log$', name)
    if not m:
        continue
    mode, world = {'oc': 'ownercompute', 'pairs': 'pairs'}[m.group(1)], int(m.group(2))
    text = open(os.path.join(O, name)).read().splitlines()
    rec = {'sweeps': [l for l in text if l.startswith('sweep ')]}
    plan = [l for l in text if l.startswith('{')]
    if plan:
        rec['plan'] = json.loads(plan[0])
    rec['setup'] = [l for l in text if l.startswith('plans + setup') or l.startswith('boundary vars') or l.startswith('interior variables')]
    stats = os.path.join(O, '%s_sim_%s%d_kernel_stats.csv' % (tag, m.group(1), world))
    if os.path.exists(stats):
        n = world * 3
        per = {}
        for r in csv.DictReader(open(stats)):
            k = r['Name'].split('(')[0].replace('void ', '')
            if 'lhvi::' in k and not any(s in k for s in setup):
                per[k.replace('lhvi::', '')] = round(float(r['TotalDurationNs']) / n / 1e6, 4)
        rec['device_ms_per_rank_and_sweep'] = {'total': round(sum(per.values()), 3), 'by_kernel': dict(sorted(per.items(), key=lambda kv: -kv[1]))}
    out['%s_%d' % (mode, world)] = rec
json.dump(out, open(os.path.join(ROOT, 'profiles', tag + '_sim_sharded.json'), 'w'), indent=1)
for k, v in out.items():
    if k != 'note':
        print(k, v['sweeps'][-1][9:] if v['sweeps'] else '', '| device', v.get('device_ms_per_rank_and_sweep', {}).get('total'))
