"""Turn rocprofv3 outputs (kernel-trace stats + separate --pmc passes) into the committed summaries under profiles/.

usage: python scripts/summarize_pmc.py <round tag> <stats dir> <fetch dir> <write dir> [<sq dir> ...]
Writes profiles/<tag>_kernel_stats.csv (copy), profiles/<tag>_traffic.json (per-kernel HBM bytes per launch, with the
gfx950 correction: FETCH_SIZE counts 64-B requests for 128-B fetches, so read bytes = 2 * FETCH_SIZE * 1024;
WRITE_SIZE * 1024 is exact) and profiles/<tag>_pmc.md."""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, stats_dir, fetch_dir, write_dir = sys.argv[1:5]
sq_dirs = sys.argv[5:]
prof = os.path.join(ROOT, 'profiles')


def counters(d):
    # (gpurun merges every call's outputs into the same directory: take the newest collection)
    f = sorted(glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True), key=os.path.getmtime, reverse=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        acc[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


st = sorted(glob.glob(os.path.join(stats_dir, '**', '*kernel_stats.csv'), recursive=True), key=os.path.getmtime, reverse=True)
if st:
    shutil.copy(st[0], os.path.join(prof, tag + '_kernel_stats.csv'))
fetch, write = counters(fetch_dir), counters(write_dir)
traffic = {}
for k in fetch:
    if 'lhvi' not in k:
        continue
    rd = 2.0 * fetch[k].get('FETCH_SIZE', 0.0) * 1024
    wr = write.get(k, {}).get('WRITE_SIZE', 0.0) * 1024
    traffic[k] = {'read_bytes': rd, 'write_bytes': wr, 'hbm_bytes': rd + wr}
json.dump(traffic, open(os.path.join(prof, tag + '_traffic.json'), 'w'), indent=1)
lines = ['# %s PMC summary (per launch averages)\n' % tag, '| kernel | read GB (2x FETCH_SIZE) | write GB | ' + ' | '.join([]) + '\n']
lines = ['# %s PMC summary (per-launch averages; see the file name for the workload)\n\n' % tag,
         '| kernel | HBM read GB (2 x FETCH_SIZE) | HBM write GB |\n|---|---|---|\n']
for k, v in sorted(traffic.items(), key=lambda kv: -kv[1]['hbm_bytes']):
    lines.append('| %s | %.3f | %.3f |\n' % (k, v['read_bytes'] / 1e9, v['write_bytes'] / 1e9))
for d in sq_dirs:
    c = counters(d)
    names = sorted({n for v in c.values() for n in v})
    lines.append('\n| kernel | ' + ' | '.join(names) + ' |\n|---|' + '---|' * len(names) + '\n')
    for k, v in c.items():
        if 'lhvi' in k:
            lines.append('| %s | ' % k + ' | '.join('%.4g' % v.get(n, float('nan')) for n in names) + ' |\n')
open(os.path.join(prof, tag + '_pmc.md'), 'w').writelines(lines)
print('wrote', tag, list(traffic)[:3])
