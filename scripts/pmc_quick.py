"""print per-kernel averages of every counter in rocprofv3 --pmc output dirs (tuning aid)
usage: python scripts/pmc_quick.py <dir> [<dir> ...]"""
import collections, csv, glob, os, sys
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, cs in acc.items():
            if 'f2v_fast' in k:
                print(k, {c: '%.4g' % (sum(v) / len(v)) for c, v in cs.items()}, 'launches', len(next(iter(cs.values()))))
