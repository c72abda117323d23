#!/bin/bash
# HBM traffic of the Gaussian sweep PER GRAPH: one pair of rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; never combined with a
# trace domain) for each of the three 10 M-edge-class graphs -- the random expander, the RGM ground graph, the Kalman-filter graph --
# so that a kernel's per-launch average belongs to one graph.  Writes profiles/<tag>_traffic.json keyed graph -> kernel and
# gpurun_out/<tag>_configs.jsonl with the sweep times of the same three.  Run through gpurun from the repo root.
set -e
tag=${1:-r05_gauss}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
export KALMAN_T=${KALMAN_T:-12000}
cd /tmp && export TMPDIR=/tmp
for gname in random rgm kalman; do
  if [ $gname = random ]; then what=gauss; unset GAUSS_REL_ONLY; else what=gauss_rel; export GAUSS_REL_ONLY=$gname; fi
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${tag}_${gname}_fetch -- python3 $R/scripts/bench_configs.py $what > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${tag}_${gname}_write -- python3 $R/scripts/bench_configs.py $what > /dev/null 2>&1
done
unset GAUSS_REL_ONLY
cd $R
python3 scripts/bench_configs.py gauss gauss_rel > $O/${tag}_configs.jsonl 2> $O/${tag}_plain.log
python3 - $tag <<'PY'
import collections, csv, glob, json, os, sys
tag = sys.argv[1]
O = 'gpurun_out'
def counters(d):
    f = sorted(glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True), key=os.path.getmtime, reverse=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        acc[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
out = {}
for g in ('random', 'rgm', 'kalman'):
    fe, wr = counters('%s/%s_%s_fetch' % (O, tag, g)), counters('%s/%s_%s_write' % (O, tag, g))
    out[g] = {k: {'read_bytes': 2.0 * v.get('FETCH_SIZE', 0.0) * 1024, 'write_bytes': wr.get(k, {}).get('WRITE_SIZE', 0.0) * 1024}
              for k, v in fe.items() if 'gabp' in k}
    for k, v in out[g].items():
        v['hbm_bytes'] = v['read_bytes'] + v['write_bytes']
json.dump(out, open('profiles/%s_traffic.json' % tag, 'w'), indent=1)
open('%s/%s_traffic.json' % (O, tag), 'w').write(json.dumps(out, indent=1))
for g, ks in out.items():
    for k, v in ks.items():
        print(g, k[-40:], round(v['hbm_bytes'] / 1e9, 3), 'GB')
PY
cut -c1-230 $O/${tag}_configs.jsonl
