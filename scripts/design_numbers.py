"""Generate the measured-numbers block of docs/measurement.md (DESIGN.md section 5 until round 4) from the committed profile files,
so that the document cannot quote a number that no file holds.

usage: python scripts/design_numbers.py <tag>            -> prints the block (markdown)
       python scripts/design_numbers.py <tag> --write    -> also rewrites the block between the markers in docs/measurement.md

Inputs (all under profiles/): <tag>_bench_default.json (bench.py's line, plain run), <tag>_final_bench_under_rocprof.json,
<tag>_final_kernel_stats.csv (rocprofv3 --kernel-trace --stats of bench.py), <tag>_final_traffic.json / _pmc.md (separate
--pmc passes, scripts/summarize_pmc.py), <tag>_secondary_configs.jsonl (scripts/bench_configs.py), <tag>_secondary_traffic.json.
tests/test_host_api.py::test_design_numbers_are_the_committed_profiles keeps docs/measurement.md equal to this script's output.
"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, 'profiles')
BEGIN, END = '<!-- numbers:begin (scripts/design_numbers.py) -->', '<!-- numbers:end -->'


def _json_line(path):
    with open(path) as fh:
        for line in fh:
            line = line.strip()
            if line.startswith('{'):
                return json.loads(line)
    raise ValueError('no JSON line in ' + path)


def _short(name):
    name = name.split('(')[0].replace('void ', '').replace('lhvi::', '')
    return name


def kernel_stats(path):
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            rows.append((_short(r['Name']), int(r['Calls']), float(r['AverageNs']) / 1e6, float(r['Percentage'])))
    return rows


def pmc_table(path, wanted):
    """{kernel: {counter: value}} from the markdown tables of <tag>_pmc.md"""
    out = {}
    header = None
    with open(path) as fh:
        for line in fh:
            cells = [c.strip() for c in line.strip().strip('|').split('|')]
            if line.startswith('| kernel'):
                header = cells
            elif header and line.startswith('|') and not line.startswith('|---'):
                k = _short(cells[0])
                for name, val in zip(header[1:], cells[1:]):
                    if name in wanted:
                        try:
                            out.setdefault(k, {})[name] = float(val)
                        except ValueError:
                            pass
    return out


def block(tag):
    L = []
    b = _json_line(os.path.join(PROF, tag + '_bench_default.json'))
    r = b['roofline']
    L.append('Headline (`profiles/%s_bench_default.json`, `python bench.py` on one MI355X): **%.1f sweeps/s = %.3ge9 edge-messages/s, '
             '%.2f ms per sweep** at E = %d, n = %d, T = %d.' % (tag, b['value'], b['edge_messages_per_sec'] / 1e9, b['ms_per_step'],
                                                                 b['config']['edges'], b['config']['particles'], b['config']['integral_points']))
    L.append('Dominant kernel `%s`: %.2f ms per launch (HIP events), %.1f TFLOP/s algorithmic = **%.3f of the fp64 vector roof** '
             '(%.1f); executed %.1f TFLOP/s = %.3f; algorithmic HBM view %.0f GB/s = %.3f of 8 TB/s; whole sweep %.0f GB/s = %.3f '
             '(target %.2f: %s).' % (r['kernel'], r['kernel_ms'], r['achieved'], r['frac'], r['peak'], r['executed']['achieved'],
                                     r['executed']['frac'], r['hbm']['achieved'], r['hbm']['frac'], r['sweep_hbm']['achieved'],
                                     r['sweep_hbm']['frac'], r['hbm_target'], 'met' if r['hbm_target_met'] else 'missed'))
    c = b.get('cpu_baseline')
    if c:
        L.append('CPU baselines of the same run: C port (OpenMP, %d threads) %.3g edge-messages/s; pure-Python restatement (1 thread) '
                 '%.3g edge-messages/s.' % (c['cores'], c['edge_messages_per_sec'], c['python']['edge_messages_per_sec']))
    u = _json_line(os.path.join(PROF, tag + '_final_bench_under_rocprof.json'))
    L.append('')
    L.append('Per-kernel averages (`profiles/%s_final_kernel_stats.csv`, rocprofv3 `--kernel-trace --stats` of `bench.py`; that run: '
             '%.1f sweeps/s, %.2f ms per sweep) and HBM bytes per launch (`profiles/%s_final_traffic.json`, separate `--pmc` passes):'
             % (tag, u['value'], u['ms_per_step'], tag))
    L.append('')
    L.append('| kernel | calls | avg ms | % of GPU time | HBM read GB | HBM write GB | TB/s |')
    L.append('|---|---|---|---|---|---|---|')
    traffic = {_short(k): v for k, v in json.load(open(os.path.join(PROF, tag + '_final_traffic.json'))).items()}
    for name, calls, ms, pct in kernel_stats(os.path.join(PROF, tag + '_final_kernel_stats.csv')):
        if pct < 0.5 or not name.startswith(('pbp_', 'gabp_', 'vi_')):
            continue
        t = traffic.get(name)
        L.append('| `%s` | %d | %.3f | %.1f | %s | %s | %s |' % (
            name, calls, ms, pct, '%.2f' % (t['read_bytes'] / 1e9) if t else '', '%.2f' % (t['write_bytes'] / 1e9) if t else '',
            '%.2f' % (t['hbm_bytes'] / ms / 1e9) if t else ''))
    want = ('SQ_INSTS_VALU', 'SQ_INSTS_LDS', 'SQ_INSTS_SALU', 'GRBM_GUI_ACTIVE', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_ACTIVE_INST_VALU')
    pmc = pmc_table(os.path.join(PROF, tag + '_final_pmc.md'), want).get('pbp_f2v_heavy_kernel', {})
    if pmc:
        L.append('')
        L.append('`pbp_f2v_heavy_kernel` counters per launch (`profiles/%s_final_pmc.md`): %.3ge9 VALU, %.3ge9 LDS and %.3ge9 scalar '
                 'wave-instructions in %.3ge8 GRBM cycles; LDS bank conflicts %.3ge9 of %.3ge9 LDS-active cycles (%.0f %%).'
                 % (tag, pmc.get('SQ_INSTS_VALU', 0) / 1e9, pmc.get('SQ_INSTS_LDS', 0) / 1e9, pmc.get('SQ_INSTS_SALU', 0) / 1e9,
                    pmc.get('GRBM_GUI_ACTIVE', 0) / 1e8, pmc.get('SQ_LDS_BANK_CONFLICT', 0) / 1e9, pmc.get('SQ_LDS_IDX_ACTIVE', 0) / 1e9,
                    100.0 * pmc.get('SQ_LDS_BANK_CONFLICT', 0) / max(pmc.get('SQ_LDS_IDX_ACTIVE', 1), 1)))
    L.append('')
    L.append('Other configurations (`profiles/%s_secondary_configs.jsonl`, `scripts/bench_configs.py`):' % tag)
    L.append('')
    with open(os.path.join(PROF, tag + '_secondary_configs.jsonl')) as fh:
        for line in fh:
            line = line.strip()
            if not line.startswith('{'):
                continue
            d = json.loads(line)
            cfg = d.pop('config')
            L.append('* %s: %s' % (cfg, ', '.join('%s = %s' % (k, ('%.4g' % v) if isinstance(v, float) else v) for k, v in d.items())))
    for extra, what in (('_cfg3_configs.jsonl', 'scaled cfg 3 under rocprofv3 (`scripts/profile_cfg3.sh`)'),
                        ('_cfg3_routed_vs_generic.jsonl', 'scaled cfg 3, plain run, conditional-quadratic routing against the generic kernel')):
        path = os.path.join(PROF, tag + extra)
        if os.path.exists(path):
            d = _json_line(path)
            cfg = d.pop('config')
            L.append('* (`profiles/%s%s`: %s) %s: %s' % (tag, extra, what, cfg, ', '.join(
                '%s = %s' % (k, ('%.4g' % v) if isinstance(v, float) else v) for k, v in d.items())))
    # round 4: further measurement files, every JSON line of each (long per-sweep lists dropped)
    for extra, what in (('_c2f_configs.jsonl', 'particle coarse-to-fine on arrays, `scripts/bench_configs.py c2f_pbp`'),
                        ('_vi_configs.jsonl', 'the variational step on the models the reference published timings for, `scripts/bench_configs.py vi_models vi_scaled`'),
                        ('_particles.jsonl', 'the headline workload at the demos\' particle counts and with EP proposals, `scripts/profile_particles.sh`'),
                        ('_demo_loop.jsonl', 'Demo/RGM/demo.py through the object API, `scripts/bench_configs.py demo_loop`'),
                        ('_refsize.jsonl', 'the reference\'s own calls at the reference\'s sizes through the unchanged API: cold (first call of a process) and warm wall time, '
                                           'the C oracle beside; `scripts/bench_configs.py refsize`')):
        path = os.path.join(PROF, tag + extra)
        if not os.path.exists(path):
            continue
        L.append('')
        L.append('`profiles/%s%s` (%s):' % (tag, extra, what))
        L.append('')
        with open(path) as fh:
            for line in fh:
                line = line.strip()
                if not line.startswith('{'):
                    continue
                d = json.loads(line)
                if extra == '_particles.jsonl':
                    L.append('* n = %d, %s proposals: %.1f sweeps/s, %.2f ms per sweep, heavy-class kernel %.2f ms' % (
                        d['config']['particles'], d['config']['proposal'], d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))
                    continue
                cfg = d.pop('config')
                for drop in ('model', 'totals_s'):
                    d.pop(drop, None)
                def fmt(v):
                    if isinstance(v, float):
                        return '%.4g' % v
                    if isinstance(v, dict):
                        return '{' + ', '.join('%s: %s' % (k, fmt(x)) for k, x in v.items()) + '}'
                    if isinstance(v, list):
                        return '[' + ', '.join(fmt(x) for x in v) + ']'
                    return str(v)
                L.append('* %s: %s' % (cfg, ', '.join('%s = %s' % (k, fmt(v)) for k, v in d.items())))
    sec = os.path.join(PROF, tag + '_secondary_traffic.json')
    if os.path.exists(sec):
        t = {_short(k): v for k, v in json.load(open(sec)).items()}
        parts = ['`%s` %.2f GB read + %.2f GB written' % (k, v['read_bytes'] / 1e9, v['write_bytes'] / 1e9)
                 for k, v in t.items() if k.startswith('gabp_') and k != 'gabp_init_kernel']
        if parts:
            L.append('')
            L.append('Gaussian sweep at 10 M edges, HBM bytes per launch (`profiles/%s_secondary_traffic.json`): %s.' % (tag, '; '.join(sorted(parts))))
    gt = os.path.join(PROF, tag + '_gauss_traffic.json')
    if os.path.exists(gt):
        t = json.load(open(gt))
        L.append('')
        L.append('Gaussian sweep, HBM bytes per launch PER GRAPH (`profiles/%s_gauss_traffic.json`: one pair of `--pmc` passes per graph, '
                 '`scripts/profile_gauss.sh`; algorithmic = 76 B x edges):' % tag)
        L.append('')
        edges = {}
        cfgs = os.path.join(PROF, tag + '_gauss_configs.jsonl')
        if os.path.exists(cfgs):
            for line in open(cfgs):
                if line.startswith('{'):
                    d = json.loads(line)
                    key = 'random' if 'random' in d['config'] else ('rgm' if 'RGM' in d['config'] else 'kalman')
                    if 'pull form' in d['config'] and 'graph arrays' not in d['config']:
                        edges[key] = (d['edges'], d.get('sweep_ms'), d.get('hbm_frac'))
        for gname, ks in t.items():
            # one sweep of the records form = the row kernel + the hub-row kernel (rows of more than 512 entries: the RGM's templates)
            rec = [v for k, v in ks.items() if 'pull' in k and 'rec' in k]
            if not rec:
                continue
            rd, wr = sum(v['read_bytes'] for v in rec), sum(v['write_bytes'] for v in rec)
            e = edges.get(gname)
            L.append('* %s (%s): %.3f GB read + %.3f GB written%s' % (
                gname, ' + '.join('`%s`' % _short(k) for k in sorted(ks) if 'pull' in k and 'rec' in k), rd / 1e9, wr / 1e9,
                (' = %.2f x the algorithmic %.3f GB; sweep %.3f ms, %.3f of the HBM roof'
                 % ((rd + wr) / (76.0 * e[0]), 76.0 * e[0] / 1e9, e[1], e[2])) if e else ''))
    sim = os.path.join(PROF, tag + '_sim_sharded.json')
    if os.path.exists(sim):
        t = json.load(open(sim))
        L.append('')
        L.append('One-GPU rehearsal of the sharded sweep (`profiles/%s_sim_sharded.json`; per-rank compute only, the exchange is NOT '
                 'included -- not a throughput figure):' % tag)
        L.append('')
        for k, v in t.items():
            if k == 'note' or not v.get('sweeps'):
                continue
            plan = v.get('plan', {})
            L.append('* %s: %s; device time per rank and sweep %s ms%s' % (
                k, v['sweeps'][-1].split(': ', 1)[1], v.get('device_ms_per_rank_and_sweep', {}).get('total'),
                ('; payload %.3f GB per sweep, busiest pair %.1f MB' % (plan['total_payload_GB'], plan['busiest_pair_MB'])) if plan else ''))
    return '\n'.join(L)


def main():
    tag = sys.argv[1]
    text = block(tag)
    if '--write' in sys.argv:
        path = os.path.join(ROOT, 'docs', 'measurement.md')
        s = open(path).read()
        pat = re.compile(re.escape(BEGIN) + '.*?' + re.escape(END), re.S)
        if not pat.search(s):
            raise SystemExit('markers not found in docs/measurement.md')
        s = pat.sub(lambda m: BEGIN + '\n' + text + '\n' + END, s)
        open(path, 'w').write(s)
    print(text)


if __name__ == '__main__':
    main()
