"""device time per rank and sweep of a rehearsal (scripts/profile_sim_sharded.sh): sums the sweep kernels of the kernel-trace
statistics and divides by ranks x sweeps.  usage: python scripts/sim_kernel_time.py <kernel_stats.csv> <world> [sweeps=3]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
world, sweeps = int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 3
n = world * sweeps
setup = ('describe', 'classify', 'init_kernel', 'resample_uniq_kernel<false>')       # once per rank, not per sweep
tot, lines = 0.0, []
for r in rows:
    name = r['Name'].split('(')[0].replace('void ', '')
    if 'lhvi::' not in name or any(s in name for s in setup):
        continue
    per = float(r['TotalDurationNs']) / n / 1e6
    tot += per
    lines.append((per, name.replace('lhvi::', ''), int(r['Calls'])))
for per, name, calls in sorted(lines, reverse=True):
    print('  %-38s %5d launches  %.3f ms per rank and sweep' % (name[:38], calls, per))
print('device time per rank and sweep (mean over ranks): %.3f ms' % tot)
