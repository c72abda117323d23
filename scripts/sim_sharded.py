"""Rehearse the edge-sharded sweep for `world` ranks on ONE GPU (loopback exchange): checks that the plans build at full
size and times each rank's three phases -- boundary pack, interior part (overlaps the exchange), boundary part -- i.e. everything
except the real all_to_all.  Tuning / validation aid."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'lifted-hybrid-variational-inference_amd')]
import numpy as np, torch
from lhvi import synth, dist

E = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
flat = synth.hybrid_mrf_flat(V=E // 4, deg=4, seed=0)
group = dist.LoopbackGroup(world)
t0 = time.perf_counter()
fac_owner = dist.partition_factors(flat, world)           # the one global step, done once (rank 0's job in a real run)
runners = [dist.ShardedRunner(flat, n=64, seed=1, rank=r, world=world, group=group, fac_owner=fac_owner) for r in range(world)]
print('plans + setup: %.1f s' % (time.perf_counter() - t0), flush=True)
for r in runners:
    r.init()
torch.cuda.synchronize()
nb = [r.nb for r in runners]
sent = [r.n_elems * 8 / 1e6 for r in runners]
print('boundary vars per rank', nb[:3], '... send MB per rank', ['%.0f' % s for s in sent[:3]], 'per peer MB %.0f' % (sent[0] / max(world - 1, 1)))
print('interior variables per rank: %s of %s' % ([r.n_int for r in runners[:3]], [r.plan.flat.V for r in runners[:3]]),
      'interior heavy edges: %s of %s' % ([r.bp.part_counts['heavy'] for r in runners[:3]], [r.bp.n_heavy for r in runners[:3]]))


def timed(fn):
    torch.cuda.synchronize(); t = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, time.perf_counter() - t


for it in range(3):
    tp, ti, tb = [], [], []
    sends = []
    for r in runners:
        send, t = timed(lambda: r.pre(part=1))          # boundary partials + pack: precedes the exchange
        sends.append(send); tp.append(t)
    for r, s in zip(runners, sends):
        group.post(r.rank, s, r.counts)
    for r in runners:
        ti.append(timed(r.interior)[1])                 # runs while the rows are in flight
    for r in runners:
        recv = group.collect(r.rank, r.W)
        tb.append(timed(lambda: r.boundary(recv))[1])
    print('sweep %d: per-rank pack %.2f ms, interior %.2f ms, boundary %.2f ms (max over ranks %.2f / %.2f / %.2f; slowest rank total %.2f)' %
          (it, 1e3 * np.mean(tp), 1e3 * np.mean(ti), 1e3 * np.mean(tb), 1e3 * max(tp), 1e3 * max(ti), 1e3 * max(tb),
           1e3 * max(a + b + c for a, b, c in zip(tp, ti, tb))), flush=True)
