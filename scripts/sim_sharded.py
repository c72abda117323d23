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
mode = sys.argv[3] if len(sys.argv) > 3 else 'pairs'          # pairs | ownercompute
flat = synth.hybrid_mrf_flat(V=E // 4, deg=4, seed=0)
group = dist.LoopbackGroup(world)


def timed(fn):
    torch.cuda.synchronize(); t = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, time.perf_counter() - t


if mode == 'ownercompute':
    # owner-computes split: per rank the three phases -- v->f + proposal of the owned variables + pack, interior part (new
    # particles of the owned variables + f->v of the factors without a ghost: overlaps the exchange), boundary part (unpack, ghosts'
    # particles, f->v of the cut factors) -- and the payload table
    import json
    t0 = time.perf_counter()
    owner = dist.partition_variables(flat, world)
    runners = [dist.OwnerRunner(flat, n=64, seed=1, rank=r, world=world, group=group, var_owner=owner) for r in range(world)]
    print('plans + setup: %.1f s' % (time.perf_counter() - t0), flush=True)
    for r in runners:
        r.init()
    torch.cuda.synchronize()
    sent = np.array([[8e-6 * c for c in r.lay['send']['counts']] for r in runners])
    print(json.dumps(dict(world=world, edges=int(flat.E), owned=[r.n_owned for r in runners], ghosts=[r.n_ghost for r in runners],
                          local_target_edges=[r.local_edges() for r in runners], cut_rows_sent=[int(r.lay['send']['row_edge'].size + r.lay['send']['cont_edge'].size) for r in runners],
                          send_MB_per_rank=[round(float(x), 1) for x in sent.sum(axis=1)], busiest_pair_MB=round(float(sent.max()), 1),
                          total_payload_GB=round(float(sent.sum()) / 1e3, 3))), flush=True)
    for it in range(3):
        ta, ti, tb = [], [], []
        sends = []
        for r in runners:
            send, t = timed(r.owned_half)
            sends.append(send); ta.append(t)
        for r, s in zip(runners, sends):
            group.post(r.rank, s, r.counts)
        for r in runners:
            ti.append(timed(r.interior)[1])
        for r in runners:
            recv = group.collect(r.rank, None)
            tb.append(timed(lambda: r.boundary(recv))[1])
        print('sweep %d: per-rank owned half %.2f ms, interior %.2f ms, boundary %.2f ms (max over ranks %.2f / %.2f / %.2f; slowest rank total %.2f; mean total %.2f)' %
              (it, 1e3 * np.mean(ta), 1e3 * np.mean(ti), 1e3 * np.mean(tb), 1e3 * max(ta), 1e3 * max(ti), 1e3 * max(tb),
               1e3 * max(a + b + c for a, b, c in zip(ta, ti, tb)), 1e3 * np.mean([a + b + c for a, b, c in zip(ta, ti, tb)])), flush=True)
    sys.exit(0)

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
