import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'lifted-hybrid-variational-inference_amd')]
import numpy as np, torch
from lhvi import lifting
rng = np.random.default_rng(0)
for n, nseg in ((250_000, 1), (250_000, 72), (250_000, 437), (250_000, 5000), (1_000_000, 400_000), (250_000, 2_500_000)):
    vals = torch.from_numpy(rng.uniform(-30, 30, n)).cuda()
    seg = np.sort(rng.integers(0, nseg, n))
    lengths = torch.from_numpy(np.bincount(seg, minlength=nseg)).cuda()
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = lifting.segment_sums(vals, lengths)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ref = torch.segment_reduce(vals, 'sum', lengths=lengths, unsafe=True)
    torch.cuda.synchronize(); dt2 = time.perf_counter() - t0
    print('n %d segments %d: segment_sums %.3f ms (torch tree reduction %.3f ms)' % (n, nseg, dt * 1e3, dt2 * 1e3))
