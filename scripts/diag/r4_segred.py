import numpy as np, torch
rng = np.random.default_rng(0)
for n, nseg in ((1000, 37), (100000, 500), (250000, 4000)):
    vals = rng.uniform(-30, 30, n)
    seg = np.sort(rng.integers(0, nseg, n))
    lengths = np.bincount(seg, minlength=nseg)
    want = np.zeros(nseg); np.add.at(want, seg, vals)
    got = torch.segment_reduce(torch.from_numpy(vals).cuda(), 'sum', lengths=torch.from_numpy(lengths).cuda(), unsafe=True).cpu().numpy()
    got_cpu = torch.segment_reduce(torch.from_numpy(vals), 'sum', lengths=torch.from_numpy(lengths), unsafe=True).numpy()
    print(n, nseg, 'cuda == sequential:', (got == want).all(), int((got != want).sum()), 'cpu == sequential:', (got_cpu == want).all())
