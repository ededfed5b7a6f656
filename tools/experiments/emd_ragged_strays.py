import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from rfnet_amd import _raw as R
from oracle.oracle import Oracle
orc = Oracle()
for n, m in [(2048, 640), (640, 2048), (513, 700), (2000, 700), (700, 2000)]:
    rng = np.random.RandomState(n + m)
    a = (rng.random_sample((2, n, 3)) - 0.5).astype(np.float32); c = (rng.random_sample((2, m, 3)) - 0.5).astype(np.float32)
    om = orc.approx_match(a[:1], c[:1])
    for mode in ("auto", "swept"):
        gm = R.approx_match(torch.from_numpy(a).cuda(), torch.from_numpy(c).cuda(), mode=mode)[:1].cpu().numpy()
        bad = np.abs(gm - om) > 1e-6 + 1e-4 * np.abs(om)
        print(n, m, mode, int(bad.sum()), float(np.abs(gm - om).max()), "colsum", float(np.abs(gm.sum(1) - om.sum(1)).max()), "rowsum", float(np.abs(gm.sum(2) - om.sum(2)).max()))
