import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from oracle.oracle import Oracle
from rfnet_amd import _raw as R
orc = Oracle()
for (n, m) in ((4096, 4096), (4096, 5000)):
    rng = np.random.RandomState(n + m)
    a = (rng.random_sample((1, n, 3)) - 0.5).astype(np.float32)
    c = (rng.random_sample((1, m, 3)) - 0.5).astype(np.float32)
    om = orc.approx_match(a, c)
    gm = R.approx_match(torch.from_numpy(a).cuda(), torch.from_numpy(c).cuda()).cpu().numpy()
    err = np.abs(gm - om)
    strict = err <= 1e-6 + 1e-4 * np.abs(om)
    oc = orc.match_cost(a, c, om)
    gc = R.match_cost(torch.from_numpy(a).cuda(), torch.from_numpy(c).cuda(), torch.from_numpy(gm).cuda()).cpu().numpy()
    print(n, m, "max abs err %.3e" % err.max(), "outside strict", int((~strict).sum()), "of", strict.size, "cost rel err %.2e" % abs(gc[0] / oc[0] - 1),
          "row sum err %.2e col sum err %.2e" % (np.abs(gm.sum(1) - om.sum(1)).max(), np.abs(gm.sum(2) - om.sum(2)).max()))
