import sys, numpy as np, torch
sys.path.insert(0, '.')
from rfnet_amd import _raw as R
from oracle.oracle import Oracle
orc = Oracle()
rng = np.random.RandomState(31)
B, N = 21, 2000  # 8.4e7 pairs: past the expansion's threshold; the oracle on the first and last sample
a = (rng.random_sample((B, N, 3)) - 0.5).astype(np.float32)
c = (rng.random_sample((B, N, 3)) - 0.5).astype(np.float32)
pick = [0, B - 1]
cu = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
for scale in (1.0, 2.2, 4.0):
    aa, cc = a * np.float32(scale), c * np.float32(scale)
    om = orc.approx_match(aa[pick], cc[pick])
    got = R.approx_match(cu(aa), cu(cc))[pick].cpu().numpy()
    bad = np.abs(got - om) > 1e-6 + 1e-4 * np.abs(om)
    print("scale", scale, "outside", int(bad.sum()), "max abs", float(np.abs(got - om).max()))
a2 = a.copy(); a2[1, 7, 2] = np.nan
got = R.approx_match(cu(a2), cu(c))[pick].cpu().numpy()
om = orc.approx_match(a[pick], c[pick])
bad = np.abs(got - om) > 1e-6 + 1e-4 * np.abs(om)
print("nan neighbour: outside", int(bad.sum()), float(np.abs(got - om).max()))
