import sys, numpy as np, torch
sys.path.insert(0, '.')
from rfnet_amd import _raw as R
from oracle.oracle import Oracle
orc = Oracle()
rng = np.random.RandomState(31)
a = (rng.random_sample((2, 1500, 3)) - 0.5).astype(np.float32)
c = (rng.random_sample((2, 1500, 3)) - 0.5).astype(np.float32)
cu = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
for scale in (1.0, 2.2, 4.0):
    aa, cc = a * np.float32(scale), c * np.float32(scale)
    om = orc.approx_match(aa, cc)
    got = R.approx_match(cu(aa), cu(cc)).cpu().numpy()
    bad = np.abs(got - om) > 1e-6 + 1e-4 * np.abs(om)
    print("scale", scale, "outside", int(bad.sum()), "max abs", float(np.abs(got - om).max()))
a2 = a.copy(); a2[1, 7, 2] = np.nan
got = R.approx_match(cu(a2), cu(c)).cpu().numpy()
om = orc.approx_match(a[:1], c[:1])
bad = np.abs(got[:1] - om) > 1e-6 + 1e-4 * np.abs(om)
print("nan neighbour: outside", int(bad.sum()), float(np.abs(got[:1] - om).max()))
