"""Per-kernel times (the library's own event brackets) of approx_match at C4 size for clouds inside the unit cube (expanded) and
four times larger (the expansion refused on the device: direct sums inside the same launches)."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from rfnet_amd import _lib, _raw as R
rng = np.random.RandomState(100)
for sc in (1.0, 4.0):
    u = torch.from_numpy(((rng.random_sample((32, 2048, 3)) - 0.5) * sc).astype(np.float32)).cuda()
    v = torch.from_numpy(((rng.random_sample((32, 2048, 3)) - 0.5) * sc).astype(np.float32)).cuda()
    for _ in range(3): R.approx_match(u, v)
    torch.cuda.synchronize()
    _lib.profile_collect(); _lib.profile_enable(True)
    for _ in range(10): R.approx_match(u, v)
    torch.cuda.synchronize(); _lib.profile_enable(False)
    pr = _lib.profile_collect()
    print("scale", sc, {k: (round(x[0] / 10, 4), x[1] // 10) for k, x in pr.items()})
