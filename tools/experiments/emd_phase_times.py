import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from rfnet_amd import _lib, _raw as R
rng = np.random.RandomState(100)
u = torch.from_numpy((rng.random_sample((32, 2048, 3)) - 0.5).astype(np.float32)).cuda()
v = torch.from_numpy((rng.random_sample((32, 2048, 3)) - 0.5).astype(np.float32)).cuda()
for _ in range(3): m = R.approx_match(u, v)
torch.cuda.synchronize(); _lib.profile_collect(); _lib.profile_enable(True)
for _ in range(10): m = R.approx_match(u, v)
torch.cuda.synchronize(); _lib.profile_enable(False)
pr = _lib.profile_collect()
print({k: round(v_[0] / 10, 4) for k, v_ in pr.items()}, "total", round(sum(v_[0] for v_ in pr.values()) / 10, 4), float(m.sum()))
