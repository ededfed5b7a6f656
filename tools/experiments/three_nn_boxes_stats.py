import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from rfnet_amd import _raw as R
for (b, n, m) in [(32, 16384, 1024), (32, 16384, 16384), (32, 4096, 1024)]:
    rng = np.random.RandomState(1)
    u = torch.from_numpy(rng.random_sample((b, n, 3)).astype(np.float32)).cuda()
    k = torch.from_numpy(rng.random_sample((b, m, 3)).astype(np.float32)).cuda()
    d, i = R.three_nn(u, k, form="boxes")
    print(b, n, m, "sb visits per wave mean/max", i[..., 1].float().mean().item(), i[..., 1].max().item(), "block scans mean/max", i[..., 2].float().mean().item(), i[..., 2].max().item(), "of", m // 16)
