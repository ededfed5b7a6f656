#!/usr/bin/env python3
"""approx_match + match_cost at C4 a few times, for profilers."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from rfnet_amd import _raw as R
rng = np.random.RandomState(100)
u = torch.from_numpy((rng.random_sample((32, 2048, 3)) - 0.5).astype(np.float32)).cuda()
v = torch.from_numpy((rng.random_sample((32, 2048, 3)) - 0.5).astype(np.float32)).cuda()
for _ in range(6):
    R.match_cost(u, v, R.approx_match(u, v))
torch.cuda.synchronize()
