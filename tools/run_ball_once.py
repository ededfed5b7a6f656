#!/usr/bin/env python3
"""C3's query_ball_point a few times (boxed form with its own sort; then the scan form), for profilers."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from rfnet_amd import _raw as R
rng = np.random.RandomState(100)
xyz = torch.from_numpy(rng.random_sample((32, 16384, 3)).astype(np.float32)).cuda()
q = R.gather_point(xyz, R.farthest_point_sample(1024, xyz))
r = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
for _ in range(5):
    R.query_ball_point(r, 32, xyz, q, form="boxes")
    R.query_ball_point(r, 32, xyz, q, form="scan")
torch.cuda.synchronize()
