#!/usr/bin/env python3
"""The interpolation ops a few times at feature-propagation sizes (32 x 16384 unknown, 1024 known sampled points, c = 128), for
profilers: three_nn boxed (with its sort) and scan, three_interpolate rows form, its gradient as LDS tiles, group_point rows."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from rfnet_amd import _raw as R
rng = np.random.RandomState(100)
xyz = torch.from_numpy(rng.random_sample((32, 16384, 3)).astype(np.float32)).cuda()
q = R.gather_point(xyz, R.farthest_point_sample(1024, xyz))
pts = torch.randn(32, 1024, 128, device="cuda")
w = torch.rand(32, 16384, 3, device="cuda")
go = torch.randn(32, 16384, 128, device="cuda")
feat = torch.randn(32, 16384, 64, device="cuda")
gi, _ = R.query_ball_point(0.1, 32, xyz, q)
for _ in range(5):
    d, i = R.three_nn(xyz, q, form="boxes")
    R.three_nn(xyz, q, form="scan")
    R.three_interpolate(pts, i, w)
    R.three_interpolate_grad(pts, i, w, go)
    R.group_point(feat, gi)
torch.cuda.synchronize()
