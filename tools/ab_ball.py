#!/usr/bin/env python3
"""query_ball_point at C3 (32 x 16384 dataset, 1024 FPS queries, r = 0.1, nsample = 32): the scan kernel against the boxed
kernel (with its own sort, and on a ready rf_nn_sort handle), same device; plus other radii.  Kernel times from the
library's event brackets."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from rfnet_amd import _lib, _raw as R


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


rng = np.random.RandomState(100)
xyz = torch.from_numpy(rng.random_sample((32, 16384, 3)).astype(np.float32)).cuda()
fi = R.farthest_point_sample(1024, xyz)
q = R.gather_point(xyz, fi)
h = R.nn_sort(xyz)
for r in (0.1, 0.02, 0.2, 0.4, 1.0):
    s = R.query_ball_point(r, 32, xyz, q, form="scan")
    g = R.query_ball_point(r, 32, xyz, q, form="boxes")
    same = bool(torch.equal(s[0], g[0]) and torch.equal(s[1], g[1]))
    ts = timed(lambda: R.query_ball_point(r, 32, xyz, q, form="scan"))
    tb = timed(lambda: R.query_ball_point(r, 32, xyz, q, form="boxes"))
    th = timed(lambda: R.query_ball_point(r, 32, xyz, q, form="boxes", sorted1=h.buf))
    print(f"r={r:<5} scan {ts:.4f} ms   boxes {tb:.4f} ms   boxes on a handle {th:.4f} ms   identical={same}  mean cnt {float(s[1].float().mean()):.2f}")
_lib.profile_collect(); _lib.profile_enable(True)
for _ in range(10):
    R.query_ball_point(0.1, 32, xyz, q, form="boxes")
torch.cuda.synchronize(); _lib.profile_enable(False)
print("kernels at r=0.1:", {k: round(v[0] / v[1], 4) for k, v in _lib.profile_collect().items()})
