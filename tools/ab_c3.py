#!/usr/bin/env python3
"""BASELINE.json configs[2] (32 x 16384 -> 1024 samples, r = 0.1, nsample = 32): the four ops one after another against
rf_sample_and_group (with and without the auxiliary stream), same device; wall time per pass between fences."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from rfnet_amd import _lib, _raw as R


def wall(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


rng = np.random.RandomState(100)
xyz = torch.from_numpy(rng.random_sample((32, 16384, 3)).astype(np.float32)).cuda()


def chain(form):
    fi = R.farthest_point_sample(1024, xyz)
    nx = R.gather_point(xyz, fi)
    gi, cnt = R.query_ball_point(0.1, 32, xyz, nx, form=form)
    return R.group_point(xyz, gi)


aux = torch.cuda.Stream()
print(f"four ops, scan ball query      {wall(lambda: chain('scan')):.4f} ms per pass")
print(f"four ops, boxed ball query     {wall(lambda: chain('boxes')):.4f} ms per pass")
print(f"rf_sample_and_group            {wall(lambda: R.sample_and_group(1024, 0.1, 32, xyz)):.4f} ms per pass")
print(f"rf_sample_and_group + aux      {wall(lambda: R.sample_and_group(1024, 0.1, 32, xyz, aux_stream=aux)):.4f} ms per pass")
_lib.profile_collect(); _lib.profile_enable(True)
for _ in range(10):
    R.sample_and_group(1024, 0.1, 32, xyz)
torch.cuda.synchronize(); _lib.profile_enable(False)
print("kernels per pass:", {k: round(v[0] / 10, 4) for k, v in _lib.profile_collect().items()})
fps0 = wall(lambda: R.farthest_point_sample(1024, xyz))
print(f"farthest_point_sample alone    {fps0:.4f} ms")
