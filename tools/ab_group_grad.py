"""group_point_grad and three_interpolate_grad: the sorted-slots route (scatter_rows.hip, rf_*_grad_ws) against the atomic / inline
route of the same library, same process, hipEvent-timed, with the routes' kernels.  usage: python tools/ab_group_grad.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rfnet_amd import _raw as R, _lib

def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def kernels(fn, reps=10):
    _lib.profile_collect(); _lib.profile_enable(True)
    for _ in range(reps): fn()
    torch.cuda.synchronize(); _lib.profile_enable(False)
    return {k: round(v[0] / reps * 1e3, 1) for k, v in _lib.profile_collect().items()}

rng = np.random.RandomState(100)
print("group_point_grad: b, n, m, nsample, c")
for b, n, m, ns, c in [(32, 16384, 1024, 32, 64), (32, 16384, 1024, 32, 3), (32, 16384, 1024, 32, 128), (32, 16384, 16384, 1, 3), (32, 3000, 64, 32, 64),
                       (8, 16384, 1024, 32, 64), (32, 4096, 512, 16, 64), (4, 16384, 1024, 32, 16), (2, 16384, 1024, 32, 16)]:
    xyz = torch.from_numpy(rng.random_sample((b, n, 3)).astype(np.float32)).cuda()
    if ns > 1:
        q = R.gather_point(xyz, R.farthest_point_sample(m, xyz))
        idx, _ = R.query_ball_point(0.1, ns, xyz, q)
    else:
        idx = torch.from_numpy(rng.randint(0, n, size=(b, m, 1)).astype(np.int32)).cuda()
    pts = torch.zeros(b, n, c, device="cuda")
    go = torch.randn(b, m, ns, c, device="cuda")
    ws = _lib.lib.rf_grouppoint_grad_workspace_bytes(b, n, c, m, ns)
    ta = timed(lambda: R.group_point_grad(pts, idx, go, form="atomic"))
    tw = timed(lambda: R.group_point_grad(pts, idx, go))
    data = 4.0 * b * (m * ns * c + n * c) + 12.0 * b * m * ns
    print(f"  {b:3d} {n:6d} {m:6d} {ns:3d} {c:4d}  atomic {ta*1e3:8.1f} us   auto {tw*1e3:8.1f} us ({'sorted slots' if ws else 'atomic: below the threshold'})"
          f"   data {data/1e6:7.1f} MB -> {data/(tw*1e-3)/1e9:7.0f} GB/s   {kernels(lambda: R.group_point_grad(pts, idx, go))}")
print("three_interpolate_grad: b, n (unknown), m (known), c")
for b, n, m, c in [(32, 16384, 4096, 64), (32, 16384, 16384, 64), (32, 16384, 1024, 64), (8, 16384, 4096, 64), (32, 16384, 4096, 16), (32, 3000, 16384, 64)]:
    a = torch.from_numpy(rng.random_sample((b, n, 3)).astype(np.float32)).cuda()
    k = torch.from_numpy(rng.random_sample((b, m, 3)).astype(np.float32)).cuda()
    d, idx = R.three_nn(a, k)
    w = torch.rand(b, n, 3, device="cuda")
    pts = torch.zeros(b, m, c, device="cuda")
    go = torch.randn(b, n, c, device="cuda")
    ws = _lib.lib.rf_threeinterpolate_grad_workspace_bytes(b, n, c, m)
    ta = timed(lambda: R.three_interpolate_grad(pts, idx, w, go, form="inline"))
    tw = timed(lambda: R.three_interpolate_grad(pts, idx, w, go))
    data = 4.0 * b * (n * c + m * c) + 24.0 * b * n
    print(f"  {b:3d} {n:6d} {m:6d} {c:4d}  inline {ta*1e3:8.1f} us   auto {tw*1e3:8.1f} us ({'sorted slots' if ws else 'inline: tile or below the threshold'})"
          f"   data {data/1e6:7.1f} MB -> {data/(tw*1e-3)/1e9:7.0f} GB/s   {kernels(lambda: R.three_interpolate_grad(pts, idx, w, go))}")
