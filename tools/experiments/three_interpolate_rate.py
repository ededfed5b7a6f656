#!/usr/bin/env python3
"""three_interpolate / its gradient at feature-propagation shapes: wall time against the bytes that must move
(out (b,n,c) written once; points (b,m,c) L2-resident; idx + weight 24 B per unknown point)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from rfnet_amd import _raw as R


def timed(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


rng = np.random.RandomState(0)
for (b, n, m, c) in [(32, 16384, 1024, 128), (32, 16384, 1024, 64), (32, 16384, 1024, 3), (32, 4096, 1024, 256), (32, 1024, 256, 512), (8, 16384, 4096, 32), (32, 16384, 1024, 13)]:
    u = torch.from_numpy(rng.random_sample((b, n, 3)).astype(np.float32)).cuda()
    k = torch.from_numpy(rng.random_sample((b, m, 3)).astype(np.float32)).cuda()
    pts = torch.from_numpy(rng.standard_normal((b, m, c)).astype(np.float32)).cuda()
    d, i = R.three_nn(u, k)
    w = torch.rand(b, n, 3, device="cuda")
    go = torch.randn(b, n, c, device="cuda")
    tf = timed(lambda: R.three_interpolate(pts, i, w))
    tg = timed(lambda: R.three_interpolate_grad(pts, i, w, go))
    by = 4.0 * b * n * c + 24.0 * b * n + 4.0 * b * m * c
    print(f"{b}x{n}x{m} c={c}: fwd {tf:.4f} ms = {by / tf / 1e6:.0f} GB/s   grad {tg:.4f} ms = {by / tg / 1e6:.0f} GB/s", flush=True)
