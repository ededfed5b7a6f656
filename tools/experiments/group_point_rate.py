#!/usr/bin/env python3
"""group_point / its gradient with feature channels: wall time against the bytes that must move (out (b,m,ns,c) written once)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from rfnet_amd import _raw as R

def timed(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it

rng = np.random.RandomState(0)
out = []
for (b, n, m, ns, c) in [(32, 16384, 1024, 32, 64), (32, 16384, 1024, 32, 3), (32, 4096, 512, 32, 128), (32, 16384, 1024, 32, 6), (8, 1024, 256, 64, 256)]:
    pts = torch.randn(b, n, c, device="cuda")
    idx = torch.from_numpy(rng.randint(0, n, (b, m, ns)).astype(np.int32)).cuda()
    go = torch.randn(b, m, ns, c, device="cuda")
    tf = timed(lambda: R.group_point(pts, idx))
    tg = timed(lambda: R.group_point_grad(pts, idx, go))
    by = 4.0 * b * m * ns * c
    out.append(f"{b}x{n}->{m}x{ns} c={c}: fwd {tf:.4f} ms = {by / tf / 1e6:.0f} GB/s  grad {tg:.4f} ms = {by / tg / 1e6:.0f} GB/s")
print(os.environ.get("RFOPS_LIB", "base")[-20:], " | ".join(out))
