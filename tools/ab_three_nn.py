#!/usr/bin/env python3
"""three_nn: the boxed form over sorted sets against the scan, per shape -- identical outputs? and the wall time of each
(sort included; with rf_nn_sort handles of both sets in the last column).
usage: python tools/ab_three_nn.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
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


shapes = [(32, 16384, 1024, "uniform"), (32, 16384, 1024, "randn"), (32, 16384, 4096, "uniform"), (32, 16384, 16384, "uniform"),
          (32, 4096, 1024, "uniform"), (32, 2048, 512, "uniform"), (32, 1024, 256, "uniform"), (8, 16384, 1024, "uniform"),
          (1, 16384, 2048, "uniform"), (32, 16384, 64, "uniform"), (4, 65536, 4096, "uniform"), (32, 16384, 1024, "lattice"),
          (32, 3000, 700, "sphere")]
for (b, n, m, kind) in shapes:
    rng = np.random.RandomState(n + m)
    if kind == "uniform":
        u, k = rng.random_sample((b, n, 3)), rng.random_sample((b, m, 3))
    elif kind == "randn":
        u, k = rng.standard_normal((b, n, 3)), rng.standard_normal((b, m, 3))
    elif kind == "lattice":  # many exact ties
        u, k = rng.randint(0, 12, (b, n, 3)) / 8.0, rng.randint(0, 12, (b, m, 3)) / 8.0
    else:
        u, k = rng.standard_normal((b, n, 3)), rng.standard_normal((b, m, 3))
        u /= np.linalg.norm(u, axis=2, keepdims=True)
        k /= np.linalg.norm(k, axis=2, keepdims=True)
    u, k = torch.from_numpy(u.astype(np.float32)).cuda(), torch.from_numpy(k.astype(np.float32)).cuda()
    d0, i0 = R.three_nn(u, k, form="scan")
    d1, i1 = R.three_nn(u, k, form="boxes")
    same = bool(torch.equal(d0, d1) and torch.equal(i0, i1))
    if not same:
        bad = ((d0 != d1) | (i0 != i1)).any(dim=2)
        w = bad.nonzero()[0].tolist()
        print("  MISMATCH rows", int(bad.sum()), "first", w, d0[w[0], w[1]].tolist(), i0[w[0], w[1]].tolist(), d1[w[0], w[1]].tolist(), i1[w[0], w[1]].tolist())
    ts = timed(lambda: R.three_nn(u, k, form="scan"))
    tb = timed(lambda: R.three_nn(u, k, form="boxes"))
    h1, h2 = R.nn_sort(u), R.nn_sort(k)
    th = timed(lambda: R.three_nn(u, k, form="boxes", sorted1=h1.buf, sorted2=h2.buf))
    print(f"{b}x{n}x{m} {kind:8s} identical {same}  scan {ts:.4f} ms  boxes {tb:.4f} ms  on handles {th:.4f} ms", flush=True)
