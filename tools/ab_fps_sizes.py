#!/usr/bin/env python3
"""fps_reg (unsorted cloud) against nnp_sort + fps_sorted over cloud sizes and sample counts: where does the sorted form pay?
kernel times (ms) by the library's event brackets, uniform clouds, batch 32.   usage: python tools/ab_fps_sizes.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from rfnet_amd import _lib, _raw as R

def kern(fn, it=4):
    for _ in range(2): fn()
    torch.cuda.synchronize(); _lib.profile_collect(); _lib.profile_enable(True)
    for _ in range(it): fn()
    torch.cuda.synchronize(); _lib.profile_enable(False)
    return {k: v[0] / it for k, v in _lib.profile_collect().items()}

rng = np.random.RandomState(7)
for n in (1500, 2048, 3000, 4096, 6000, 8192, 12000, 16384):
    x = torch.from_numpy(rng.random_sample((32, n, 3)).astype(np.float32)).cuda()
    row = []
    for m in (32, 64, 128, 256, 512, 1024, 2048):
        if m > n: continue
        a = kern(lambda: R.farthest_point_sample_reg(m, x))
        b = kern(lambda: R.farthest_point_sample_sorted(m, x))
        same = bool(torch.equal(R.farthest_point_sample_reg(m, x), R.farthest_point_sample_sorted(m, x)))
        ra = sum(a.values()); rb = sum(b.values())
        row.append("m=%d: %.3f vs %.3f%s%s" % (m, ra, rb, "" if same else " DIFFERENT", " <" if rb < ra else ""))
    print("n=%5d  " % n + " | ".join(row), flush=True)
