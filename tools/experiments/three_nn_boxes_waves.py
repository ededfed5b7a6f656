#!/usr/bin/env python3
"""three_nn_boxes_kernel on handles at four shapes, for builds with 1 / 2 / 4 / 8 waves per workgroup
(python tools/build_variant.py tbwN -DRFI_TB_WAVES=N; RFOPS_LIB selects the build).  Measured: within 4 % of each other; 4 kept."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from rfnet_amd import _raw as R
def timed(fn, it=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
out=[]
for (b, n, m) in [(32, 16384, 1024), (32, 16384, 16384), (32, 4096, 1024), (8, 16384, 1024)]:
    rng = np.random.RandomState(1)
    u = torch.from_numpy(rng.random_sample((b, n, 3)).astype(np.float32)).cuda()
    k = torch.from_numpy(rng.random_sample((b, m, 3)).astype(np.float32)).cuda()
    h1, h2 = R.nn_sort(u), R.nn_sort(k)
    out.append("%dx%dx%d %.4f" % (b, n, m, timed(lambda: R.three_nn(u, k, form="boxes", sorted1=h1.buf, sorted2=h2.buf))))
print(os.environ.get("RFOPS_LIB", "base")[-22:], " | ".join(out))
