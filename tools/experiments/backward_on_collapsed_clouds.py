#!/usr/bin/env python3
"""The Chamfer backward's LDS sums on COLLAPSED clouds (thousands of sources on one destination: the untrained RFNet's output):
rf_chamfer_step and rf_nn_distance_grad per build.  usage: RFOPS_LIB=... python tools/experiments/backward_on_collapsed_clouds.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from rfnet_amd import _lib, _raw as R

def timed(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3

rng = np.random.RandomState(3)
B, N, M = 32, 2048, 16384
out = []
for spots in (0, 2000, 120, 8, 1):
    a = rng.randn(B, N, 3).astype(np.float32)
    if spots:
        c = rng.randn(B, spots, 3)[:, rng.randint(0, spots, M)] + 1e-4 * rng.randn(B, M, 3)
    else:
        c = rng.randn(B, M, 3)
    a, c = torch.from_numpy(a).cuda(), torch.from_numpy(c.astype(np.float32)).cuda()
    g1, g2 = torch.ones(B, N, device="cuda"), torch.ones(B, M, device="cuda")
    plan = R.ChamferStep(B, N, M, "cuda")
    o = plan(a, c, g1, g2)
    ts = timed(lambda: plan(a, c, g1, g2))
    _lib.profile_collect(); _lib.profile_enable(True)
    for _ in range(20): plan(a, c, g1, g2)
    torch.cuda.synchronize(); _lib.profile_enable(False)
    pr = {k: v[0] / v[1] * 1e3 for k, v in _lib.profile_collect().items()}
    tg = timed(lambda: R.nn_distance_grad(a, c, g1, o[1], g2, o[3]))
    # and the other way round (the collapsed cloud as the small set's target is the same; as SOURCE set of few destinations:)
    out.append(f"spots={spots or 'randn'}: step {ts:.1f} us (grad_sorted {pr.get('nnp_grad_sorted', 0):.1f}) nn_grad {tg:.1f} us")
print(os.environ.get("RFOPS_LIB", "base")[-20:], " | ".join(out))
