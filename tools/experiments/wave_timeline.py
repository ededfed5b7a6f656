#!/usr/bin/env python3
"""When does every wave of one nnp_sweep launch start and end?  Needs a library built with -DRFP_SG_STAMPS=2
(tools/build_variant.py tl -DRFP_SG_STAMPS=2; RFOPS_LIB=...): that build writes each wave's s_memrealtime at entry and at
its output store (100 MHz) INTO the outputs (idx = start, dist bits = end) instead of the results.
usage: RFOPS_LIB=rfnet_amd/variants/librfops_tl.so python tools/experiments/wave_timeline.py [B N M]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from rfnet_amd import _raw as R  # noqa: E402

B, N, M = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 2048, 16384)
rng = np.random.RandomState(100)
a = torch.from_numpy(rng.randn(B, N, 3).astype(np.float32)).cuda()
c = torch.from_numpy(rng.randn(B, M, 3).astype(np.float32)).cuda()
for _ in range(3):
    out = R.nn_distance(a, c, mode="culled")
torch.cuda.synchronize()
t0 = None
res = {}
for d, (dist, idx) in enumerate(((out[0], out[1]), (out[2], out[3]))):
    st = idx.cpu().numpy().astype(np.uint32).astype(np.int64).ravel()
    en = dist.cpu().numpy().view(np.uint32).astype(np.int64).ravel()
    waves = np.unique(np.stack([st, en], 1), axis=0)  # one (start, end) pair per wave (64 or 16 queries share it)
    res[d] = waves
t0 = min(w[:, 0].min() for w in res.values())
tend = max(w[:, 1].max() for w in res.values())
print(f"launch span (first wave start -> last output store): {(tend - t0) / 100:.1f} us")
for d, w in res.items():
    s, e = (w[:, 0] - t0) / 100.0, (w[:, 1] - t0) / 100.0
    dur = e - s
    print(f"dir{d}: {len(w)} waves  duration mean {dur.mean():.1f} us  p50 {np.percentile(dur, 50):.1f}  p90 {np.percentile(dur, 90):.1f}  "
          f"p99 {np.percentile(dur, 99):.1f}  max {dur.max():.1f} | start p50 {np.percentile(s, 50):.1f}  p90 {np.percentile(s, 90):.1f}  max {s.max():.1f} | "
          f"end p50 {np.percentile(e, 50):.1f}  p90 {np.percentile(e, 90):.1f}  p99 {np.percentile(e, 99):.1f}  max {e.max():.1f}")
    order = np.argsort(-e)[:8]
    print("   last to end (start, duration):", "  ".join(f"({s[i]:.1f}, {dur[i]:.1f})" for i in order))
    # resident waves over time
    ts = np.arange(0, (tend - t0) / 100.0, 2.0)
    print("   resident at t (us):", "  ".join(f"{t:.0f}:{int(((s <= t) & (e > t)).sum())}" for t in ts))
