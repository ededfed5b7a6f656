#!/usr/bin/env python3
"""approx_match -> match_cost -> match_cost_grad at C4 as ONE sequence (what a training step runs): per-kernel hipEvent times of the
two streaming kernels inside the sequence, and alone in a loop.  The 512 MiB `match` tensor is twice the 256 MB memory-side
cache: in which order a kernel walks it decides how much of the previous kernel's tail it still finds there.
usage: [RFOPS_LIB=...] python tools/experiments/match_order.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from rfnet_amd import _lib, _raw as R  # noqa: E402

rng = np.random.RandomState(100)
u = torch.from_numpy((rng.random_sample((32, 2048, 3)) - 0.5).astype(np.float32)).cuda()
v = torch.from_numpy((rng.random_sample((32, 2048, 3)) - 0.5).astype(np.float32)).cuda()


def prof(fn, it):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    _lib.profile_collect()
    _lib.profile_enable(True)
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    return {k: round(x[0] / x[1] * 1e3, 1) for k, x in _lib.profile_collect().items() if k in ("mc_partial", "mc_grad", "am_match")}


def seq():
    mt = R.approx_match(u, v)
    c = R.match_cost(u, v, mt)
    g = R.match_cost_grad(u, v, mt)
    return c, g


mt = R.approx_match(u, v)
print("sequence approx_match -> match_cost -> match_cost_grad (us):", prof(seq, 8))
print("match_cost alone in a loop:", prof(lambda: R.match_cost(u, v, mt), 10), " match_cost_grad alone:", prof(lambda: R.match_cost_grad(u, v, mt), 10))
c, g = seq()
print("checksums %.6f %.6f %.6f" % (float(c.double().sum()), float(g[0].double().abs().sum()), float(g[1].double().abs().sum())))
