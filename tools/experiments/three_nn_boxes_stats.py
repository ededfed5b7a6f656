#!/usr/bin/env python3
"""three_nn_boxes_kernel: superblock visits and 16-record block scans per wave (mean / max), from a build with -DTB_STATS that
writes the two counters into idx[..., 1] and idx[..., 2] (python tools/build_variant.py tbstats -DTB_STATS).
usage: RFOPS_LIB=rfnet_amd/variants/librfops_tbstats.so python tools/experiments/three_nn_boxes_stats.py
Measured (uniform clouds): 32 x 16384 x 1024: 4.8 visits, 9.8 of 64 blocks per wave (max 25); 32 x 16384 x 16384: 22.7 visits, 35.9 of 1024 blocks."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from rfnet_amd import _raw as R
for (b, n, m) in [(32, 16384, 1024), (32, 16384, 16384), (32, 4096, 1024)]:
    rng = np.random.RandomState(1)
    u = torch.from_numpy(rng.random_sample((b, n, 3)).astype(np.float32)).cuda()
    k = torch.from_numpy(rng.random_sample((b, m, 3)).astype(np.float32)).cuda()
    d, i = R.three_nn(u, k, form="boxes")
    print(b, n, m, "sb visits per wave mean/max", i[..., 1].float().mean().item(), i[..., 1].max().item(), "block scans mean/max", i[..., 2].float().mean().item(), i[..., 2].max().item(), "of", m // 16)
