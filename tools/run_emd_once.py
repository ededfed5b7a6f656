#!/usr/bin/env python3
"""approx_match + match_cost at C4 a few times, for profilers.  usage: python tools/run_emd_once.py [50 | big]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from rfnet_amd import _raw as R
rng = np.random.RandomState(100)
u = torch.from_numpy((rng.random_sample((32, 2048, 3)) - 0.5).astype(np.float32)).cuda()
v = torch.from_numpy((rng.random_sample((32, 2048, 3)) - 0.5).astype(np.float32)).cuda()
levels = None
if len(sys.argv) > 1 and sys.argv[1] == "big":  # the evaluation size of the cost-only form: 4 x 16384^2
    u = torch.from_numpy((rng.random_sample((4, 16384, 3)) - 0.5).astype(np.float32)).cuda()
    v = torch.from_numpy((rng.random_sample((4, 16384, 3)) - 0.5).astype(np.float32)).cuda()
    for _ in range(4):
        R.earth_mover(u, v)
    torch.cuda.synchronize()
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "50":  # BASELINE configs[3]: the ten reference levels five times each
    levels = np.repeat(np.asarray([-4.0 ** j for j in range(7, -2, -1)] + [0.0], np.float32), 5).tolist()
for _ in range(6):
    R.match_cost(u, v, R.approx_match(u, v, levels=levels))
torch.cuda.synchronize()
