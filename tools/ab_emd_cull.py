#!/usr/bin/env python3
"""Culled vs dense sharp EMD levels: approx_match per-kernel times at several cloud sizes for library variants
(tools/build_variant.py ... -DRFA_CULL_MIN_PTS=...), same device, one process per variant.
usage: python tools/ab_emd_cull.py TAG [TAG ...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CHILD = r'''
import json, sys, numpy as np, torch
sys.path.insert(0, %r)
from rfnet_amd import _lib, _raw as R
res = {}
for (b, n) in ((32, 2048), (16, 4096), (8, 8192), (4, 16384)):
    rng = np.random.RandomState(100)
    u = torch.from_numpy((rng.random_sample((b, n, 3)) - 0.5).astype(np.float32)).cuda()
    v = torch.from_numpy((rng.random_sample((b, n, 3)) - 0.5).astype(np.float32)).cuda()
    for _ in range(2): c = R.earth_mover(u, v)
    torch.cuda.synchronize(); _lib.profile_collect(); _lib.profile_enable(True)
    for _ in range(5): c = R.earth_mover(u, v)
    torch.cuda.synchronize(); _lib.profile_enable(False)
    pr = _lib.profile_collect()
    tot = sum(v_[0] for v_ in pr.values()) / 5
    res["%%dx%%d" %% (b, n)] = {"total_ms": round(tot, 4), "cost0": float(c[0]), **{k: round(v_[0] / 5, 4) for k, v_ in pr.items() if k.startswith("am_p")}}
print(json.dumps(res))
''' % ROOT
for tag in sys.argv[1:] or ["base"]:
    env = dict(os.environ)
    if tag != "base":
        env["RFOPS_LIB"] = os.path.join(ROOT, "rfnet_amd", "variants", f"librfops_{tag}.so")
    out = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(tag, "FAILED", out.stderr[-800:])
        continue
    for k, v in json.loads(line[-1]).items():
        print(f"{tag:10s} {k:10s} {v}")
