#!/usr/bin/env python3
"""Same-device A/B of library variants on rf_chamfer_step at C2 over point distributions (randn, uniform, resample_pcd-style
duplicates, every point repeated 5x): wall us per step and per-kernel times; outputs checked against the dense sweep.
usage: python tools/ab_dist.py TAG [TAG ...]   ('base' = the product)"""
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CHILD = r'''
import json, sys, time, numpy as np, torch
sys.path.insert(0, %r)
from rfnet_amd import _lib, _raw as R
B, N, M = 32, 2048, 16384
rng = np.random.RandomState(300)
def t(x): return torch.from_numpy(np.ascontiguousarray(x.astype(np.float32))).cuda()
cases = {"randn": (rng.randn(B, N, 3), rng.randn(B, M, 3)), "uniform": (rng.rand(B, N, 3) - 0.5, rng.rand(B, M, 3) - 0.5)}
uniq = N // 3
base = rng.rand(B, uniq, 3) - 0.5
idx = np.concatenate([np.stack([rng.permutation(uniq) for _ in range(B)]), rng.randint(0, uniq, (B, N - uniq))], 1)
cases["resample_pcd"] = (np.take_along_axis(base, idx[..., None], 1), rng.rand(B, M, 3) - 0.5)
b5 = rng.rand(B, M // 5 + 1, 3) - 0.5
cases["x5 both"] = (np.repeat(rng.rand(B, N // 5 + 1, 3) - 0.5, 5, 1)[:, :N], np.repeat(b5, 5, 1)[:, :M])
g1 = torch.ones(B, N, device="cuda"); g2 = torch.ones(B, M, device="cuda")
res = {}
for name, (a, c) in cases.items():
    a, c = t(a), t(c)
    plan = R.ChamferStep(B, N, M, "cuda")
    out = plan(a, c, g1, g2)
    ref = R.nn_distance(a, c, mode="dense")
    ok = all(torch.equal(x, y) for x, y in zip(ref, out[:4]))
    for _ in range(5): plan(a, c, g1, g2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): plan(a, c, g1, g2)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 50 * 1e6
    _lib.profile_collect(); _lib.profile_enable(True)
    for _ in range(20): plan(a, c, g1, g2)
    torch.cuda.synchronize(); _lib.profile_enable(False)
    pr = _lib.profile_collect()
    res[name] = {"ok": ok, "wall_us": round(wall, 1), **{k: round(v[0] / v[1] * 1e3, 1) for k, v in pr.items()}}
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
        print(f"{tag:10s} {k:14s} {v}")
