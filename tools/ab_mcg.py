#!/usr/bin/env python3
"""Same-device A/B of match_cost_grad builds (tools/build_variant.py) at C4 (32 x 2048 x 2048): hipEvent time of the
gradient kernel, achieved HBM rate on its one pass over `match` (512 MiB), and the gradients against the product's.
usage: python tools/ab_mcg.py TAG [TAG ...]   ('base' = the product)
With AB_MCG_SHAPES=1 in the environment also a spread of other shapes (random non-negative `match`), kernel time per shape."""
import os, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CODE = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from rfnet_amd import _lib, _raw as R
rng = np.random.RandomState(100)
u = torch.from_numpy((rng.random_sample((32, 2048, 3)) - 0.5).astype(np.float32)).cuda()
v = torch.from_numpy((rng.random_sample((32, 2048, 3)) - 0.5).astype(np.float32)).cuda()
mt = R.approx_match(u, v)
for _ in range(3): g = R.match_cost_grad(u, v, mt)
torch.cuda.synchronize(); _lib.profile_collect(); _lib.profile_enable(True)
for _ in range(20): g = R.match_cost_grad(u, v, mt)
torch.cuda.synchronize(); _lib.profile_enable(False)
pr = {k: v_[0] / v_[1] for k, v_ in _lib.profile_collect().items()}
ms = pr.get("mc_grad", 0.0)
import os
if os.environ.get("AB_MCG_SHAPES"):
    out = []
    for (b, n, m) in [(32, 512, 512), (32, 1024, 1024), (8, 2048, 2048), (64, 2048, 2048), (32, 2048, 1024), (32, 1024, 2048), (4, 4096, 4096), (1, 16384, 16384), (32, 1028, 1000), (2, 8192, 8192)]:
        x = torch.rand(b, n, 3, device="cuda") - 0.5; y = torch.rand(b, m, 3, device="cuda") - 0.5
        w = torch.rand(b, m, n, device="cuda") ** 8 / n
        for _ in range(2): R.match_cost_grad(x, y, w)
        torch.cuda.synchronize(); _lib.profile_collect(); _lib.profile_enable(True)
        for _ in range(8): R.match_cost_grad(x, y, w)
        torch.cuda.synchronize(); _lib.profile_enable(False)
        q = _lib.profile_collect()["mc_grad"]
        out.append("%%dx%%dx%%d %%.1f us (%%.2f TB/s)" %% (b, n, m, q[0] / q[1] * 1e3, b * n * m * 4 / (q[0] / q[1] * 1e-3) / 1e12))
        del w
    print("shapes: " + " | ".join(out))
print("mc_grad %%.1f us  %%.2f TB/s of match  (kernels: %%s)  checksum %%.6f %%.6f" %% (ms * 1e3, 32 * 2048 * 2048 * 4 / (ms * 1e-3) / 1e12 if ms else 0, {k: round(x * 1e3, 1) for k, x in pr.items()}, float(g[0].double().abs().sum()), float(g[1].double().abs().sum())))
''' % ROOT
for rnd in range(2):
    for tag in (sys.argv[1:] or ["base"]):
        env = dict(os.environ)
        if tag != "base":
            env["RFOPS_LIB"] = os.path.join(ROOT, "rfnet_amd", "variants", f"librfops_{tag}.so")
        out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
        print(f"round {rnd} {tag:10s} {out.stdout.strip()} {out.stderr.strip()[-300:] if out.returncode else ''}")
