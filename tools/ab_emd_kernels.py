#!/usr/bin/env python3
"""Same-device A/B of library builds on the EMD half at C4 (32 x 2048 x 2048, U(-0.5,0.5) seed 100): approx_match wall time,
its kernels by the library's own event brackets, match_cost, the fused earth_mover, and the distance of `match` and cost
from the product build's.  usage: python tools/ab_emd_kernels.py TAG [TAG ...]   ('base' = the product; others =
rfnet_amd/variants/librfops_TAG.so)"""
import os, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CODE = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from rfnet_amd import _lib, _raw as R
rng = np.random.RandomState(100)
def t(fn, it):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
u = torch.from_numpy((rng.random_sample((32, 2048, 3)) - 0.5).astype(np.float32)).cuda()
v = torch.from_numpy((rng.random_sample((32, 2048, 3)) - 0.5).astype(np.float32)).cuda()
am = t(lambda: R.approx_match(u, v), 20)
mt = R.approx_match(u, v)
both = t(lambda: R.match_cost(u, v, R.approx_match(u, v)), 20)
fused = t(lambda: R.earth_mover(u, v), 20)
_lib.profile_collect(); _lib.profile_enable(True)
for _ in range(10): R.approx_match(u, v)
torch.cuda.synchronize(); _lib.profile_enable(False)
pr = _lib.profile_collect()
calls = 10
ks = {k: round(x[0] / calls, 4) for k, x in pr.items()}
cost = R.match_cost(u, v, mt)
print("approx_match %%.4f ms  +match_cost %%.4f ms (%%.0f calls/s)  earth_mover %%.4f ms  kernels/call %%s sum %%.4f  cost_sum %%.6f  match_abs_sum %%.6f" %% (
    am, both, 1e3 / both, fused, ks, sum(ks.values()), float(cost.double().sum()), float(mt.double().abs().sum())))
''' % ROOT
for rnd in range(2):
    for tag in (sys.argv[1:] or ["base"]):
        env = dict(os.environ)
        if tag != "base":
            env["RFOPS_LIB"] = os.path.join(ROOT, "rfnet_amd", "variants", f"librfops_{tag}.so")
        out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
        print(f"round {rnd} {tag:8s} {out.stdout.strip()} {out.stderr.strip()[-400:] if out.returncode else ''}", flush=True)
