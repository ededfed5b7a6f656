#!/usr/bin/env python3
"""Same-device A/B of library builds on approx_match / earth_mover over a spread of shapes: per shape and build the wall time of
approx_match and of earth_mover.  Clouds U(-0.5, 0.5) * scale (scale 4: every level sixteen-fold sharper).
usage: python tools/ab_emd_sizes.py TAG [TAG ...]   ('base' = the product; others = rfnet_amd/variants/librfops_TAG.so)"""
import os, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CODE = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from rfnet_amd import _raw as R
def t(fn, it):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
shapes = [(32, 512, 512, 1.0), (32, 1024, 1024, 1.0), (8, 2048, 2048, 1.0), (16, 2048, 2048, 1.0), (24, 2048, 2048, 1.0),
          (32, 2048, 2048, 1.0), (64, 2048, 2048, 1.0), (32, 2048, 1024, 1.0), (4, 4096, 4096, 1.0), (8, 4096, 4096, 1.0),
          (2, 8192, 8192, 1.0), (1, 16384, 16384, 1.0), (32, 1024, 1024, 4.0), (32, 2048, 2048, 4.0)]
out = []
for (b, n, m, sc) in shapes:
    rng = np.random.RandomState(b * 7 + n)
    u = torch.from_numpy(((rng.random_sample((b, n, 3)) - 0.5) * sc).astype(np.float32)).cuda()
    v = torch.from_numpy(((rng.random_sample((b, m, 3)) - 0.5) * sc).astype(np.float32)).cuda()
    it = 10 if b * n * m < 3e8 else 4
    am = t(lambda: R.approx_match(u, v), it)
    em = t(lambda: R.earth_mover(u, v), it)
    out.append("%%dx%%dx%%d*%%g: %%.3f / %%.3f" %% (b, n, m, sc, am, em))
print(" | ".join(out))
''' % ROOT
for rnd in range(2):
    for tag in (sys.argv[1:] or ["base"]):
        env = dict(os.environ)
        if tag != "base":
            env["RFOPS_LIB"] = os.path.join(ROOT, "rfnet_amd", "variants", f"librfops_{tag}.so")
        out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
        print(f"round {rnd} {tag:8s} {out.stdout.strip()} {out.stderr.strip()[-400:] if out.returncode else ''}", flush=True)
