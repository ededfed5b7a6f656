#!/usr/bin/env python3
"""FPS over the sorted cloud (fps_sorted_kernel) against fps_reg at C3 (32 x 16384 -> 1024, U[0,1)^3 seed 100) and a few other
clouds: indices and sample coordinates bit for bit, kernel times by the library's event brackets.
usage: python tools/ab_fps_sorted.py [TAG ...]   (variant builds of the sorted form: rfnet_amd/variants/librfops_TAG.so)"""
import os, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CODE = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from rfnet_amd import _lib, _raw as R
def kern(fn, it=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); _lib.profile_collect(); _lib.profile_enable(True)
    for _ in range(it): fn()
    torch.cuda.synchronize(); _lib.profile_enable(False)
    return {k: round(v[0] / it, 4) for k, v in _lib.profile_collect().items()}
rng = np.random.RandomState(100)
s = rng.randn(32, 16384, 3)
cases = [("C3 uniform cube", rng.random_sample((32, 16384, 3)).astype(np.float32), 1024),
         ("randn", rng.randn(32, 16384, 3).astype(np.float32), 1024),
         ("sphere surface", (s / np.linalg.norm(s, axis=-1, keepdims=True)).astype(np.float32), 1024),
         ("12000 -> 512", rng.random_sample((8, 12000, 3)).astype(np.float32), 512),
         ("lattice ties", rng.randint(0, 6, size=(4, 16384, 3)).astype(np.float32), 300)]
out = []
for name, x, m in cases:
    t = torch.from_numpy(x).cuda()
    ref = R.farthest_point_sample_reg(m, t)
    got, nx = R.farthest_point_sample_sorted(m, t, with_xyz=True)
    same = bool(torch.equal(got, ref)) and bool(torch.equal(nx, R.gather_point(t, ref)))
    a = kern(lambda: R.farthest_point_sample_reg(m, t)); b = kern(lambda: R.farthest_point_sample_sorted(m, t))
    out.append("%%s: fps_reg %%.4f | sort %%.4f + fps_sorted %%.4f identical %%s" %% (name, a["fps_reg"], b["nnp_sort"], b["fps_sorted"], same))
print(" || ".join(out))
''' % ROOT
for tag in (sys.argv[1:] or ["base"]):
    env = dict(os.environ)
    if tag != "base":
        env["RFOPS_LIB"] = os.path.join(ROOT, "rfnet_amd", "variants", f"librfops_{tag}.so")
    o = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print(f"{tag:10s} {o.stdout.strip()} {o.stderr.strip()[-400:] if o.returncode else ''}", flush=True)
