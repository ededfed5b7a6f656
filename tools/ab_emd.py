#!/usr/bin/env python3
"""Same-device A/B of approx_match under an env knob (default RF_AM_WAVES; AB_ENV=RFOPS_LIB compares two builds of the library)."""
import os, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CODE = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from rfnet_amd import _raw as R
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
u1, v1 = u[:, :1024].contiguous(), v[:, :1024].contiguous()
u0, v0 = u[:, :64].contiguous(), v[:, :64].contiguous()
print("C4 2048^2 %%.4f ms   1024^2 %%.4f ms   64^2 %%.4f ms" %% (t(lambda: R.approx_match(u, v), 10), t(lambda: R.approx_match(u1, v1), 20), t(lambda: R.approx_match(u0, v0), 20)))
''' % ROOT
ENVNAME = os.environ.get("AB_ENV", "RF_AM_WAVES")
for rnd in range(2):
    for v in (sys.argv[1:] or ["1", "2"]):
        out = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, **{ENVNAME: v}), capture_output=True, text=True)
        print(f"round {rnd} {ENVNAME}={v}: {out.stdout.strip()} {out.stderr.strip()[-300:] if out.returncode else ''}")
