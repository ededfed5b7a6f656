#!/usr/bin/env python3
"""A/B of the Chamfer forward under an environment knob (default RF_NN_WAVES, a knob of the DENSE
sweep: the tool pins RF_NN_MODE=dense unless AB_MODE says otherwise; AB_ENV=RF_NNP_SPLIT AB_MODE=culled
for the culled sweep's), each variant in its own subprocess but interleaved on the SAME device
(rule: never compare across devices/boxes)."""
import os
import subprocess
import sys

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
x1 = torch.from_numpy(rng.randn(32, 2048, 3).astype(np.float32)).cuda()
x2 = torch.from_numpy(rng.randn(32, 16384, 3).astype(np.float32)).cuda()
y1 = torch.from_numpy(rng.randn(32, 16384, 3).astype(np.float32)).cuda()
print("C2 %%.4f ms   NS %%.4f ms" %% (t(lambda: R.nn_distance(x1, x2), 50), t(lambda: R.nn_distance(y1, x2), 15)))
''' % ROOT
ENVNAME = os.environ.get("AB_ENV", "RF_NN_WAVES")
variants = sys.argv[1:] or ["2048", "4096", "8192", "16384"]
for rnd in range(2):
    for v in variants:
        env = dict(os.environ, **{ENVNAME: v, "RF_NN_MODE": os.environ.get("AB_MODE", "dense")})
        out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
        print(f"round {rnd} {ENVNAME}={v:>6s}: {out.stdout.strip()} {out.stderr.strip()[-200:] if out.returncode else ''}")
