#!/usr/bin/env python3
"""Same-device A/B of library variants (tools/build_variant.py) on the Chamfer forward: every variant
runs in its own process (RFOPS_LIB), interleaved `rounds` times; outputs are checked against the
dense sweep.  usage: python tools/ab_variants.py [--shapes C2,NS,...] TAG [TAG ...]   ('base' = the product)"""
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
SHAPES = {"C2": (32, 2048, 16384), "NS": (32, 16384, 16384), "M3": (32, 3000, 16384), "S4": (8, 4096, 4096)}

CHILD = r'''
import json, sys, numpy as np, torch
sys.path.insert(0, %r)
from rfnet_amd import _lib, _raw as R
res = {}
for name, (B, N, M), dist in json.loads(sys.argv[1]):
    rng = np.random.RandomState(100)
    if dist == "randn":
        a, c = rng.randn(B, N, 3), rng.randn(B, M, 3)
    else:
        base = rng.rand(B, 3000, 3) - 0.5
        a = base[:, :N] if N <= 3000 else rng.rand(B, N, 3) - 0.5
        c = np.take_along_axis(base, rng.randint(0, 3000, (B, M))[..., None], 1) + 0.002 * rng.randn(B, M, 3)
    a = torch.from_numpy(a.astype(np.float32)).cuda(); c = torch.from_numpy(c.astype(np.float32)).cuda()
    ref = R.nn_distance(a, c, mode="dense")
    got = R.nn_distance(a, c, mode="culled")
    ok = all(torch.equal(x, y) for x, y in zip(ref, got))
    for _ in range(5): R.nn_distance(a, c, mode="culled")
    torch.cuda.synchronize(); _lib.profile_collect(); _lib.profile_enable(True)
    for _ in range(30): R.nn_distance(a, c, mode="culled")
    torch.cuda.synchronize(); _lib.profile_enable(False)
    pr = _lib.profile_collect()
    res[name + ":" + dist] = {"ok": ok, **{k: round(v[0] / v[1] * 1e3, 1) for k, v in pr.items()}}
print(json.dumps(res))
''' % ROOT


def main():
    args = sys.argv[1:]
    shapes = ["C2", "NS"]
    if args and args[0] == "--shapes":
        shapes = args[1].split(",")
        args = args[2:]
    work = [(s, SHAPES[s], d) for s in shapes for d in ("randn", "clustered")]
    tags = args or ["base"]
    rounds = 3
    acc = {}
    for r in range(rounds):
        for tag in tags:
            env = dict(os.environ)
            if tag != "base":
                env["RFOPS_LIB"] = os.path.join(ROOT, "rfnet_amd", "variants", f"librfops_{tag}.so")
            out = subprocess.run([sys.executable, "-c", CHILD, json.dumps(work)], capture_output=True, text=True, env=env)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(tag, "FAILED", out.stderr[-500:])
                continue
            for k, v in json.loads(line[-1]).items():
                acc.setdefault((tag, k), []).append(v)
    for (tag, k), vs in sorted(acc.items(), key=lambda kv: (kv[0][1], kv[0][0])):
        sw = sorted(v.get("nnp_sweep", 0) for v in vs)
        so = sorted(v.get("nnp_sort", 0) for v in vs)
        print(f"{k:16s} {tag:14s} ok={all(v['ok'] for v in vs)} sweep_us {sw}  sort_us {so}")


if __name__ == "__main__":
    main()
