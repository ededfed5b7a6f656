#!/usr/bin/env python3
"""Same-device A/B of library variants (tools/build_variant.py) on rf_chamfer_step at C2: per-kernel hipEvent
averages and wall time per step, every variant in its own process (RFOPS_LIB), interleaved `rounds` times; the
forward outputs are checked against the dense sweep and the gradients against the original-order backward.
usage: python tools/ab_step.py [--shape B,N,M] TAG [TAG ...]   ('base' = the product)"""
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

CHILD = r'''
import json, sys, time, numpy as np, torch
sys.path.insert(0, %r)
from rfnet_amd import _lib, _raw as R
B, N, M = json.loads(sys.argv[1])
rng = np.random.RandomState(100)
a = torch.from_numpy(rng.randn(B, N, 3).astype(np.float32)).cuda()
c = torch.from_numpy(rng.randn(B, M, 3).astype(np.float32)).cuda()
g1 = torch.ones(B, N, device="cuda"); g2 = torch.ones(B, M, device="cuda")
plan = R.ChamferStep(B, N, M, "cuda")
out = plan(a, c, g1, g2)
ref = R.nn_distance(a, c, mode="dense")
ok = all(torch.equal(x, y) for x, y in zip(ref, out[:4]))
r1, r2 = R.nn_distance_grad(a, c, g1, out[1], g2, out[3])
okg = bool(torch.allclose(out[4], r1, rtol=1e-5, atol=1e-5 * float(r1.abs().max()))) and \
      bool(torch.allclose(out[5], r2, rtol=1e-5, atol=1e-5 * float(r2.abs().max())))
for _ in range(10): plan(a, c, g1, g2)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): plan(a, c, g1, g2)
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 100 * 1e6
_lib.profile_collect(); _lib.profile_enable(True)
for _ in range(50): plan(a, c, g1, g2)
torch.cuda.synchronize(); _lib.profile_enable(False)
pr = _lib.profile_collect()
res = {"ok": ok, "okg": okg, "wall_us": round(wall, 1), **{k: round(v[0] / v[1] * 1e3, 1) for k, v in pr.items()}}
# the two-op path in the same process (culled sweep without the emit + the original-order backward)
def two():
    o = R.nn_distance(a, c)
    return R.nn_distance_grad(a, c, g1, o[1], g2, o[3])
for _ in range(5): two()
torch.cuda.synchronize(); _lib.profile_collect(); _lib.profile_enable(True)
for _ in range(50): two()
torch.cuda.synchronize(); _lib.profile_enable(False)
pr = _lib.profile_collect()
res.update({"twoop_" + k: round(v[0] / v[1] * 1e3, 1) for k, v in pr.items()})
print(json.dumps(res))
''' % ROOT


def main():
    args = sys.argv[1:]
    shape = [32, 2048, 16384]
    if args and args[0] == "--shape":
        shape = [int(x) for x in args[1].split(",")]
        args = args[2:]
    tags = args or ["base"]
    acc = {}
    for r in range(3):
        for tag in tags:
            env = dict(os.environ)
            if tag != "base":
                env["RFOPS_LIB"] = os.path.join(ROOT, "rfnet_amd", "variants", f"librfops_{tag}.so")
            out = subprocess.run([sys.executable, "-c", CHILD, json.dumps(shape)], capture_output=True, text=True, env=env)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(tag, "FAILED", out.stderr[-500:])
                continue
            acc.setdefault(tag, []).append(json.loads(line[-1]))
    for tag, vs in acc.items():
        keys = [k for k in vs[0] if k not in ("ok", "okg")]
        print(f"{tag:14s} fwd_ok={all(v['ok'] for v in vs)} grad_ok={all(v['okg'] for v in vs)}  " +
              "  ".join(f"{k} {sorted(v[k] for v in vs)}" for k in keys))


if __name__ == "__main__":
    main()
