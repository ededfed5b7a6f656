#!/usr/bin/env python3
"""Random-shape soak of the boxed three_nn (rf_threenn_boxes) against the scan kernel (rf_threenn), which tests/ pin to the
oracle: both must agree bit for bit -- distances AND indices, ties included -- on every shape and cloud kind until the time
runs out.  Every 20th case also runs three_interpolate / its gradient on the result, row / tile forms against the
element-per-thread kernels' arithmetic restated in torch (out bit-exact; grad within fp32 summation noise).
usage: python tools/soak_three_nn.py [seconds=120] [seed=1]"""
import os, sys, time
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rfnet_amd import _raw as R  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda:0")


def cloud(b, n, kind):
    if kind == 0:
        a = rng.randn(b, n, 3)
    elif kind == 1:
        a = rng.rand(b, n, 3) - 0.5
    elif kind == 2:  # clustered
        k = max(1, n // 200)
        c = rng.randn(b, k, 3)
        a = c[:, rng.randint(0, k, n)] + 0.02 * rng.randn(b, n, 3)
    elif kind == 3:  # duplicates (resample_pcd style)
        base = rng.randn(b, max(1, n // 3), 3)
        a = base[:, rng.randint(0, base.shape[1], n)]
    elif kind == 4:  # lattice: exact ties in distance
        a = rng.randint(-4, 5, size=(b, n, 3)).astype(np.float64) * 0.25
    elif kind == 5:  # sphere surface
        a = rng.randn(b, n, 3)
        a /= np.linalg.norm(a, axis=2, keepdims=True) + 1e-12
    else:  # flat: no extent on one axis
        a = rng.rand(b, n, 3)
        a[..., rng.randint(0, 3)] = 0.25
    return a.astype(np.float32)


def logint(lo, hi):
    return int(round(np.exp(rng.uniform(np.log(lo), np.log(hi)))))


t_end = time.time() + secs
cases = mism = interp = 0
while time.time() < t_end:
    b = int(rng.randint(1, 5))
    n, m = logint(1, 20000), logint(1, 20000)
    ku, kk = int(rng.randint(0, 7)), int(rng.randint(0, 7))
    u, k = cloud(b, n, ku), cloud(b, m, kk)
    if rng.rand() < 0.3 and n > 8 and m > 8:
        u[:, : min(n, m) // 2] = k[:, : min(n, m) // 2]  # distance 0
    if rng.rand() < 0.1:
        u[0, rng.randint(0, n), rng.randint(0, 3)] = [np.nan, np.inf, -np.inf][rng.randint(0, 3)]
    if rng.rand() < 0.1:
        k[0, rng.randint(0, m), rng.randint(0, 3)] = [np.nan, np.inf, -np.inf][rng.randint(0, 3)]
    tu, tk = torch.from_numpy(u).to(dev), torch.from_numpy(k).to(dev)
    sd, si = R.three_nn(tu, tk, form="scan")
    bd, bi = R.three_nn(tu, tk, form="boxes")
    cases += 1
    if not (torch.equal(si, bi) and torch.equal(sd.view(torch.int32), bd.view(torch.int32))):
        mism += 1
        bad = ((si != bi) | (sd.view(torch.int32) != bd.view(torch.int32))).any(dim=2).nonzero()
        w = bad[0].tolist()
        print(f"MISMATCH b={b} n={n} m={m} kinds={ku},{kk} rows={len(bad)} first={w} scan={sd[w[0], w[1]].tolist()} {si[w[0], w[1]].tolist()} "
              f"boxes={bd[w[0], w[1]].tolist()} {bi[w[0], w[1]].tolist()}", flush=True)
    if cases % 20 == 0 and m >= 3:
        c = int(rng.choice([1, 3, 8, 16, 24, 64, 128]))
        pts = torch.randn(b, m, c, device=dev)
        w3 = torch.rand(b, n, 3, device=dev)
        go = torch.randn(b, n, c, device=dev)
        out = R.three_interpolate(pts, si, w3)
        li = si.long()
        g = [torch.gather(pts, 1, li[:, :, t:t + 1].expand(b, n, c)) * w3[:, :, t:t + 1] for t in range(3)]
        want = (g[0] + g[1]) + g[2]
        gp = R.three_interpolate_grad(pts, si, w3, go)
        wantg = torch.zeros(b, m, c, device=dev, dtype=torch.float64)
        for t in range(3):
            wantg.scatter_add_(1, li[:, :, t:t + 1].expand(b, n, c), (go * w3[:, :, t:t + 1]).double())
        interp += 1
        tol = 1e-5 * max(1.0, float(wantg.abs().max()))
        if not torch.equal(out, want) or float((gp.double() - wantg).abs().max()) > tol + 1e-5 * float(wantg.abs().max()):
            mism += 1
            print(f"INTERPOLATE MISMATCH b={b} n={n} m={m} c={c} out_equal={bool(torch.equal(out, want))} "
                  f"grad_err={float((gp.double() - wantg).abs().max()):.3e}", flush=True)
print(f"{cases} cases ({interp} with three_interpolate + grad), {mism} mismatches, {secs:.0f} s")
sys.exit(1 if mism else 0)
