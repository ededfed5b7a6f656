"""Soak: the culled Chamfer sweep against the dense one over many random shapes / cloud kinds (every
output bit for bit).  python tools/soak_culled.py [seconds [smallest cloud]]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from rfnet_amd import _raw  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
t0 = time.time()
case = bad = 0
while time.time() - t0 < budget:
    rng = np.random.RandomState(123456 + case)
    b = int(rng.choice([1, 2, 3, 5, 8, 17, 33, 70]))
    lo = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    n, m = (int(v) for v in np.exp(rng.uniform(np.log(lo), np.log(20000), size=2)))
    if b * (n + m) > 600000:
        b = max(1, 600000 // (n + m))
    kind = case % 7

    def cloud(k):
        if kind == 0:
            return rng.randn(b, k, 3)
        if kind == 1:
            return rng.random_sample((b, k, 3))
        if kind == 2:
            return rng.randint(0, 6, size=(b, k, 3)) * 0.5
        if kind == 3:
            base = rng.randn(b, max(k // 6, 1), 3)
            return np.take_along_axis(base, rng.randint(0, base.shape[1], size=(b, k, 1)), 1)
        if kind == 4:
            x = rng.randn(b, k, 3)
            return x / np.linalg.norm(x, axis=-1, keepdims=True)
        if kind == 5:
            x = rng.randn(b, k, 3)
            x[..., rng.randint(0, 3)] = 0.25
            return x
        c = rng.randn(b, 4, 3) * 30
        return c[np.arange(b)[:, None], rng.randint(0, 4, size=(b, k))] + rng.randn(b, k, 3) * 1e-3

    a = torch.from_numpy(np.ascontiguousarray(cloud(n), np.float32)).cuda()
    c = torch.from_numpy(np.ascontiguousarray(cloud(m), np.float32)).cuda()
    got = _raw.nn_distance(a, c, mode="culled")
    ref = _raw.nn_distance(a, c, mode="dense")
    ok = all(torch.equal(g, r) for g, r in zip(got, ref))
    # round-2 entry points: one direction (whichever sweep the size rule picks) and sorted handles
    d1 = _raw.nn_distance_dir(a, c, True, False)
    d2 = _raw.nn_distance_dir(a, c, False, True)
    ok = ok and torch.equal(d1[0], ref[0]) and torch.equal(d1[1], ref[1]) and torch.equal(d2[2], ref[2]) and torch.equal(d2[3], ref[3])
    if case % 3 == 0:
        h1, h2 = _raw.nn_sort(a), _raw.nn_sort(c)
        hs = _raw.nn_distance_sorted(h1, h2, case % 2 == 0, True)
        ok = ok and torch.equal(hs[2], ref[2]) and torch.equal(hs[3], ref[3]) and (hs[0] is None or torch.equal(hs[0], ref[0]))
    if not ok:
        bad += 1
        print("MISMATCH case", case, "b,n,m,kind", b, n, m, kind, flush=True)
    case += 1
print(f"{case} cases, {bad} mismatches, {time.time() - t0:.0f} s")
