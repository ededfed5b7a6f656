"""Dense vs culled Chamfer sweep on one GPU: time per call (hipEvents via torch) and the culled
sweep's counters.  python tools/ab_culled.py [kind]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from rfnet_amd import _raw  # noqa: E402
from rfnet_amd._lib import profile_collect, profile_enable  # noqa: E402


def cloud(rng, kind, b, k):
    if kind == "uniform":
        return rng.random_sample((b, k, 3)).astype(np.float32)
    if kind == "sphere":
        x = rng.randn(b, k, 3)
        return (x / np.linalg.norm(x, axis=-1, keepdims=True)).astype(np.float32)
    if kind == "dup":  # every point repeated ~5 times (resample_pcd-style duplicates)
        base = rng.randn(b, max(k // 5, 1), 3).astype(np.float32)
        return np.take_along_axis(base, rng.randint(0, base.shape[1], size=(b, k, 1)), 1)
    if kind == "same":  # the worst case: all points identical (nothing can be culled, every lane ties)
        return np.ones((b, k, 3), np.float32)
    return rng.randn(b, k, 3).astype(np.float32)


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "randn"
    rng = np.random.RandomState(100)
    for (b, n, m) in [(32, 2048, 16384), (32, 16384, 16384), (32, 3000, 16384), (32, 1024, 1024), (32, 4096, 4096),
                      (8, 2048, 16384)]:
        a = torch.from_numpy(cloud(rng, kind, b, n)).cuda()
        c = torch.from_numpy(cloud(rng, kind, b, m)).cuda()
        td = timeit(lambda: _raw.nn_distance(a, c, mode="dense"), 20)
        tc = timeit(lambda: _raw.nn_distance(a, c, mode="culled"), 20)
        st = []
        _raw.nn_distance(a, c, mode="culled", stats=st)
        profile_enable(True)
        for _ in range(10):
            _raw.nn_distance(a, c, mode="culled")
        torch.cuda.synchronize()
        prof = {k: round(v[0] / v[1], 4) for k, v in profile_collect().items()}
        profile_enable(False)
        if any(st[16:26]):
            ts = [v for v in st[16:32] if v]
            print("    sort phases (s_memtime ticks, 100 MHz):", [ts[i + 1] - ts[i] for i in range(len(ts) - 1)])
        pairs = b * n * m
        print(f"{kind} b={b} n={n} m={m}: dense {td:.3f} ms  culled {tc:.3f} ms  ({td / tc:.2f}x)  kernels {prof}")
        for d, (q, k) in enumerate(((n, m), (m, n))):
            w, steps, tests, scans = st[4 * d:4 * d + 4]
            print(f"    dir{d}: waves {w} steps/wave {steps / w:.1f} (max {tests}) scans/wave {scans / w:.1f} (max {st[8 + d]})"
                  f"  pairs evaluated {scans * 16 * 64 / pairs * 100:.2f}% of b*n*m")


if __name__ == "__main__":
    main()
