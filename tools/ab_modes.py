"""auto vs dense vs culled Chamfer forward over a spread of shapes and point distributions, both
directions and one direction only (is the size rule of `culled_pays` ever far off, and on which
data?).  python tools/ab_modes.py [--quick]"""
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from rfnet_amd import _lib  # noqa: E402
from rfnet_amd import _raw  # noqa: E402


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    _lib.profile_collect()
    _lib.profile_enable(True)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    return sum(v[0] for v in _lib.profile_collect().values()) / iters  # ms of kernels per call


def clouds(kind, rng, b, n, m):
    if kind == "randn":
        return rng.randn(b, n, 3), rng.randn(b, m, 3)
    if kind == "uniform":
        return rng.rand(b, n, 3) - 0.5, rng.rand(b, m, 3) - 0.5
    # "snapped": what the untrained RFNet emits -- the larger cloud collapsed onto ~3000 spots of the
    # smaller one's parent cloud plus a small move (merge_layer with decfactor ~ 1, vv_recon.py:132-139)
    base = rng.rand(b, 3000, 3) - 0.5
    lo, hi = (n, m) if n <= m else (m, n)
    small = base[:, :lo] if lo <= 3000 else rng.rand(b, lo, 3) - 0.5
    big = np.take_along_axis(base, rng.randint(0, 3000, (b, hi))[..., None], 1) + 0.01 * np.tanh(rng.randn(b, hi, 3))
    return (small, big) if n <= m else (big, small)


SHAPES = [(1, 65536, 65536), (1, 16384, 16384), (2, 65536, 4096), (128, 1024, 1024), (4, 3000, 16384),
          (32, 512, 16384), (32, 3000, 1024), (256, 2048, 2048), (8, 8192, 8192), (64, 2048, 16384),
          (1, 4096, 4096), (16, 700, 20000), (32, 2048, 16384), (32, 3000, 16384), (32, 16384, 16384),
          (32, 1024, 16384), (8, 2048, 2048), (32, 4096, 4096), (2, 16384, 16384), (4, 2048, 16384)]
if "--quick" in sys.argv:
    SHAPES = SHAPES[12:16]
rng = np.random.RandomState(3)
worst = {}
for kind in ("randn", "uniform", "snapped"):
    for (b, n, m) in SHAPES:
        x, y = clouds(kind, rng, b, n, m)
        a = torch.from_numpy(x.astype(np.float32)).cuda()
        c = torch.from_numpy(y.astype(np.float32)).cuda()
        t = {mode: timeit(lambda: _raw.nn_distance(a, c, mode=mode)) for mode in ("auto", "dense", "culled")}
        pick = "culled" if abs(t["auto"] - t["culled"]) < abs(t["auto"] - t["dense"]) else "dense"
        best = min(t["dense"], t["culled"])
        # one direction (direction 2: the m-side queries, as merge_layer / zero_groupnear use it)
        h1, h2 = _raw.nn_sort(a), _raw.nn_sort(c)
        t1 = {"auto": timeit(lambda: _raw.nn_distance_dir(a, c, False, True)),
              "sorted": timeit(lambda: _raw.nn_distance_sorted(h1, h2, False, True)),
              "sort": timeit(lambda: (_raw.nn_sort(a), _raw.nn_sort(c)))}
        flag = "ok" if t["auto"] <= 1.15 * best else "WRONG PICK"
        worst[(b, n, m)] = max(worst.get((b, n, m), 0.0), t["culled"] / t["dense"])
        print(f"{kind:8s} b={b:4d} n={n:6d} m={m:6d}: dense {t['dense']:.3f} culled {t['culled']:.3f} auto {t['auto']:.3f} ms "
              f"(takes {pick}; {flag}) | dir2 only: auto {t1['auto']:.3f}, sweep on handles {t1['sorted']:.3f} (+ sorts {t1['sort']:.3f})",
              flush=True)
print("worst culled/dense ratio per shape over the distributions:")
for k, v in worst.items():
    print("  ", k, f"{v:.2f}")
