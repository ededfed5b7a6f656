"""auto vs dense vs culled Chamfer forward over a spread of shapes (is the size rule of
`culled_pays` ever far off?).  python tools/ab_modes.py"""
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from rfnet_amd import _raw  # noqa: E402
from ab_culled import timeit  # noqa: E402

rng = np.random.RandomState(3)
for (b, n, m) in [(1, 65536, 65536), (1, 16384, 16384), (2, 65536, 4096), (128, 1024, 1024), (4, 3000, 16384),
                  (32, 512, 16384), (32, 3000, 1024), (256, 2048, 2048), (8, 8192, 8192), (64, 2048, 16384),
                  (1, 4096, 4096), (16, 700, 20000)]:
    a = torch.from_numpy(rng.randn(b, n, 3).astype(np.float32)).cuda()
    c = torch.from_numpy(rng.randn(b, m, 3).astype(np.float32)).cuda()
    t = {}
    for mode in ("auto", "dense", "culled"):
        t[mode] = timeit(lambda: _raw.nn_distance(a, c, mode=mode), 10)
    pick = "culled" if abs(t["auto"] - t["culled"]) < abs(t["auto"] - t["dense"]) else "dense"
    best = min(t["dense"], t["culled"])
    print(f"b={b:4d} n={n:6d} m={m:6d}: dense {t['dense']:.3f}  culled {t['culled']:.3f}  auto {t['auto']:.3f} ms "
          f"(takes {pick}; {'ok' if t['auto'] <= 1.15 * best else 'WRONG PICK'})")
