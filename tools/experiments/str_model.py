"""numpy model of an STR (sort-tile-recursive) key that a counting sort can produce (round 3): x slabs from the marginal
256-bin histogram, y strips from per-slab 256-bin histograms, 9 bits of z inside a strip, boustrophedon in y and z so that
a leaf straddling two strips stays compact -- against the product's equalised-Hilbert order and the exact STR of
cull_model_orders.sort_str, counted with cull_model.sim (block scans / box tests / superblock steps per 64-query group)."""
import sys

import numpy as np

import cull_model as cm
from cull_model_orders import sort_str


def sort_str_hist(p, leaf=64, hb=256, zbits=9, snake=True):
    n = len(p)
    s = max(1, round((n / leaf) ** (1 / 3)))
    lo, hi = p.min(0), p.max(0)
    f = np.minimum(((p - lo) / (hi - lo + 1e-30) * hb).astype(np.int64), hb - 1)
    # slabs: equal mass from the x histogram
    hx = np.bincount(f[:, 0], minlength=hb)
    cx = np.cumsum(hx) - hx
    slab_of_bin = np.minimum(cx * s // n, s - 1)
    slab = slab_of_bin[f[:, 0]]
    strip = np.zeros(n, np.int64)
    for a in range(s):
        m = slab == a
        hy = np.bincount(f[m, 1], minlength=hb)
        cy = np.cumsum(hy) - hy
        strip_of_bin = np.minimum(cy * s // max(m.sum(), 1), s - 1)
        st = strip_of_bin[f[m, 1]]
        if snake and (a & 1):
            st = s - 1 - st
        strip[m] = st
    z = np.minimum(((p[:, 2] - lo[2]) / (hi[2] - lo[2] + 1e-30) * (1 << zbits)).astype(np.int64), (1 << zbits) - 1)
    col = slab * s + strip
    if snake:
        z = np.where(col & 1, (1 << zbits) - 1 - z, z)
    key = col * (1 << zbits) + z
    return p[np.argsort(key, kind="stable")]


if __name__ == "__main__":
    rng = np.random.RandomState(100)
    for kind in ("randn", "uniform", "sphere"):
        def gen(n):
            if kind == "randn":
                return rng.randn(n, 3).astype(np.float32)
            if kind == "uniform":
                return rng.rand(n, 3).astype(np.float32)
            x = rng.randn(n, 3)
            return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
        A, B = gen(2048), gen(16384)
        for name, f in (("hilbert5eq (product)", lambda p: cm.sort_hilbert_eq(p, 5)), ("STR-64 exact", lambda p: sort_str(p, 64)),
                        ("STR-64 hist, snake", lambda p: sort_str_hist(p, 64)), ("STR-64 hist, raster", lambda p: sort_str_hist(p, 64, snake=False)),
                        ("STR-128 hist, snake", lambda p: sort_str_hist(p, 128)), ("STR-32 hist, snake", lambda p: sort_str_hist(p, 32))):
            As, Bs = f(A), f(B)
            r = np.random.RandomState(1)
            s1 = cm.sim(As, Bs, 64, 16, 4, 16, r)
            s2 = cm.sim(Bs, As, 64, 16, 4, 32, r)
            c = (cm.cost(s1, 16, 28) * 32 * 32 + cm.cost(s2, 16, 22) * 256 * 32) / 9.3e11 * 1e6
            print(f"{kind:8s} {name:22s} A>B scans {s1[0]:5.0f} tests {s1[1]:5.0f} steps {s1[2]:4.0f} | B>A scans {s2[0]:5.1f} tests {s2[1]:4.0f} "
                  f"steps {s2[2]:4.0f} | modelled C2 sweep {c:5.1f} us")
