"""Variants of the histogram-STR key (str_model.py) in the numpy culling model: slab count, and the order INSIDE a strip --
z only (product), or z tiles of ~64 points whose points go by (x, y) quadrant of the strip first (blocks = pencils, not plates)."""
import numpy as np

import cull_model as cm


def sort_str2(p, s=None, leaf=64, hb=256, zbits=9, quad=False, sy=None):
    n = len(p)
    s = s or max(1, round((n / leaf) ** (1 / 3)))
    sy = sy or s
    lo, hi = p.min(0), p.max(0)
    f = np.minimum(((p - lo) / (hi - lo + 1e-30) * hb).astype(np.int64), hb - 1)
    hx = np.bincount(f[:, 0], minlength=hb)
    cx = np.cumsum(hx) - hx
    slab = np.minimum(cx * s // n, s - 1)[f[:, 0]]
    strip = np.zeros(n, np.int64)
    ymid = np.zeros(n, bool)
    for a in range(s):
        m = slab == a
        hy = np.bincount(f[m, 1], minlength=hb)
        cy = np.cumsum(hy) - hy
        st2 = np.minimum(cy * (2 * sy) // max(m.sum(), 1), 2 * sy - 1)[f[m, 1]]  # half-strips
        st = st2 // 2
        ymid[m] = (st2 & 1) == 1
        if a & 1:
            st = sy - 1 - st
        strip[m] = st
    # x half inside the slab: from the x histogram at twice the resolution
    xh = (np.minimum(cx * (2 * s) // n, 2 * s - 1)[f[:, 0]] & 1) == 1
    hz = np.bincount(f[:, 2], minlength=hb)
    cz = np.cumsum(hz) - hz
    zq = np.minimum(cz * (1 << zbits) // n, (1 << zbits) - 1)[f[:, 2]]
    col = slab * sy + strip
    zq = np.where(col & 1, (1 << zbits) - 1 - zq, zq)
    if quad:
        ntile = max(1, round(n / (s * sy) / leaf))  # z tiles per strip
        zt = zq * ntile >> zbits
        q = xh.astype(np.int64) * 2 + (xh ^ ymid).astype(np.int64)  # gray order of the quadrants
        key = ((col * ntile + zt) * 4 + q) * (1 << zbits) + zq
    else:
        key = col * (1 << zbits) + zq
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
        cfgs = [("product: s=3/6", dict(), dict()), ("quad s=3/6", dict(quad=True), dict(quad=True)),
                ("s=3/5", dict(), dict(s=5)), ("s=3/7", dict(), dict(s=7)), ("s=4/6", dict(s=4), dict()), ("s=2/6", dict(s=2), dict()),
                ("s=3x4/6x7", dict(s=3, sy=4), dict(s=6, sy=7)), ("quad s=3/5", dict(quad=True), dict(quad=True, s=5))]
        for name, ka, kb in cfgs:
            As, Bs = sort_str2(A, **ka), sort_str2(B, **kb)
            r = np.random.RandomState(1)
            s1 = cm.sim(As, Bs, 64, 16, 4, 16, r)
            s2 = cm.sim(Bs, As, 64, 16, 4, 32, r)
            c = (cm.cost(s1, 16, 28) * 32 * 32 + cm.cost(s2, 16, 22) * 256 * 32) / 9.3e11 * 1e6
            print(f"{kind:8s} {name:18s} A>B scans {s1[0]:5.0f} tests {s1[1]:5.0f} steps {s1[2]:4.0f} | B>A scans {s2[0]:5.1f} tests {s2[1]:4.0f} "
                  f"steps {s2[2]:4.0f} | modelled C2 sweep {c:5.1f} us", flush=True)
