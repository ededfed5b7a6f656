#!/usr/bin/env python3
"""Which columns a wave of the EMD sweeps must keep at the sharp levels (approxmatch.hip, am_rowk/am_rowl SKIP): numpy model.
A column is kept when it lies within the level's cut-off radius (exp2(d2 * c) is exactly +0 beyond it) of at least one of the
wave's rows; rows = 64 x RPT consecutive points of the cloud in sort-tile-recursive order (modelled: x slabs -> y strips -> z),
or in input order.  C4: two uniform clouds of 2048 points in a unit cube.  CPU only."""
import numpy as np

rng = np.random.RandomState(0)
n = 2048
A = rng.rand(n, 3) - 0.5
B = rng.rand(n, 3) - 0.5


def str_order(P, leaf=32):
    s = max(1, int(round((len(P) / leaf) ** (1 / 3))))
    out = []
    for xs in np.array_split(np.argsort(P[:, 0]), s):
        for ys in np.array_split(xs[np.argsort(P[xs, 1])], s):
            out.append(ys[np.argsort(P[ys, 2])])
    return np.concatenate(out)


oa = str_order(A)
log2e = 1.44269502
for rows in (64, 128):
    for j in (7, 6, 5, 4):
        r = np.sqrt(161.0 / (4.0 ** j * log2e))
        keep_s, keep_u = [], []
        for g in range(0, n, rows):
            for order, acc in ((oa[g:g + rows], keep_s), (np.arange(g, g + rows), keep_u)):
                d = np.sqrt(((A[order][:, None, :] - B[None, :, :]) ** 2).sum(-1)).min(0)
                acc.append((d < r).mean())
        print(f"rows per wave {rows:3d}  level -4^{j} (cut-off {r:.3f}): columns kept {np.mean(keep_s):.3f} with sorted rows, "
              f"{np.mean(keep_u):.3f} with rows in input order")
