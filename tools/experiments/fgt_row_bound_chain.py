"""The per-row certificate of emd_fgt.hip on the weights the schedule REALLY meets at its broad levels (fgt_row_bound.py uses flat weights):
the chain in float64 numpy for one sample of C4 (the reference schedule, tf_approxmatch.cu:21-177), and at every phase of levels
-1, -0.25: the true truncation error of the degree-P series per row, the bound E(x)/S(x), and the share of rows the bound fails.
usage: python tools/experiments/fgt_row_bound_chain.py [kind]   (filled | corners | partial)"""
import math
import sys

import numpy as np

kind = sys.argv[1] if len(sys.argv) > 1 else "filled"
rng = np.random.RandomState(100)
n = 2048
if kind == "filled":
    A = rng.random_sample((n, 3)) - 0.5
    B = rng.random_sample((n, 3)) - 0.5
elif kind == "corners":
    A = np.clip(0.45 + 0.03 * rng.randn(n, 3), -.5, .5)
    B = np.clip(-0.45 + 0.03 * rng.randn(n, 3), -.5, .5)
else:
    A = np.clip(0.3 + 0.1 * rng.randn(n, 3), -.5, .5)
    B = rng.random_sample((n, 3)) - 0.5
D2 = ((A[:, None, :] - B[None, :, :]) ** 2).sum(-1)  # [k][l]
levels = [-4.0 ** j for j in range(7, -2, -1)] + [0.0]
lo = np.minimum(A.min(0), B.min(0)); hi = np.maximum(A.max(0), B.max(0)); O = 0.5 * (lo + hi)
xa, xb = A - O, B - O
ra, rb = np.linalg.norm(xa, axis=1), np.linalg.norm(xb, axis=1)


def check(rows, cols, rrow, rcol, w, a, P, what):
    """rows/cols relative to O; S[row] = sum_col w exp(-a |row-col|^2)"""
    if a == 0:
        return
    g = 2 * a
    W = w * np.exp(-a * rcol ** 2)
    t = g * rows @ cols.T
    ser = np.zeros_like(t); term = np.ones_like(t)
    for k in range(P + 1):
        ser += term; term = term * t / (k + 1)
    St = np.exp(-a * rrow ** 2) * (np.exp(t) @ W)
    Ss = np.exp(-a * rrow ** 2) * (ser @ W)
    rel = np.abs(Ss - St) / np.maximum(St, 1e-300)
    Q = (W * rcol ** (P + 1)).sum()
    bound = np.exp(-a * rrow ** 2) * (g * rrow) ** (P + 1) / math.factorial(P + 1) * np.exp(g * rrow * rcol.max()) * Q
    rb_ = bound / np.maximum(Ss, 1e-300)
    print(f"  {what:34s} a={a:5.2f} P={P:2d}: true rel err max {rel.max():.2e} | bound max {rb_.max():.2e} median {np.median(rb_):.2e} | "
          f"rows failing 1e-7: {(rb_ > 1e-7).mean():.4f}  3e-7: {(rb_ > 3e-7).mean():.4f}  1e-6: {(rb_ > 1e-6).mean():.4f}")


remL = np.ones(n); remR = np.ones(n)
ratioR_prev = None; a_prev = None
for lv, level in enumerate(levels):
    a = -level
    E = np.exp(level * D2)
    if lv >= 7:
        print(f"level {level}:")
        P = 10 if a > 0.3 else 8
        check(xa, xb, ra, rb, remR, a, P, "P1: rows xyz1, w = remainR")
    ratioL = remL / (1e-9 + E @ remR)
    if lv >= 7:
        check(xb, xa, rb, ra, ratioL, a, P, "P2: rows xyz2, w = ratioL")
    sumr = E.T @ ratioL
    s = sumr * remR
    cons = np.minimum(remR / (s + 1e-9), 1.0)
    ratioR = remR * cons
    remR = np.maximum(0.0, remR - s)
    if lv >= 7 and lv + 1 < len(levels):
        check(xa, xb, ra, rb, ratioR, a, P, "P3: rows xyz1, w = ratioR")
    remL = np.maximum(0.0, remL - ratioL * (E @ ratioR))
    print(f"   after level {level}: remainL sum {remL.sum():.4f}  remainR sum {remR.sum():.4f}") if lv >= 5 else None
