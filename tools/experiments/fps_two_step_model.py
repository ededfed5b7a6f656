"""Could fps_sorted_kernel take SEVERAL samples out of one round of its serial chain (scan -> wave reduction -> barrier -> 16-slot
reduction -> winner's coordinates)?  numpy model on C3's cloud (16384 -> 1024), lanes = 16 consecutive sorted points as in the kernel.
Exact rule: with the lanes' maxima sorted (t1 >= t2 >= ...: candidates c1, c2, ...), c_k is the k-th next sample for certain when
  (i)  d2(c_i, c_k) >= t_k for every accepted i < k   (c_k's running minimum does not change),
  (ii) bound_i < t_k for every accepted i < k          (what is left in the lane c_i came from cannot reach t_k), and t_k > 0
-- every other point only falls.  bound_i is an upper bound on the lane's other 15 points once c_i is a sample:
  mode D       min(t_i, squared diameter of the lane's box)   -- free (a constant per lane)
  mode second  the lane's SECOND largest running minimum       -- exact; costs a top-2 per lane and per scan
Result (this file's output): with D the second candidate is accepted in 4 % of the rounds (the lanes' boxes are not small against
the running minima: 986 rounds for 1024 samples); with the exact second 91 % (535 rounds at K = 2, 303 at K = 4, 204 at K = 8).
NOT BUILT.  What a round of K samples saves is (K - 1) / K of the 0.29 us fixed chain per sample and the idle SIMDs of a round
that scans 2.9 of 16 waves; what it adds, per touched wave and round, is a top-2 per lane (+~35 VALU per scan), one more wave
reduction PER CANDIDATE the wave offers, each depending on the one before (+~0.1 us each: DPP chain, ballot, indexed register
reads of rank and bound), and behind the barrier a K-fold selection over 16 x K' entries (+~0.07 us per candidate).  By the
busiest-SIMD model of fps_region_model.py: K = 2 0.66-0.75 us per sample, K = 8 with two candidates per wave ~0.70, against 0.767
today -- inside the model's error, for a rewrite of the kernel's bookkeeping; the bookkeeping, not the scan, is what a round costs.
usage: python tools/experiments/fps_two_step_model.py"""
import numpy as np

rng = np.random.RandomState(100)
n, m = 16384, 1024
P = rng.random_sample((32, n, 3)).astype(np.float32)[0]


def str_order(P, SS=6):
    order = []
    xs = np.argsort(P[:, 0], kind="stable")
    for si, slab in enumerate(np.array_split(xs, SS)):
        ys = slab[np.argsort(P[slab, 1], kind="stable")]
        strips = np.array_split(ys, SS)
        if si & 1:
            strips = strips[::-1]
        for ti, strip in enumerate(strips):
            zs = strip[np.argsort(P[strip, 2], kind="stable")]
            col = si * SS + (ti if not (si & 1) else SS - 1 - ti)
            order.append(zs[::-1] if col & 1 else zs)
    return np.concatenate(order)


lanes = str_order(P).reshape(1024, 16)
X = P[lanes]
D = ((X.max(1) - X.min(1)) ** 2).sum(1)
tdp = np.full(n, 1e38, np.float32); cur = 0; plain = [0]
for j in range(1, m):
    tdp = np.minimum(tdp, ((P - P[cur]) ** 2).sum(1)); cur = int(tdp.argmax()); plain.append(cur)
for K in (2, 4, 8):
    for mode in ("D", "second"):
        td = np.full(lanes.shape, 1e38, np.float32); samples = [0]; last = [0]; rounds = 0; got = np.zeros(K + 1, int)
        while len(samples) < m:
            for s in last:
                td = np.minimum(td, ((X - P[s]) ** 2).sum(-1))
            rounds += 1
            lmx = td.max(1); sec = np.sort(td, 1)[:, -2]
            c, bounds = [], []
            for q, L in enumerate(np.argsort(-lmx, kind="stable")[:K]):
                t = lmx[L]; cq = int(lanes[L][td[L].argmax()])
                if q > 0 and not (t > 0 and all(b < t for b in bounds) and all(((P[ci] - P[cq]) ** 2).sum() >= t for ci in c)
                                  and len(samples) + len(c) < m):
                    break
                c.append(cq); bounds.append(min(t, D[L]) if mode == "D" else sec[L])
            got[len(c)] += 1; samples += c; last = c
        print(f"K = {K}  bound = {mode:6s}  identical to plain FPS: {plain == samples[:m]}   rounds {rounds:4d}   rounds by samples taken {got[1:].tolist()}")
