#!/usr/bin/env python3
"""Soak of approx_match's live-column route (am_compact_kernel / am_rowk CMP / am_p2_live_kernel: from the third level on the
sweeps run over the columns and rows of set 2 that are not exactly dead) against the pinned swept route of the same library
(RF_EMD_SWEPT: every level a dense sweep), on random shapes from the route's lower edge (512 points) up, random batch sizes and
cloud kinds (box-filling, clusters in opposite corners, a partial shape against a complete one, scaled, duplicated points).
Per case: the two routes' costs within rel 1e-5 (the op's tolerance), every match entry within 2e-3 of a unit mass, the entries
outside abs 1e-6 + rel 1e-4 counted (clamp flips: tests/test_oracle_golden.py::test_match_bar_is_ill_conditioned); the fused
earth_mover cost and its gradients on both routes.  `large`: clouds of 4096 .. 9000 points (where rounds 3-5's cost-only earth_mover ran
its sharp levels culled: this soak is what found that route 1e-5 .. 3e-5 off).
`batch`: batches of 8 .. 24 clouds of 1500 .. 2600 points -- past the 6e7 pairs per call from which the default route sorts the rows of
the sharp levels spatially, skips columns and hands level 1 the column lists of level 0.
usage: python tools/soak_emd_live.py [seconds] [seed] [large | batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rfnet_amd import _raw as R

T = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
LARGE = len(sys.argv) > 3 and sys.argv[3] == "large"
BATCH = len(sys.argv) > 3 and sys.argv[3] == "batch"
LO, HI = (4096, 9000) if LARGE else ((1500, 2600) if BATCH else (512, 3000))
t0 = time.time()
cases = bad_cost = bad_entry = 0
strays = entries = 0
worst_cost = worst_entry = worst_grad = 0.0


def cloud(b, n, kind):
    if kind == 0:
        return rng.random_sample((b, n, 3)) - 0.5
    if kind == 1:
        s = 1.0 if rng.rand() < 0.5 else -1.0
        return np.clip(s * 0.45 + 0.03 * rng.randn(b, n, 3), -0.5, 0.5)
    if kind == 2:
        return np.clip(0.3 * rng.randn(b, n, 3) * rng.rand(1, 1, 3), -0.5, 0.5)
    if kind == 3:
        return (rng.random_sample((b, n, 3)) - 0.5) * float(np.exp(rng.uniform(np.log(0.3), np.log(3.0))))
    x = rng.random_sample((b, n, 3)) - 0.5
    x[:, n // 2:] = x[:, : n - n // 2]
    return x


while time.time() - t0 < T:
    b = rng.randint(8, 25) if BATCH else rng.randint(1, 3 if LARGE else 5)
    n = int(round(np.exp(rng.uniform(np.log(LO), np.log(HI)))))
    m = n if rng.rand() < 0.4 else int(round(np.exp(rng.uniform(np.log(LO), np.log(HI)))))
    ka, kc = rng.randint(0, 5), rng.randint(0, 5)
    a = torch.from_numpy(cloud(b, n, ka).astype(np.float32)).cuda()
    c = torch.from_numpy(cloud(b, m, kc).astype(np.float32)).cuda()
    ma, ms = R.approx_match(a, c), R.approx_match(a, c, mode="swept")
    ca, cs = R.match_cost(a, c, ma), R.match_cost(a, c, ms)
    fa, ga1, ga2 = R.earth_mover(a, c, with_grad=True)
    fs, gs1, gs2 = R.earth_mover(a, c, with_grad=True, mode="swept")
    rc = float(((ca - cs).abs() / cs.abs().clamp_min(1e-30)).max())
    rf = float(((fa - fs).abs() / fs.abs().clamp_min(1e-30)).max())
    if LARGE:  # the cost-only form
        fc = R.earth_mover(a, c)
        rf = max(rf, float(((fc - fs).abs() / fs.abs().clamp_min(1e-30)).max()))
    d = (ma - ms).abs()
    me = float(d.max())
    st = int((d > 1e-6 + 1e-4 * ms.abs()).sum())
    gscale = max(1.0, float(gs1.abs().max()), float(gs2.abs().max()))
    ge = max(float((ga1 - gs1).abs().max()), float((ga2 - gs2).abs().max())) / gscale
    cases += 1
    strays += st
    entries += d.numel()
    worst_cost, worst_entry, worst_grad = max(worst_cost, rc, rf), max(worst_entry, me), max(worst_grad, ge)
    if not (rc <= 1e-5 and rf <= 1e-5):
        bad_cost += 1
        print(f"COST b={b} n={n} m={m} kinds={ka},{kc}: rel {rc:.2e} fused {rf:.2e}")
    if not me < 2e-3:
        bad_entry += 1
        print(f"ENTRY b={b} n={n} m={m} kinds={ka},{kc}: max |d match| {me:.2e}, {st} strays")
print(f"{cases} cases, {bad_cost} outside the cost bar (rel 1e-5), {bad_entry} with an entry beyond 2e-3; worst cost rel {worst_cost:.2e}, "
      f"worst |d match| {worst_entry:.2e}, worst gradient difference {worst_grad:.2e} of the scale; {strays} of {entries} entries "
      f"({strays / max(entries, 1):.2e}) outside abs 1e-6 + rel 1e-4 between the two routes; {time.time() - t0:.0f} s")
sys.exit(1 if bad_cost or bad_entry else 0)
