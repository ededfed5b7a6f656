"""numpy model (round 4): what a per-query INITIAL upper bound ("seed") is worth to the culled sweep.
Today a lane starts with cull = +inf and inherits loose bounds from whatever superblock the group box is closest to;
22 % of the lane x block evaluations of a small-set group are needed by the lane that runs them (r03_str_model.txt).
Variants counted with the product's traversal (64 queries per wave, 16-candidate blocks, best-first superblocks):
  none    no seed (the product)
  oracle  seed = the final minimum (lower bound on what any seed can give)
  sbq     per lane: the candidate superblock nearest to the QUERY (not to the group box), its nearest block, 16 pairs
  sbq4    as sbq but only the 4 records of that block nearest in order to ... (first 4): cheaper gather
usage: python tools/experiments/seed_model.py [randn|uniform|sphere]"""
import sys

import numpy as np

import str_model as sm


def boxes(C, BS=16, SB=4):
    nb = len(C) // BS
    Cb = C[: nb * BS].reshape(nb, BS, 3)
    blo, bhi = Cb.min(1), Cb.max(1)
    nsb = nb // SB
    slo = blo.reshape(nsb, SB, 3).min(1)
    shi = bhi.reshape(nsb, SB, 3).max(1)
    return Cb, blo, bhi, slo, shi


def bound(q, lo, hi):  # q (...,3) vs boxes (k,3) -> (..., k)
    g = np.maximum(0, np.maximum(lo[None] - q[:, None], q[:, None] - hi[None]))
    return (g ** 2).sum(-1)


def seed_sbq(q, Cb, blo, bhi, slo, shi, nrec=16):
    """per query: nearest superblock by box bound, nearest of its 4 blocks, min d2 over its first nrec records"""
    lb = bound(q, slo, shi)  # (Q, nsb)
    s = lb.argmin(1)
    out = np.empty(len(q))
    for i in range(len(q)):
        bl = np.arange(s[i] * 4, s[i] * 4 + 4)
        lbb = bound(q[i : i + 1], blo[bl], bhi[bl])[0]
        b = bl[lbb.argmin()]
        out[i] = ((Cb[b][:nrec] - q[i]) ** 2).sum(1).min()
    return out


def sim(Q, C, seed, QW=64, BS=16, SB=4, nwaves=16, rng=None, rebox=True):
    Cb, blo, bhi, slo, shi = boxes(C, BS, SB)
    nw = len(Q) // QW
    waves = rng.choice(nw, min(nwaves, nw), replace=False)
    tot = dict(scans=0, tests=0, steps=0, lane_scans=0, lane_need=0, seedpairs=0)
    for w in waves:
        q = Q[w * QW : (w + 1) * QW]
        final = ((q[:, None, :] - C[None]) ** 2).sum(2).min(1)
        if seed == "none":
            best = np.full(QW, np.inf)
        elif seed == "oracle":
            best = final.copy()
        elif seed == "sbq":
            best = seed_sbq(q, Cb, blo, bhi, slo, shi, 16)
            tot["seedpairs"] += 16 * QW
        elif seed == "sbq4":
            best = seed_sbq(q, Cb, blo, bhi, slo, shi, 4)
            tot["seedpairs"] += 4 * QW
        act = np.ones(QW, bool)
        qlo, qhi = q.min(0), q.max(0)
        lbs = (np.maximum(0, np.maximum(slo - qhi, qlo - shi)) ** 2).sum(1)
        done = np.zeros(len(slo), bool)
        nact_ref = QW
        while True:
            cand = np.where(~done)[0]
            if len(cand) == 0:
                break
            s = cand[lbs[cand].argmin()]
            bnd = lbs[s]
            act = best >= bnd
            if not act.any():
                break
            if rebox and act.sum() * 2 <= nact_ref:
                nact_ref = act.sum()
                qlo, qhi = q[act].min(0), q[act].max(0)
                lbs = (np.maximum(0, np.maximum(slo - qhi, qlo - shi)) ** 2).sum(1)
                continue
            done[s] = True
            tot["steps"] += 1
            for bidx in range(s * SB, (s + 1) * SB):
                lbq = bound(q, blo[bidx : bidx + 1], bhi[bidx : bidx + 1])[:, 0]
                tot["tests"] += 1
                need = lbq <= best
                if not need.any():
                    continue
                d = ((q[:, None, :] - Cb[bidx][None]) ** 2).sum(2).min(1)
                best = np.minimum(best, d)
                tot["scans"] += 1
                tot["lane_scans"] += QW
                tot["lane_need"] += int(need.sum())
        assert np.allclose(best, final), "culling lost a neighbour"
    k = len(waves)
    return {a: b / k for a, b in tot.items()}


if __name__ == "__main__":
    kind = sys.argv[1] if len(sys.argv) > 1 else "randn"
    rng = np.random.RandomState(100)

    def gen(n):
        if kind == "randn":
            return rng.randn(n, 3).astype(np.float32)
        if kind == "uniform":
            return rng.rand(n, 3).astype(np.float32)
        x = rng.randn(n, 3)
        return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)

    A, B = gen(2048), gen(16384)
    As, Bs = sm.sort_str_hist(A, 64), sm.sort_str_hist(B, 64)
    for seed in ("none", "oracle", "sbq", "sbq4"):
        r = np.random.RandomState(1)
        s1 = sim(As, Bs, seed, nwaves=16, rng=r)
        s2 = sim(Bs, As, seed, nwaves=32, rng=r)
        print(f"{kind:8s} seed {seed:7s} small-set group: scans {s1['scans']:6.1f} tests {s1['tests']:6.1f} steps {s1['steps']:5.1f} "
              f"lane-need {100 * s1['lane_need'] / max(s1['lane_scans'], 1):4.0f} % | large-set group: scans {s2['scans']:5.1f} "
              f"tests {s2['tests']:5.1f} steps {s2['steps']:4.1f} lane-need {100 * s2['lane_need'] / max(s2['lane_scans'], 1):4.0f} %")

    print("query tile sizes (pairs evaluated per query = scans * 16; per 64 queries: scans / tests / steps summed over the tiles)")
    for QW in (64, 32, 16, 8, 4, 1):
        for seed in ("none", "oracle", "sbq"):
            r = np.random.RandomState(1)
            s1 = sim(As, Bs, seed, QW=QW, nwaves=max(16, 256 // QW), rng=r)
            s2 = sim(Bs, As, seed, QW=QW, nwaves=max(32, 256 // QW), rng=r)
            f = 64 // QW
            print(f"{kind:8s} QW {QW:2d} seed {seed:7s} small: pairs/query {s1['scans'] * 16:6.0f} per-64q scans {s1['scans'] * f:6.1f} tests {s1['tests'] * f:6.0f} steps {s1['steps'] * f:5.0f}"
                  f" | large: pairs/query {s2['scans'] * 16:5.0f} per-64q scans {s2['scans'] * f:5.1f} tests {s2['tests'] * f:5.0f} steps {s2['steps'] * f:4.0f}")
