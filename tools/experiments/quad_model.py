"""numpy model (round 4) of a QUAD-PER-QUERY sweep for the sparse-query direction (2048 queries in 16384 candidates at C2):
a wave owns a 16-query tile (one 16-record block of the sorted query set); the 4 lanes of a quad share one query and split
a candidate block's 16 records; every quad keeps its OWN list of needed blocks, so quads scan different blocks at the same
time (per-lane gathers) and a wave runs max-over-quads scan rounds instead of the union-over-queries of the 64-lane form.

  1 keys of all superblocks against the tile box, s0 = nearest
  2 seed: per query the nearest block of s0, 16 pairs -> best_q
  3 U = max best_q; superblocks with key <= U; their blocks with box-to-tile-box bound <= U  (tile-level block list BL)
  4 per query: blocks of BL with bound(q, box) <= best_q  -> the quad's list (with the bound)
  5 drain: pop, skip when bound > best_q (strict: ties are scanned), else scan 16 pairs
counts per tile: |SB list|, |BL|, list length / scans / pops per query (mean and max over the tile's 16 queries)
usage: python tools/experiments/quad_model.py [randn|uniform|sphere]"""
import sys

import numpy as np

import str_model as sm
from seed_model import boxes, bound


def boxbox(alo, ahi, blo, bhi):
    g = np.maximum(0, np.maximum(blo - ahi, alo - bhi))
    return (g ** 2).sum(-1)


def tile_sim(Q, C, QW=16, ntiles=64, rng=None, sort_lists=False, nseed_sb=1):
    Cb, blo, bhi, slo, shi = boxes(C)
    nt = len(Q) // QW
    tiles = rng.choice(nt, min(ntiles, nt), replace=False)
    acc = dict(sbl=[], bl=[], llen=[], llen_max=[], scans=[], scans_max=[], pops_max=[], seed_ratio=[])
    for t in tiles:
        q = Q[t * QW : (t + 1) * QW]
        final = ((q[:, None, :] - C[None]) ** 2).sum(2).min(1)
        tlo, thi = q.min(0), q.max(0)
        kb = boxbox(tlo, thi, slo, shi)
        order = np.argsort(kb, kind="stable")
        best = np.full(QW, np.inf)
        seedblk = np.full((QW, nseed_sb), -1)
        if nseed_sb == 0:  # per query: nearest superblock among those that overlap the tile box (key 0) plus the nearest
            S0 = order[: max(1, int((kb == 0).sum()))]
            acc.setdefault("s0", []).append(len(S0))
            lbs0 = bound(q, slo[S0], shi[S0])  # (QW, |S0|)
            sq = S0[lbs0.argmin(1)]
            seedblk = np.full((QW, 1), -1)
            for i in range(QW):
                lbk = bound(q[i : i + 1], blo[sq[i] * 4 : sq[i] * 4 + 4], bhi[sq[i] * 4 : sq[i] * 4 + 4])[0]
                b = sq[i] * 4 + lbk.argmin()
                seedblk[i, 0] = b
                best[i] = ((Cb[b] - q[i]) ** 2).sum(1).min()
        for r in range(nseed_sb):
            s0 = order[r]
            lbk = bound(q, blo[s0 * 4 : s0 * 4 + 4], bhi[s0 * 4 : s0 * 4 + 4])  # (QW, 4)
            j = lbk.argmin(1)
            for i in range(QW):
                b = s0 * 4 + j[i]
                seedblk[i, r] = b
                best[i] = min(best[i], ((Cb[b] - q[i]) ** 2).sum(1).min())
        acc["seed_ratio"].append(np.sqrt(best / final).mean())
        U = best.max()
        sbl = order[kb[order] <= U]
        blk = (sbl[:, None] * 4 + np.arange(4)[None]).ravel()
        tb = boxbox(tlo, thi, blo[blk], bhi[blk])
        BL = blk[tb <= U]
        lbq = bound(q, blo[BL], bhi[BL])  # (QW, |BL|)
        llen, scans, pops = [], [], []
        for i in range(QW):
            m = lbq[i] <= best[i]
            m &= ~np.isin(BL, seedblk[i])
            ent = BL[m]
            lbs = lbq[i][m]
            if sort_lists:
                o = np.argsort(lbs, kind="stable")
                ent, lbs = ent[o], lbs[o]
            ns = 0
            for b, lb in zip(ent, lbs):
                if lb > best[i]:
                    continue
                best[i] = min(best[i], ((Cb[b] - q[i]) ** 2).sum(1).min())
                ns += 1
            llen.append(len(ent))
            scans.append(ns)
            pops.append(len(ent))
        assert np.allclose(best, final)
        acc["sbl"].append(len(sbl))
        acc["bl"].append(len(BL))
        acc["llen"].append(np.mean(llen))
        acc["llen_max"].append(np.max(llen))
        acc["scans"].append(np.mean(scans))
        acc["scans_max"].append(np.max(scans))
        acc["pops_max"].append(np.max(pops))
    return {k: float(np.mean(v)) for k, v in acc.items()}


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

    for nq, nc in ((2048, 16384), (16384, 2048), (4096, 4096), (16384, 16384)):
        A, B = gen(nq), gen(nc)
        As, Bs = sm.sort_str_hist(A, 64), sm.sort_str_hist(B, 64)
        for nseed in (0,):
            for srt in (False, True):
                r = tile_sim(As, Bs, rng=np.random.RandomState(1), sort_lists=srt, nseed_sb=nseed)
                print(f"{kind} {nq} in {nc} seeds {nseed} sorted {int(srt)}: seed/final dist {r['seed_ratio']:.2f} | SB list {r['sbl']:5.1f} block list {r['bl']:5.1f} "
                      f"S0 {r.get('s0', 0):4.1f} | per query: list {r['llen']:4.1f} (max {r['llen_max']:4.1f}) scans {r['scans']:4.1f} (max in tile {r['scans_max']:4.1f}) pops max {r['pops_max']:4.1f}")
