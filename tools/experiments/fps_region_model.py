"""How many of fps_sorted_kernel's 16 waves a new sample touches per iteration, as a function of how the sorted cloud's 64-point
tiles are dealt to the waves: the product (a wave = 1024 CONSECUTIVE sorted points: 16 tiles in a row along the sort's z runs) against
waves made of 16 tiles that are close together (a kd split of the tile centres).  numpy model of the kernel's own rule on C3's cloud:
a lane (16 consecutive sorted points) is touched when the box bound of the new sample is below the lane's largest running minimum; a
wave scans when any of its lanes is touched.  usage: python tools/experiments/fps_region_model.py"""
import numpy as np

rng = np.random.RandomState(100)
n, m = 16384, 1024
P = rng.random_sample((32, n, 3)).astype(np.float32)[0]


def str_order(P, SS=6):
    """sort-tile-recursive order as nnp_sort_reg makes it (equal-count slabs in x, strips in y per slab, z inside, boustrophedon)"""
    n = len(P)
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


def kd_groups(centres, ngroups=16):
    """tile ids in 16 groups of equal size by median splits along the longest axis"""
    groups = [np.arange(len(centres))]
    while len(groups) < ngroups:
        nxt = []
        for g in groups:
            c = centres[g]
            ax = np.argmax(c.max(0) - c.min(0))
            o = g[np.argsort(c[:, ax], kind="stable")]
            nxt += [o[: len(o) // 2], o[len(o) // 2:]]
        groups = nxt
    return np.concatenate(groups)


def simulate(P, lane_pts):
    """lane_pts: (1024 lanes, 16) point ids; wave w = lanes 64 w .. 64 w + 63.  -> mean touched waves / lanes per iteration"""
    X = P[lane_pts]                       # (1024, 16, 3)
    lo, hi = X.min(1), X.max(1)
    td = np.full(lane_pts.shape, 1e38, np.float32)
    old = 0
    tw = tl = 0
    for j in range(1, m):
        s = P[old]
        g = np.maximum(np.maximum(lo - s, s - hi), 0)
        lb = (g * g).sum(1)
        lmx = td.max(1)
        touched = lb < lmx
        wt = touched.reshape(16, 64).any(1)
        tw += wt.sum(); tl += touched.sum()
        rows = np.repeat(wt, 64)           # a touched wave scans all its lanes
        d = ((X[rows] - s) ** 2).sum(-1)
        td[rows] = np.minimum(td[rows], d)
        flat = td.reshape(-1)
        old = lane_pts.reshape(-1)[int(np.argmax(flat))]
    return tw / (m - 1), tl / (m - 1)


order = str_order(P)
base = order.reshape(1024, 16)
print("product (1024 consecutive sorted points per wave): touched waves %.2f of 16, touched lanes %.1f of 1024 per iteration" % simulate(P, base))
tiles = order.reshape(256, 64)
cen = P[tiles].mean(1)
perm = kd_groups(cen)
kd = tiles[perm].reshape(1024, 16)
print("kd-grouped tiles (16 nearby tiles per wave):         touched waves %.2f of 16, touched lanes %.1f of 1024 per iteration" % simulate(P, kd))
# lanes as cubes too: inside a wave's 1024 points, 64 lanes of 16 points by kd split of the points
def kd_points(ids, leaf=16):
    groups = [ids]
    while len(groups[0]) > leaf:
        nxt = []
        for g in groups:
            c = P[g]
            ax = np.argmax(c.max(0) - c.min(0))
            o = g[np.argsort(c[:, ax], kind="stable")]
            nxt += [o[: len(o) // 2], o[len(o) // 2:]]
        groups = nxt
    return np.stack(groups)
wave_ids = tiles[perm].reshape(16, 1024)
full = np.concatenate([kd_points(w) for w in wave_ids])
print("kd waves AND kd lanes (16-point cubes):              touched waves %.2f of 16, touched lanes %.1f of 1024 per iteration" % simulate(P, full))
whole = kd_points(np.arange(n), 16)   # pure kd tree over the cloud: 1024 leaves in tree order = waves of 64 consecutive leaves
print("pure kd tree over the whole cloud:                   touched waves %.2f of 16, touched lanes %.1f of 1024 per iteration" % simulate(P, whole))

# ---- other orders the sort could emit for an FPS handle ----
def str_general(P, sx, sy, snake=True):
    order = []
    xs = np.argsort(P[:, 0], kind="stable")
    for si, slab in enumerate(np.array_split(xs, sx)):
        ys = slab[np.argsort(P[slab, 1], kind="stable")]
        strips = np.array_split(ys, sy)
        if snake and si & 1:
            strips = strips[::-1]
        for ti, strip in enumerate(strips):
            zs = strip[np.argsort(P[strip, 2], kind="stable")]
            order.append(zs[::-1] if snake and ((si * sy + ti) & 1) else zs)
    return np.concatenate(order)

for sx, sy in ((4, 4), (8, 8), (4, 8), (8, 4), (16, 16)):
    o = str_general(P, sx, sy).reshape(1024, 16)
    print(f"STR {sx:2d} x {sy:2d}:                                         touched waves %.2f of 16, touched lanes %.1f of 1024 per iteration" % simulate(P, o))

def nested(ids, plan):
    """plan: list of (axis, parts): split recursively, equal counts"""
    if not plan:
        return [ids]
    ax, parts = plan[0]
    o = ids[np.argsort(P[ids, ax], kind="stable")]
    out = []
    for piece in np.array_split(o, parts):
        out += nested(piece, plan[1:])
    return out

for name, plan in (("x4 y4 z4 | x4 y4 z4 (two-level grid)", [(0, 4), (1, 4), (2, 4), (0, 4), (1, 4), (2, 1)]),
                   ("x2 y2 z4 | x4 y4 z4 (16 wave cells, 64 lane cells)", [(0, 2), (1, 2), (2, 4), (0, 4), (1, 4), (2, 4)]),
                   ("x4 y4 | z4 x2 y2 | z.. ", [(0, 4), (1, 4), (2, 4), (0, 2), (1, 2), (2, 4)])):
    o = np.concatenate(nested(np.arange(n), plan))
    print(f"{name:52s} touched waves %.2f of 16, touched lanes %.1f of 1024 per iteration" % simulate(P, o.reshape(1024, 16)))

# ---- what nnp_sort_reg's own tables can emit: 8 equal-mass slabs in x, 8 strips in y per slab, a GLOBAL equal-mass z rank (16 levels) ----
def grid_order(P, variant):
    n = len(P)
    xr = np.argsort(np.argsort(P[:, 0], kind="stable"), kind="stable")
    slab = (xr * 8 // n)
    strip = np.zeros(n, int)
    for s_ in range(8):
        ids = np.where(slab == s_)[0]
        yr = np.argsort(np.argsort(P[ids, 1], kind="stable"), kind="stable")
        strip[ids] = yr * 8 // len(ids)
    zr = np.argsort(np.argsort(P[:, 2], kind="stable"), kind="stable")
    z16 = zr * 16 // n
    zlow = (zr * 512 // n) & 31
    if variant == "wave2x2x4 lane4x4x4":
        wc = (slab >> 2) << 3 | (strip >> 2) << 2 | (z16 >> 2)
        lc = (slab & 3) << 4 | (strip & 3) << 2 | (z16 & 3)
    key = (wc << 11) | (lc << 5) | zlow
    return np.argsort(key, kind="stable")

o = grid_order(P, "wave2x2x4 lane4x4x4").reshape(1024, 16)
print("sort's own tables, key = wave cell 2x2x4 | lane cell 4x4x4 | z: touched waves %.2f of 16, touched lanes %.1f of 1024 per iteration" % simulate(P, o))
