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

# ---- what an iteration costs is the busiest SIMD, not the sum: wave w sits on SIMD w % 4 ----
def simulate_simd(P, lane_pts, wave_of_cell=None):
    X = P[lane_pts]
    lo, hi = X.min(1), X.max(1)
    td = np.full(lane_pts.shape, 1e38, np.float32)
    old = 0
    mx = tot = 0
    for j in range(1, m):
        s = P[old]
        g = np.maximum(np.maximum(lo - s, s - hi), 0)
        touched = (g * g).sum(1) < td.max(1)
        wt = touched.reshape(16, 64).any(1)
        ids = np.where(wt)[0] if wave_of_cell is None else wave_of_cell[np.where(wt)[0]]
        mx += np.bincount(ids % 4, minlength=4).max() if len(ids) else 0
        tot += wt.sum()
        rows = np.repeat(wt, 64)
        d = ((X[rows] - s) ** 2).sum(-1)
        td[rows] = np.minimum(td[rows], d)
        old = lane_pts.reshape(-1)[int(np.argmax(td.reshape(-1)))]
    return tot / (m - 1), mx / (m - 1)

print("busiest SIMD per iteration (touched waves on it), wave w on SIMD w % 4:")
print("  product order:            touched %.2f, busiest SIMD %.2f" % simulate_simd(P, base))
print("  sort's-tables grid order: touched %.2f, busiest SIMD %.2f" % simulate_simd(P, o))
# the grid order with the 16 cells dealt to the SIMDs so that face neighbours differ: cell (x, y, z) of 2 x 2 x 4 -> SIMD (x + 2 y + z) % 4 ... wave = any of that SIMD's four
cells = np.arange(16)
cx, cy, cz = cells >> 3, (cells >> 2) & 1, cells & 3
simd = (cx + 2 * cy + cz) % 4
wave_of_cell = np.zeros(16, int)
for s_ in range(4):
    wave_of_cell[np.where(simd == s_)[0]] = s_ + 4 * np.arange((simd == s_).sum())
print("  ... cells dealt so that neighbours sit on different SIMDs: touched %.2f, busiest SIMD %.2f" % simulate_simd(P, o, wave_of_cell))

# ---- the same question for the PRODUCT's order: which of its 16 chunks of 1024 sorted points should share a SIMD? ----
def touched_sets(P, lane_pts):
    X = P[lane_pts]
    lo, hi = X.min(1), X.max(1)
    td = np.full(lane_pts.shape, 1e38, np.float32)
    old = 0
    out = []
    for j in range(1, m):
        s = P[old]
        g = np.maximum(np.maximum(lo - s, s - hi), 0)
        wt = ((g * g).sum(1) < td.max(1)).reshape(16, 64).any(1)
        out.append(wt.copy())
        rows = np.repeat(wt, 64)
        d = ((X[rows] - s) ** 2).sum(-1)
        td[rows] = np.minimum(td[rows], d)
        old = lane_pts.reshape(-1)[int(np.argmax(td.reshape(-1)))]
    return np.array(out)

def busiest(T, simd_of_chunk):
    return np.stack([(T[:, simd_of_chunk == s_]).sum(1) for s_ in range(4)], 1).max(1).mean()

T = touched_sets(P, base)
assign = np.arange(16) % 4
best = busiest(T, assign)
rs = np.random.RandomState(0)
for it in range(4000):  # local search: swap the SIMDs of two chunks
    i, j = rs.randint(0, 16, 2)
    if assign[i] == assign[j]:
        continue
    assign[i], assign[j] = assign[j], assign[i]
    v = busiest(T, assign)
    if v < best:
        best = v
    else:
        assign[i], assign[j] = assign[j], assign[i]
print("product order, chunks dealt to SIMDs by local search on THIS cloud: busiest SIMD %.2f (round-robin %.2f); assignment %s" % (best, busiest(T, np.arange(16) % 4), assign.tolist()))
# does one cloud's assignment carry to another?
P2 = rng.random_sample((n, 3)).astype(np.float32)
base2 = str_order(P2).reshape(1024, 16)
T2 = touched_sets(P2, base2)
print("  the same assignment on another uniform cloud: busiest %.2f (round-robin %.2f)" % (busiest(T2, assign), busiest(T2, np.arange(16) % 4)))
