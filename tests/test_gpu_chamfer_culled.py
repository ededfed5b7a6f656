"""GPU parity of the culled Chamfer sweep (rfnet_amd/csrc/nn_pruned.hip, RF_NN_CULLED).

The culled sweep must return exactly what the dense sweep and the oracle return -- distances
bit for bit, and the LOWEST ORIGINAL INDEX on ties although candidates are visited in Hilbert
order -- on every input: it only skips pairs whose bounding-box lower bound is strictly above
the running minimum.  Small cases are checked against the oracle (NmDistanceKernel restated,
oracle/rfops_oracle.c); large ones against the dense sweep (itself oracle-checked in
test_gpu_chamfer.py) plus oracle slices.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

NAMES = ("dist1", "idx1", "dist2", "idx2")


def _run(xyz1, xyz2, mode, stats=None):
    from rfnet_amd import _raw
    out = _raw.nn_distance(torch.from_numpy(xyz1).cuda(), torch.from_numpy(xyz2).cuda(), mode=mode, stats=stats)
    return [t.cpu().numpy() for t in out]


def _same(got, exp, what):
    for g, e, name in zip(got, exp, NAMES):
        assert g.dtype == e.dtype and g.shape == e.shape
        assert np.array_equal(g, e), f"{what} {name}: {np.sum(g != e)} mismatches of {g.size}"


def _check_vs_oracle(orc, a, c):
    got = _run(a, c, "culled")
    _same(got, orc.nn_distance(a, c), "culled vs oracle")
    return got


@pytest.mark.parametrize("b,n,m", [(1, 1, 1), (2, 7, 5), (3, 1, 300), (2, 300, 1), (3, 256, 1024),
                                   (2, 1025, 4097), (1, 5000, 300), (5, 64, 3000), (33, 100, 17),
                                   (2, 63, 65), (1, 64, 64), (70, 130, 129)])
def test_random_shapes_vs_oracle(orc, b, n, m):
    rng = np.random.RandomState(b * 1000 + n + m)
    _check_vs_oracle(orc, rng.randn(b, n, 3).astype(np.float32), rng.randn(b, m, 3).astype(np.float32))


def test_golden_ragged_and_dup(orc, golden):
    g = golden("nn_distance_ragged")
    for tag in ("a", "b", "dup"):
        got = _check_vs_oracle(orc, g[f"{tag}_xyz1"], g[f"{tag}_xyz2"])
        assert np.array_equal(got[1], g[f"{tag}_ref_idx1"]) and np.array_equal(got[3], g[f"{tag}_ref_idx2"])
    g = golden("nn_distance_c1")
    got = _check_vs_oracle(orc, g["xyz1"], g["xyz2"])
    assert np.array_equal(got[1], g["ref_idx1"]) and np.array_equal(got[3], g["ref_idx2"])


def test_many_exact_ties_vs_oracle(orc):
    rng = np.random.RandomState(1)
    grid = rng.randint(0, 4, size=(2, 2000, 3)).astype(np.float32)  # lattice: masses of ties across blocks
    _check_vs_oracle(orc, grid[:, :900], grid[:, 900:])
    same = np.ones((1, 700, 3), np.float32)  # all distances 0: every idx must be 0
    got = _check_vs_oracle(orc, same, same[:, :333])
    assert (got[1] == 0).all() and (got[3] == 0).all()
    # duplicated points as resample_pcd (data_util.py:8-13) makes them: draws with replacement
    base = rng.randn(3, 500, 3).astype(np.float32)
    dup = np.take_along_axis(base, rng.randint(0, 500, size=(3, 3000, 1)), 1)
    _check_vs_oracle(orc, dup, base)
    _check_vs_oracle(orc, base, dup)
    _check_vs_oracle(orc, dup, dup[:, ::-1].copy())


def test_degenerate_geometry_vs_oracle(orc):
    rng = np.random.RandomState(2)
    line = np.zeros((2, 1500, 3), np.float32)
    line[..., 0] = rng.randn(2, 1500)  # collinear: two axes have zero extent
    _check_vs_oracle(orc, line[:, :700], line[:, 700:])
    out = rng.randn(1, 3000, 3).astype(np.float32)
    out[0, :3] *= 1e6  # far outliers stretch the bounding box
    _check_vs_oracle(orc, out[:, :1000], out[:, 1000:])
    big = (rng.randn(1, 500, 3) * 1e18).astype(np.float32)
    big[0, :5] = 3e38
    c = (rng.randn(1, 300, 3) * 1e18).astype(np.float32)
    c[0, :3] = -3e38  # d2 overflows to +inf for these pairs
    _check_vs_oracle(orc, big, c)
    tiny = (rng.randn(1, 400, 3) * 1e-30).astype(np.float32)  # d2 underflows: everything ties at 0 or denormals
    _check_vs_oracle(orc, tiny[:, :150], tiny[:, 150:])


@pytest.mark.parametrize("kind", ["randn", "uniform", "sphere", "lattice", "dup", "collapsed"])
# (80 x 3500^2: one wave per group in both directions, one-wave workgroups; 70 x 700 x 5000: shared
# groups in one direction, single-wave groups packed 4 to a workgroup in the other)
@pytest.mark.parametrize("b,n,m", [(4, 2048, 16384), (2, 16384, 16384), (40, 1000, 3000), (80, 3500, 3500),
                                   (70, 700, 5000)])
def test_culled_equals_dense(orc, kind, b, n, m):
    """All four outputs identical to the dense sweep's, plus an oracle slice."""
    rng = np.random.RandomState(len(kind) * 100 + b + n)

    def cloud(k):
        if kind == "randn":
            return rng.randn(b, k, 3).astype(np.float32)
        if kind == "uniform":
            return rng.random_sample((b, k, 3)).astype(np.float32)
        if kind == "sphere":
            x = rng.randn(b, k, 3)
            return (x / np.linalg.norm(x, axis=-1, keepdims=True)).astype(np.float32)
        if kind == "lattice":
            return rng.randint(0, 12, size=(b, k, 3)).astype(np.float32)
        if kind == "collapsed":  # ~120 spots of near-copies (the untrained network's output): the sort's crowded flag sends the
            # few-groups direction against such a candidate cloud through the shared-group sweep instead of the quad tiles
            # (consecutive points mostly share a spot, as the network's outputs do: that coherence is what the sort's test sees)
            spots = rng.random_sample((b, 120, 3)) - 0.5
            sid = np.where(rng.random_sample((b, k)) < 0.95, (np.arange(k) * 120 // k)[None], rng.randint(0, 120, (b, k)))
            return (spots[np.arange(b)[:, None], sid] + 6e-6 * rng.randn(b, k, 3)).astype(np.float32)
        base = rng.randn(b, max(k // 5, 1), 3).astype(np.float32)
        return np.take_along_axis(base, rng.randint(0, base.shape[1], size=(b, k, 1)), 1)

    a, c = cloud(n), cloud(m)
    stats = []
    got = _run(a, c, "culled", stats)
    _same(got, _run(a, c, "dense"), f"{kind} culled vs dense")
    e = orc.nn_distance(a[:1, :200], c[:1])
    assert np.array_equal(got[0][0, :200], e[0][0]) and np.array_equal(got[1][0, :200], e[1][0])
    # the sweep did cull: fewer pairs evaluated than b*n*m in each direction (not for the lattice /
    # duplicate clouds, where most boxes overlap most queries)
    if (b, n, m) == (4, 2048, 16384):
        # both directions have few groups here: quad-per-query tiles (16 pairs per counted scan) -- unless the CANDIDATE
        # cloud is crowded, which the sort flags and the sweep answers with the shared-group path (1024 pairs per scan)
        # (direction 0's candidates are the 16384-point cloud; a 2048-point cloud feeds the sort's test from too few waves to be flagged)
        # (the lattice / duplicate clouds may or may not be flagged -- 12 values per axis crowd a wave's x bins too; either path is exact)
        if kind in ("randn", "uniform", "sphere", "collapsed"):
            assert stats[14] == (1024 if kind == "collapsed" else 16) and stats[15] in (16, 1024), (kind, stats[14], stats[15])
    if kind in ("randn", "uniform", "sphere"):
        pairs = [stats[12], stats[13]]  # directed pairs evaluated per direction, summed by the kernel
        assert pairs[0] > 0 and pairs[1] > 0 and pairs[0] % 16 == 0 and pairs[1] % 16 == 0
        assert pairs[0] < 0.6 * b * n * m and pairs[1] < 0.6 * b * n * m, (stats, b * n * m)


@pytest.mark.parametrize("b,n,m,kind", [
    (2, 2048, 16384, "randn"),     # C2's launch shape: 4 waves per group on the 2048 side, packed one-wave groups on the other
    (1, 16384, 16384, "randn"),    # north-star shape: register-resident sort of both, one wave per group, key lists in registers
    (1, 16384, 16384, "dup"),      # the same with every point repeated ~5x: the tie machinery at full size
    (1, 3000, 16384, "uniform"),   # the model's merge_layer shape (ragged 3000: padded superblocks)
    (1, 20000, 17000, "randn"),    # beyond the register-resident sort (> 16384 points): the streaming sort, LDS key lists
])
def test_full_cloud_oracle_at_large_sizes(orc, b, n, m, kind):
    """No proxy: EVERY output of the culled sweep against the C oracle over the whole clouds, one case per
    launch shape at >= 16384 points (the oracle needs a few seconds per case: up to 3.4e8 pairs per direction)."""
    rng = np.random.RandomState(n + m)

    def cloud(k):
        if kind == "randn":
            return rng.randn(b, k, 3).astype(np.float32)
        if kind == "uniform":
            return (rng.random_sample((b, k, 3)) - 0.5).astype(np.float32)
        base = rng.randn(b, max(k // 5, 1), 3).astype(np.float32)
        return np.take_along_axis(base, rng.randint(0, base.shape[1], size=(b, k, 1)), 1)

    a, c = cloud(n), cloud(m)
    _same(_run(a, c, "culled"), orc.nn_distance(a, c), f"{kind} {b}x{n}x{m} culled vs oracle (all points)")


def test_c2_and_auto_mode(orc):
    """BASELINE.json configs[1] through the public op (auto mode picks the culled sweep at this size)."""
    from tf_ops.CD.tf_nndistance import nn_distance
    rng = np.random.RandomState(100)
    a = rng.randn(32, 2048, 3).astype(np.float32)
    c = rng.randn(32, 16384, 3).astype(np.float32)
    auto = [t.cpu().numpy() for t in nn_distance(torch.from_numpy(a).cuda(), torch.from_numpy(c).cuda())]
    _same(auto, _run(a, c, "dense"), "auto vs dense")
    _same(auto, _run(a, c, "culled"), "auto vs culled")
    for bi in (0, 31):
        e = orc.nn_distance(a[bi:bi + 1], c[bi:bi + 1])
        for k in range(4):
            assert np.array_equal(auto[k][bi], e[k][0])


def test_large_clouds_65536(orc):
    rng = np.random.RandomState(11)
    a = rng.randn(1, 65536, 3).astype(np.float32)
    c = rng.randn(1, 32768, 3).astype(np.float32)
    got = _run(a, c, "culled")
    _same(got, _run(a, c, "dense"), "65536 culled vs dense")
    from rfnet_amd._lib import RfopsError
    with pytest.raises(RfopsError):
        _run(np.zeros((1, 65537, 3), np.float32), c, "culled")  # beyond the culled sweep's limit: explicit error
    _run(np.zeros((1, 65537, 3), np.float32), c[:, :100], "auto")  # auto falls back to the dense sweep


@pytest.mark.parametrize("seed", range(40))
def test_fuzz_shapes_and_kinds_vs_oracle(orc, seed):
    """Seeded random shapes / cloud kinds through the culled sweep, every output against the oracle."""
    rng = np.random.RandomState(7000 + seed)
    b = int(rng.randint(1, 9))
    n, m = (int(v) for v in rng.randint(1, 2600, size=2))
    kind = ("randn", "uniform", "lattice", "dup", "plane", "clusters")[seed % 6]

    def cloud(k):
        if kind == "randn":
            return rng.randn(b, k, 3)
        if kind == "uniform":
            return rng.random_sample((b, k, 3)) * 10 - 5
        if kind == "lattice":
            return rng.randint(0, 7, size=(b, k, 3)).astype(np.float64) * 0.25
        if kind == "dup":
            base = rng.randn(b, max(k // 4, 1), 3)
            return np.take_along_axis(base, rng.randint(0, base.shape[1], size=(b, k, 1)), 1)
        if kind == "plane":
            x = rng.randn(b, k, 3)
            x[..., 2] = 0.5  # zero extent on one axis
            return x
        centres = rng.randn(b, 5, 3) * 20
        return centres[np.arange(b)[:, None], rng.randint(0, 5, size=(b, k))] + rng.randn(b, k, 3) * 0.01

    _check_vs_oracle(orc, cloud(n).astype(np.float32), cloud(m).astype(np.float32))


@pytest.mark.parametrize("n,m", [(300, 700), (3000, 5000), (4096, 4096)])
def test_non_finite_inputs_policy(orc, n, m):
    """NaN coordinates.  The reference's result depends on WHERE the NaN sits (its first candidate of
    a 512-tile is taken unconditionally, tf_nndistance_g.cu:27-31,118): candidate 0 being NaN poisons
    every query, a NaN at another tile start hides that tile, elsewhere it is ignored.  The policy
    here (INTEGRATION.md "Non-finite inputs"), identical in both sweeps and pinned by this test:
      * a point with a NaN coordinate gets (dist = NaN, idx = 0) -- what the reference returns for it;
      * a NaN point is never anybody's nearest neighbour (the reference: only when it sits at a tile
        start), i.e. every other point gets exactly what it would get with the NaN points removed;
      * every index is in range.
    Infinite coordinates need no rule: d2 is +inf (or NaN for inf - inf) and strict '<' never takes it."""
    rng = np.random.RandomState(5)
    a = rng.randn(2, n, 3).astype(np.float32)
    c = rng.randn(2, m, 3).astype(np.float32)
    a[0, 17] = np.nan
    a[0, 5, 1] = np.nan
    c[0, 0, 0] = np.nan          # the reference's poison position
    c[0, 123, 1] = np.nan
    c[1, m - 1] = np.nan
    c[1, 512 % m, 2] = np.nan    # a tile start
    outs = {mode: _run(a, c, mode) for mode in ("dense", "culled")}
    for x, y in zip(outs["dense"], outs["culled"]):
        assert np.array_equal(x, y, equal_nan=True)
    d1, i1, d2, i2 = outs["culled"]
    nan1, nan2 = np.isnan(a).any(-1), np.isnan(c).any(-1)
    assert np.isnan(d1[nan1]).all() and (i1[nan1] == 0).all()
    assert np.isnan(d2[nan2]).all() and (i2[nan2] == 0).all()
    assert i1.min() >= 0 and i1.max() < m and i2.min() >= 0 and i2.max() < n
    for bi in range(2):
        keep1, keep2 = np.flatnonzero(~nan1[bi]), np.flatnonzero(~nan2[bi])
        e = orc.nn_distance(a[bi:bi + 1, keep1], c[bi:bi + 1, keep2])
        assert np.array_equal(d1[bi, keep1], e[0][0]) and np.array_equal(i1[bi, keep1], keep2[e[1][0]])
        assert np.array_equal(d2[bi, keep2], e[2][0]) and np.array_equal(i2[bi, keep2], keep1[e[3][0]])
    # infinities: every output stays in range and both sweeps agree
    a2, c2 = a.copy(), c.copy()
    a2[np.isnan(a2)] = np.inf
    c2[np.isnan(c2)] = -np.inf
    o1, o2 = _run(a2, c2, "dense"), _run(a2, c2, "culled")
    fin1, fin2 = np.isfinite(a2).all(-1), np.isfinite(c2).all(-1)
    assert np.array_equal(o1[0][fin1], o2[0][fin1]) and np.array_equal(o1[1][fin1], o2[1][fin1])
    assert np.array_equal(o1[2][fin2], o2[2][fin2]) and np.array_equal(o1[3][fin2], o2[3][fin2])
    for o in (o1, o2):
        assert o[1].min() >= 0 and o[1].max() < m and o[3].min() >= 0 and o[3].max() < n


def test_stress_ties_against_dense():
    """150 seeded clouds built to tie (coarse lattices, duplicates, mirrored halves, points on a few
    planes): every output of the culled sweep equal to the dense sweep's, bit for bit."""
    bad = []
    for seed in range(150):
        rng = np.random.RandomState(9000 + seed)
        b = int(rng.randint(1, 4))
        n, m = (int(v) for v in rng.randint(2048, 5000, size=2))
        kind = seed % 5
        if kind == 0:
            a, c = (rng.randint(0, 9, size=(b, k, 3)).astype(np.float32) * 0.5 for k in (n, m))
        elif kind == 1:
            base = rng.randn(b, 600, 3).astype(np.float32)
            a, c = (np.take_along_axis(base, rng.randint(0, 600, size=(b, k, 1)), 1) for k in (n, m))
        elif kind == 2:
            h = rng.randn(b, m // 2, 3).astype(np.float32)
            c = np.concatenate([h, h * np.array([-1, 1, 1], np.float32)], 1)  # mirrored: equidistant pairs
            a = np.zeros((b, n, 3), np.float32)
            a[..., 1:] = rng.randn(b, n, 2)  # queries on the mirror plane
        elif kind == 3:
            a, c = (rng.randn(b, k, 3).astype(np.float32) for k in (n, m))
            a[..., 2] = np.round(a[..., 2])
            c[..., 2] = np.round(c[..., 2])
        else:
            a = rng.randn(b, n, 3).astype(np.float32)
            c = np.concatenate([a[:, : m // 2], a[:, : m - m // 2]], 1)  # the queries themselves, twice over
        a, c = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(c, np.float32)
        got, ref = _run(a, c, "culled"), _run(a, c, "dense")
        if not all(np.array_equal(g, r) for g, r in zip(got, ref)):
            bad.append(seed)
    assert not bad, f"culled != dense for seeds {bad}"


def test_two_streams_concurrently():
    """The culled sweep keeps no state outside the caller's workspace: two streams, each with its
    own inputs (the wrapper gives each stream its own scratch), interleaved launches, results equal
    to the sequential ones."""
    from rfnet_amd import _raw
    rng = np.random.RandomState(77)
    xs = [(torch.from_numpy(rng.randn(6, 2500, 3).astype(np.float32)).cuda(),
           torch.from_numpy(rng.randn(6, 9000, 3).astype(np.float32)).cuda()) for _ in range(2)]
    ref = [[t.clone() for t in _raw.nn_distance(a, c, mode="culled")] for a, c in xs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [None, None]
    for rep in range(5):
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                outs[i] = _raw.nn_distance(*xs[i], mode="culled")
    for st in streams:
        st.synchronize()
    for i in range(2):
        for g, e in zip(outs[i], ref[i]):
            assert torch.equal(g, e)
