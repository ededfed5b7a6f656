"""GPU parity: farthest_point_sample, gather_point(+grad), query_ball_point, group_point(+grad),
three_nn, three_interpolate(+grad).  All index outputs are bit-exact vs the oracle; copies are
bit-exact; atomically accumulated gradients within rel 1e-5 / abs 1e-6."""
import numpy as np
import pytest
import torch

from conftest import assert_rel

pytestmark = pytest.mark.gpu


def cu(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


# ------------------------------------------------------------------ FPS / gather
def test_fps_golden_and_tie_rule(orc, golden):
    from tf_ops.sampling.tf_sampling import farthest_point_sample, gather_point, gather_point_grad
    g = golden("sampling")
    idx = farthest_point_sample(64, cu(g["inp"]))
    assert idx.dtype == torch.int32
    assert np.array_equal(idx.cpu().numpy(), g["cuda_fps_idx"])
    t = farthest_point_sample(40, cu(g["tie_inp"])).cpu().numpy()
    assert np.array_equal(t, g["cuda_tie_fps_idx"])
    out = gather_point(cu(g["inp"]), idx)
    assert np.array_equal(out.cpu().numpy(), g["cuda_gathered"])
    gi = gather_point_grad(cu(g["inp"]), idx, cu(g["grad_out"])).cpu().numpy()
    assert_rel(gi, g["cuda_grad_inp"], 1e-5, 1e-6)


@pytest.mark.parametrize("b,n,m", [(1, 1, 1), (2, 1, 4), (3, 17, 17), (2, 17, 40), (2, 512, 64),
                                   (3, 513, 100), (2, 1024, 128), (2, 1500, 77), (2, 3000, 32),
                                   (1, 4096, 300), (2, 8000, 64), (2, 16384, 64), (33, 700, 9)])
def test_fps_random_shapes(orc, b, n, m):
    from tf_ops.sampling.tf_sampling import farthest_point_sample
    rng = np.random.RandomState(n * 7 + m)
    p = rng.rand(b, n, 3).astype(np.float32)
    got = farthest_point_sample(m, cu(p)).cpu().numpy()
    assert np.array_equal(got, orc.farthest_point_sample(m, p))


def test_fps_ties_on_lattice(orc):
    from tf_ops.sampling.tf_sampling import farthest_point_sample
    rng = np.random.RandomState(2)
    for n in (600, 2048, 5000):
        p = rng.randint(0, 3, size=(2, n, 3)).astype(np.float32)  # 27 distinct sites: ties galore
        got = farthest_point_sample(50, cu(p)).cpu().numpy()
        assert np.array_equal(got, orc.farthest_point_sample(50, p)), n


def test_fps_memory_fallback_beyond_16384(orc):
    from tf_ops.sampling.tf_sampling import farthest_point_sample
    p = np.random.RandomState(8).rand(2, 20000, 3).astype(np.float32)
    got = farthest_point_sample(48, cu(p)).cpu().numpy()
    assert np.array_equal(got, orc.farthest_point_sample(48, p))


def test_c3_config_fps_ballquery_group(orc):
    """BASELINE.json configs[2]: FPS 16384->1024 + query_ball_point(r=0.1,K=32) + group_point, B=32.
    Oracle on 3 of the 32 clouds for FPS (the first, the last and one drawn at random; serial 1023 x 16384 updates
    each), on all for the rest."""
    from tf_ops.grouping.tf_grouping import group_point, query_ball_point
    from tf_ops.sampling.tf_sampling import farthest_point_sample, gather_point
    xyz = np.random.RandomState(100).random_sample((32, 16384, 3)).astype(np.float32)
    t = cu(xyz)
    idx = farthest_point_sample(1024, t)
    hi = idx.cpu().numpy()
    for bi in (0, 31, int(np.random.RandomState().randint(1, 31))):
        assert np.array_equal(hi[bi], orc.farthest_point_sample(1024, xyz[bi:bi + 1])[0]), bi
    # size-independent properties on all clouds: a permutation prefix, first index 0
    assert (hi[:, 0] == 0).all()
    assert all(len(set(r.tolist())) == 1024 for r in hi)
    new_xyz = gather_point(t, idx)
    assert np.array_equal(new_xyz.cpu().numpy(), orc.gather_point(xyz, hi))
    qidx, cnt = query_ball_point(0.1, 32, t, new_xyz)
    oi, oc = orc.query_ball_point(0.1, 32, xyz, new_xyz.cpu().numpy())
    assert np.array_equal(cnt.cpu().numpy(), oc)
    assert np.array_equal(qidx.cpu().numpy(), oi)
    grouped = group_point(t, qidx)
    assert tuple(grouped.shape) == (32, 1024, 32, 3)
    assert np.array_equal(grouped.cpu().numpy(), orc.group_point(xyz, oi))


# ------------------------------------------------------------------ ball query / grouping
@pytest.mark.parametrize("tag,radius,ns", [("r1_k32", 0.1, 32), ("r3_k64", 0.3, 64)])
def test_grouping_golden(golden, tag, radius, ns):
    from tf_ops.grouping.tf_grouping import group_point, group_point_grad, query_ball_point
    g = golden("grouping")
    idx, cnt = query_ball_point(radius, ns, cu(g["xyz1"]), cu(g["xyz2"]))
    assert np.array_equal(idx.cpu().numpy(), g[f"{tag}_ref_idx"])
    assert np.array_equal(cnt.cpu().numpy(), g[f"{tag}_cuda_pts_cnt"])
    grouped = group_point(cu(g["points"]), idx)
    assert np.array_equal(grouped.cpu().numpy(), g[f"{tag}_ref_grouped"])
    gp = group_point_grad(cu(g["points"]), idx, cu(g[f"{tag}_grad_out"])).cpu().numpy()
    assert_rel(gp, g[f"{tag}_ref_grad_points"], 1e-5, 1e-6)


@pytest.mark.parametrize("b,n,m,ns,r", [(1, 1, 1, 1, 0.5), (2, 63, 10, 8, 0.3), (2, 65, 33, 70, 0.4),
                                        (3, 1000, 257, 16, 0.15), (2, 5000, 100, 300, 0.2),
                                        (2, 200, 50, 4, 1e-6)])
def test_query_ball_random(orc, b, n, m, ns, r):
    from tf_ops.grouping.tf_grouping import query_ball_point
    rng = np.random.RandomState(n + m + ns)
    ds = rng.rand(b, n, 3).astype(np.float32)
    q = rng.rand(b, m, 3).astype(np.float32)
    q[:, : m // 2] = ds[:, : m // 2] if n >= m // 2 else q[:, : m // 2]
    idx, cnt = query_ball_point(r, ns, cu(ds), cu(q))
    oi, oc = orc.query_ball_point(r, ns, ds, q, fill=0)  # wrapper zero-fills empty rows
    assert np.array_equal(cnt.cpu().numpy(), oc)
    assert np.array_equal(idx.cpu().numpy(), oi)


def test_query_ball_radius_boundary_is_in_distance_domain(orc):
    """Points whose sqrt_rn(d2) straddles the radius by one ulp: the compare must happen after
    a correctly rounded sqrt, not in the squared domain."""
    from tf_ops.grouping.tf_grouping import query_ball_point
    rng = np.random.RandomState(4)
    r = np.float32(0.1)
    dirs = rng.randn(1, 4096, 3)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    scale = r * (1 + (rng.rand(1, 4096, 1) - 0.5) * 4e-7)  # within +-2e-7 relative of r
    ds = (dirs * scale).astype(np.float32)
    q = np.zeros((1, 1, 3), np.float32)
    idx, cnt = query_ball_point(float(r), 4096, cu(ds), cu(q))
    oi, oc = orc.query_ball_point(float(r), 4096, ds, q)
    assert 100 < oc[0, 0] < 4000  # the set really straddles the boundary
    assert np.array_equal(cnt.cpu().numpy(), oc) and np.array_equal(idx.cpu().numpy(), oi)


@pytest.mark.parametrize("ns", [8, 48, 80])  # lanes kernel (<=32 / <=64 samples) and the wave-per-8-queries kernel
def test_query_ball_nan_is_a_hit_like_the_reference(orc, ns):
    """max(sqrtf(NaN), 1e-20f) < r: fmaxf drops the NaN, so a pair with a NaN coordinate on either side
    IS inside every ball of radius > 1e-20 (tf_grouping_g.cu:24-26; PTX max.f32).  Pinned against the
    oracle, whose C fmaxf has the same semantics."""
    from rfnet_amd import _raw as R
    rng = np.random.RandomState(ns)
    pts = rng.rand(2, 700, 3).astype(np.float32)
    q = pts[:, :90].copy()
    pts[0, 3, 1] = np.nan       # a NaN dataset point: inside everyone's ball
    pts[1, 650] = np.nan
    q[0, 7, 0] = np.nan         # a NaN query: every dataset point is inside
    oi, oc = orc.query_ball_point(np.float32(0.15), ns, pts, q)
    gi, gc = R.query_ball_point(0.15, ns, cu(pts), cu(q))
    assert np.array_equal(gc.cpu().numpy(), oc) and np.array_equal(gi.cpu().numpy(), oi)
    assert oc[0, 7] == ns and list(oi[0, 7]) == list(range(ns))
    assert (oi[0, :, :] == 3).any(-1).all()  # point 3 is in every ball of cloud 0
    # a radius inside the 1e-20 clamp: nothing is inside, not even the NaNs
    gi, gc = R.query_ball_point(1e-21, ns, cu(pts), cu(q))
    assert int(gc.sum()) == 0


def test_group_point_model_shape_and_autograd(orc):
    """merge_layer's use: c=3, nsample=1 (vv_recon.py:135) and the reference's gradient test
    shapes (tf_grouping_op_test.py: points (1,128,16), 8 queries, nsample 32)."""
    from tf_ops.grouping.tf_grouping import group_point, query_ball_point
    rng = np.random.RandomState(6)
    pts = rng.rand(4, 1024, 3).astype(np.float32)
    idx = rng.randint(0, 1024, size=(4, 3000, 1)).astype(np.int32)
    out = group_point(cu(pts), cu(idx))
    assert np.array_equal(out.cpu().numpy(), orc.group_point(pts, idx))
    points = rng.rand(1, 128, 16).astype(np.float32)
    xyz1 = rng.rand(1, 128, 3).astype(np.float32)
    xyz2 = rng.rand(1, 8, 3).astype(np.float32)
    qi, _ = query_ball_point(0.3, 32, cu(xyz1), cu(xyz2))
    tp = cu(points).requires_grad_(True)
    g = group_point(tp, qi)
    w = cu(rng.rand(1, 8, 32, 16).astype(np.float32))
    (g * w).sum().backward()
    exp = orc.group_point_grad(points, qi.cpu().numpy(), w.cpu().numpy())
    assert_rel(tp.grad.cpu().numpy(), exp, 1e-5, 1e-6)


# ------------------------------------------------------------------ interpolation
def test_interpolate_golden(golden):
    from tf_ops.interpolation.tf_interpolate import three_interpolate, three_interpolate_grad, three_nn
    g = golden("interpolate")
    d, i = three_nn(cu(g["xyz1"]), cu(g["xyz2"]))
    assert np.array_equal(d.cpu().numpy(), g["ref_dist"]) and np.array_equal(i.cpu().numpy(), g["ref_idx"])
    out = three_interpolate(cu(g["points"]), i, cu(g["weight"]))
    assert np.array_equal(out.cpu().numpy(), g["ref_out"])
    gp = three_interpolate_grad(cu(g["points"]), i, cu(g["weight"]), cu(g["grad_out"]))
    assert_rel(gp.cpu().numpy(), g["ref_grad_points"], 1e-5, 1e-6)
    d, i = three_nn(cu(g["b_xyz1"]), cu(g["b_xyz2"]))
    assert np.array_equal(d.cpu().numpy(), g["b_ref_dist"]) and np.array_equal(i.cpu().numpy(), g["b_ref_idx"])
    out = three_interpolate(cu(g["b_points"]), i, cu(g["b_weight"]))
    assert np.array_equal(out.cpu().numpy(), g["b_ref_out"])
    d, i = three_nn(cu(g["b_xyz1"]), cu(g["b_xyz2_small"]))  # m = 2 < 3
    assert np.array_equal(d.cpu().numpy(), g["b_small_ref_dist"])
    assert np.array_equal(i.cpu().numpy(), g["b_small_ref_idx"])


@pytest.mark.parametrize("b,n,m,c", [(1, 1, 1, 1), (2, 300, 3, 4), (2, 1000, 2500, 7), (3, 64, 1025, 32)])
def test_interpolate_random(orc, b, n, m, c):
    from tf_ops.interpolation.tf_interpolate import three_interpolate, three_nn
    rng = np.random.RandomState(n + m)
    u = rng.randn(b, n, 3).astype(np.float32)
    k = rng.randn(b, m, 3).astype(np.float32)
    if m > 10:
        k[:, m // 2:m // 2 + 5] = k[:, :5]  # duplicates: earlier index wins
    d, i = three_nn(cu(u), cu(k))
    od, oi = orc.three_nn(u, k)
    assert np.array_equal(d.cpu().numpy(), od) and np.array_equal(i.cpu().numpy(), oi)
    pts = rng.randn(b, m, c).astype(np.float32)
    w = rng.rand(b, n, 3).astype(np.float32)
    tp = cu(pts).requires_grad_(True)
    out = three_interpolate(tp, i, cu(w))
    assert np.array_equal(out.detach().cpu().numpy(), orc.three_interpolate(pts, oi, w))
    go = rng.randn(b, n, c).astype(np.float32)
    out.backward(cu(go))
    assert_rel(tp.grad.cpu().numpy(), orc.three_interpolate_grad(pts, oi, w, go), 1e-5, 1e-5)


@pytest.mark.parametrize("b,n,m,c", [(2, 5000, 300, 64), (2, 3000, 1024, 16), (1, 70000, 128, 8), (3, 2500, 2048, 8), (2, 4100, 256, 40),
                                     (2, 2000, 513, 128), (1, 9000, 4096, 32), (2, 777, 50, 6), (2, 1000, 90, 12)])
def test_interpolate_row_and_tile_forms(orc, b, n, m, c):
    """three_interpolate as rows (a thread row per unknown point, channels as 4-wide vectors where c % 4 == 0 and the tensors are
    16-byte aligned) and its gradient as LDS tiles of doubles (slices of 8..64 channels -- or all of a c like 6 or 12 in one slice --
    with m * cs <= 16384, the unknown points in parts; the element-per-thread kernel outside that: 4096 known points here): out bit-exact, grad_points within fp32 summation noise of the
    reference's sequential sums (tf_interpolate.cpp:107-153) -- also through views that are NOT 16-byte aligned, and with every
    unknown point on the same three known points (one address for all the adds)."""
    from rfnet_amd import _raw as R
    rng = np.random.RandomState(n + m + c)
    u = rng.rand(b, n, 3).astype(np.float32)
    k = rng.rand(b, m, 3).astype(np.float32)
    _, oi = orc.three_nn(u, k)
    pts = rng.randn(b, m, c).astype(np.float32)
    w = rng.rand(b, n, 3).astype(np.float32)
    go = rng.randn(b, n, c).astype(np.float32)
    for idx in (oi, np.broadcast_to(np.array([m - 1, 0, m // 2], np.int32), (b, n, 3)).copy()):
        want_out = orc.three_interpolate(pts, idx, w)
        want_g = orc.three_interpolate_grad(pts, idx, w, go)
        out = R.three_interpolate(cu(pts), cu(idx), cu(w))
        assert np.array_equal(out.cpu().numpy(), want_out)
        g = R.three_interpolate_grad(cu(pts), cu(idx), cu(w), cu(go))
        assert_rel(g.cpu().numpy(), want_g, 1e-5, 1e-5 * max(1.0, float(np.abs(want_g).max())))
        # the same through tensors that start 4 bytes into their allocation
        def off(a):
            flat = torch.empty(a.size + 1, dtype=torch.from_numpy(a).dtype, device="cuda")
            flat[1:] = cu(a).reshape(-1)
            return flat[1:].view(a.shape)
        out = R.three_interpolate(off(pts), cu(idx), cu(w))
        assert np.array_equal(out.cpu().numpy(), want_out)
        g = R.three_interpolate_grad(cu(pts), cu(idx), cu(w), off(go))
        assert_rel(g.cpu().numpy(), want_g, 1e-5, 1e-5 * max(1.0, float(np.abs(want_g).max())))


def test_numpy_and_cpu_tensor_inputs_round_trip(orc):
    from tf_ops.sampling.tf_sampling import farthest_point_sample, gather_point
    p = np.random.RandomState(1).rand(2, 700, 3).astype(np.float32)
    i_np = farthest_point_sample(16, p)
    assert isinstance(i_np, np.ndarray) and i_np.dtype == np.int32
    i_cpu = farthest_point_sample(16, torch.from_numpy(p))
    assert isinstance(i_cpu, torch.Tensor) and not i_cpu.is_cuda
    assert np.array_equal(i_np, i_cpu.numpy()) and np.array_equal(i_np, orc.farthest_point_sample(16, p))
    assert np.array_equal(gather_point(p, i_np), orc.gather_point(p, i_np))


# ------------------------------------------------------------------ the reference's own tests
def _numeric_grad(f, x, eps=1e-2):
    """central differences of sum(f(x) * w) wrt x, like tf.test.compute_gradient_error does per
    element (float64 accumulation on the host)."""
    g = np.zeros_like(x, dtype=np.float64)
    it = np.nditer(x, flags=["multi_index"])
    while not it.finished:
        i = it.multi_index
        old = x[i]
        x[i] = old + eps
        hi = f(x)
        x[i] = old - eps
        lo = f(x)
        x[i] = old
        g[i] = (hi - lo) / (2 * eps)
        it.iternext()
    return g


def test_reference_group_point_gradient_test():
    """tf_ops/grouping/tf_grouping_op_test.py:9-25: points (1,128,16), xyz1 (1,128,3), xyz2 (1,8,3),
    radius 0.3, nsample 32; gradient error of group_point wrt points must be < 1e-4."""
    from tf_ops.grouping.tf_grouping import group_point, query_ball_point
    rng = np.random.RandomState(0)
    points = rng.random_sample((1, 128, 16)).astype(np.float32)
    xyz1 = rng.random_sample((1, 128, 3)).astype(np.float32)
    xyz2 = rng.random_sample((1, 8, 3)).astype(np.float32)
    idx, _ = query_ball_point(0.3, 32, cu(xyz1), cu(xyz2))
    w = rng.random_sample((1, 8, 32, 16))
    tp = cu(points).requires_grad_(True)
    (group_point(tp, idx) * cu(w.astype(np.float32))).sum().backward()
    analytic = tp.grad.cpu().numpy().astype(np.float64)

    def f(p):
        return float((group_point(cu(p), idx).cpu().numpy().astype(np.float64) * w).sum())
    sub = points[:, :6].copy()  # finite differences on a slice of the points (every channel)

    def f_sub(ps):
        p = points.copy()
        p[:, :6] = ps
        return f(p)
    numeric = _numeric_grad(f_sub, sub)
    assert np.abs(numeric - analytic[:, :6]).max() < 1e-4 * max(1.0, np.abs(analytic).max())


def test_reference_three_interpolate_gradient_test():
    """tf_ops/interpolation/tf_interpolate_op_test.py:9-21: points (1,8,16), xyz1 (1,128,3),
    xyz2 (1,8,3), weights 1/3; gradient error of three_interpolate wrt points < 1e-4."""
    from tf_ops.interpolation.tf_interpolate import three_interpolate, three_nn
    rng = np.random.RandomState(0)
    points = rng.random_sample((1, 8, 16)).astype(np.float32)
    xyz1 = rng.random_sample((1, 128, 3)).astype(np.float32)
    xyz2 = rng.random_sample((1, 8, 3)).astype(np.float32)
    _, idx = three_nn(cu(xyz1), cu(xyz2))
    weight = cu(np.full((1, 128, 3), 1.0 / 3.0, np.float32))
    w = rng.random_sample((1, 128, 16))
    tp = cu(points).requires_grad_(True)
    (three_interpolate(tp, idx, weight) * cu(w.astype(np.float32))).sum().backward()
    analytic = tp.grad.cpu().numpy().astype(np.float64)

    def f(p):
        return float((three_interpolate(cu(p), idx, weight).cpu().numpy().astype(np.float64) * w).sum())
    numeric = _numeric_grad(f, points.copy())
    assert np.abs(numeric - analytic).max() < 1e-4 * max(1.0, np.abs(analytic).max())


@pytest.mark.parametrize("b,n,m", [(4, 16384, 300), (3, 8193, 64), (2, 12001, 1200), (5, 16000, 40), (1, 9000, 9000), (3, 1025, 700),
                                   (2, 2048, 2048), (3, 3000, 600), (2, 4097, 513), (2, 7000, 33)])
def test_fps_over_the_sorted_cloud_is_bit_identical(orc, b, n, m):
    """rf_farthestpointsampling_sorted (sampling.hip fps_sorted_kernel: the cloud in sort-tile-recursive order, a lane's 2 to 16
    consecutive points skipped while the new sample is provably too far to lower any of their running minima): the same indices
    as farthest_point_sample and the samples' coordinates, on uniform clouds, lattices (masses of exact ties: the reference's
    tie order through the ORIGINAL indices), duplicated points, a cloud with an outlier and a cloud with a NaN point; one cloud
    also against the oracle."""
    from rfnet_amd import _raw as R
    rng = np.random.RandomState(b * 31 + n + m)
    kinds = {"uniform": rng.random_sample((b, n, 3)), "lattice": rng.randint(0, 7, size=(b, n, 3)) / 6.0,
             "duplicates": np.repeat(rng.random_sample((b, (n + 2) // 3, 3)), 3, axis=1)[:, :n],
             "outlier": np.concatenate([rng.random_sample((b, n - 1, 3)) * 0.1, np.full((b, 1, 3), 50.0)], 1)}
    nanc = rng.random_sample((b, n, 3))
    nanc[:, n // 2, 1] = np.nan
    kinds["nan point"] = nanc
    for kind, a in kinds.items():
        x = cu(a.astype(np.float32))
        want = R.farthest_point_sample(m, x)
        got, nx = R.farthest_point_sample_sorted(m, x, with_xyz=True)
        assert torch.equal(got, want), kind
        assert torch.equal(nx.nan_to_num(nan=-7.0), R.gather_point(x, want).nan_to_num(nan=-7.0)), kind
        assert torch.equal(R.farthest_point_sample_reg(m, x), want), kind  # (whichever form the op itself chose)
    if m <= 300:
        a = kinds["lattice"].astype(np.float32)
        assert np.array_equal(R.farthest_point_sample_sorted(m, cu(a)).cpu().numpy()[:1], orc.farthest_point_sample(m, a[:1]))
