"""GPU parity of the boxed query_ball_point (grouping.hip query_ball_boxes_kernel, rf_queryballpoint_boxes): idx and
pts_cnt bit-exact against the oracle (oracle/rfops_oracle.c restating tf_ops/grouping/tf_grouping_g.cu:3-36) and against
the scan kernels, over radii from "every ball empty" to "every ball holds the cloud", ragged sizes, duplicated points,
non-finite coordinates on either side, the device-scalar radius, and a caller-provided sort handle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def cu(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def _both(R, r, ns, ds, q, **kw):
    gi, gc = R.query_ball_point(r, ns, cu(ds), cu(q), form="boxes", **kw)
    si, sc = R.query_ball_point(r, ns, cu(ds), cu(q), form="scan")
    return gi.cpu().numpy(), gc.cpu().numpy(), si.cpu().numpy(), sc.cpu().numpy()


@pytest.mark.parametrize("b,n,m,ns,r", [
    (2, 64, 10, 8, 0.3), (2, 65, 33, 64, 0.4), (3, 1000, 257, 16, 0.15), (2, 5000, 100, 64, 0.2),
    (2, 200, 50, 4, 1e-6), (4, 16384, 300, 32, 0.1), (2, 16384, 64, 32, 0.02), (2, 16384, 64, 32, 3.0),
    (1, 65536, 40, 48, 0.05), (2, 4097, 129, 1, 0.07), (3, 2048, 2048, 32, 0.25), (2, 3000, 77, 33, 0.5),
])
def test_boxes_match_oracle_and_scan(orc, b, n, m, ns, r):
    from rfnet_amd import _raw as R
    rng = np.random.RandomState(n + m + ns)
    ds = rng.rand(b, n, 3).astype(np.float32)
    q = rng.rand(b, m, 3).astype(np.float32)
    q[:, : m // 2] = ds[:, : m // 2] if n >= m // 2 else q[:, : m // 2]  # half of the queries ARE dataset points
    gi, gc, si, sc = _both(R, r, ns, ds, q)
    oi, oc = orc.query_ball_point(r, ns, ds, q, fill=0)
    assert np.array_equal(gc, oc) and np.array_equal(gi, oi)
    assert np.array_equal(sc, oc) and np.array_equal(si, oi)


@pytest.mark.parametrize("kind", ["dup", "lattice", "clustered", "flat"])
def test_boxes_degenerate_clouds(orc, kind):
    """Ties in the sort keys, empty boxes, one spot holding most of the cloud, a cloud with no extent on one axis."""
    from rfnet_amd import _raw as R
    rng = np.random.RandomState(len(kind))
    b, n, m, ns = 2, 6000, 200, 32
    if kind == "dup":
        ds = rng.rand(b, n, 3)
        ds[:, n // 3:] = ds[:, : n - n // 3]
    elif kind == "lattice":
        ds = rng.randint(0, 9, size=(b, n, 3)) / 8.0
    elif kind == "clustered":
        ds = rng.rand(b, n, 3)
        ds[:, : n // 2] = 0.5 + 1e-4 * rng.randn(b, n // 2, 3)
    else:
        ds = rng.rand(b, n, 3)
        ds[..., 2] = 0.25
    ds = ds.astype(np.float32)
    q = ds[:, rng.permutation(n)[:m]].copy()
    for r in (0.05, 0.2, 0.0011):
        gi, gc, si, sc = _both(R, r, ns, ds, q)
        oi, oc = orc.query_ball_point(r, ns, ds, q, fill=0)
        assert np.array_equal(gc, oc) and np.array_equal(gi, oi), (kind, r)
        assert np.array_equal(sc, oc) and np.array_equal(si, oi), (kind, r)


@pytest.mark.parametrize("ns", [8, 48])
def test_boxes_non_finite_points_and_queries(orc, ns):
    """A NaN coordinate on either side makes d2 NaN, and the reference's fmaxf drops it: the pair is a hit
    (tf_grouping_g.cu:24-26).  A box cannot bound a NaN point, so such a cloud (and such a query) is walked in index
    order inside the boxed kernel; +-inf coordinates are never hits unless both sides are infinite (inf - inf = NaN)."""
    from rfnet_amd import _raw as R
    rng = np.random.RandomState(ns)
    pts = rng.rand(4, 3000, 3).astype(np.float32)
    q = pts[:, :90].copy()
    pts[0, 3, 1] = np.nan
    pts[1, 2650] = np.nan
    pts[2, 100, 0] = np.inf
    pts[2, 200, 2] = -np.inf
    q[0, 7, 0] = np.nan
    q[2, 5, 0] = np.inf          # inf - inf at dataset point 100: NaN, a hit
    q[3, 11, 2] = np.nan         # a NaN query against a finite cloud
    q[3, 12, 1] = -np.inf        # an infinite query against a finite cloud: no hit at all
    oi, oc = orc.query_ball_point(np.float32(0.15), ns, pts, q)
    gi, gc, si, sc = _both(R, 0.15, ns, pts, q)
    assert np.array_equal(gc, oc) and np.array_equal(gi, oi)
    assert np.array_equal(sc, oc) and np.array_equal(si, oi)
    assert oc[0, 7] == ns and list(oi[0, 7]) == list(range(ns)) and oc[3, 12] == 0
    gi, gc = R.query_ball_point(1e-21, ns, cu(pts), cu(q), form="boxes")
    assert int(gc.sum()) == 0


def test_boxes_radius_on_device_and_sort_handle(orc):
    from rfnet_amd import _raw as R
    rng = np.random.RandomState(77)
    ds = rng.rand(3, 9000, 3).astype(np.float32)
    q = rng.rand(3, 500, 3).astype(np.float32)
    oi, oc = orc.query_ball_point(np.float32(0.12), 32, ds, q, fill=0)
    rdev = torch.tensor([0.12], dtype=torch.float32, device="cuda")
    gi, gc = R.query_ball_point(rdev, 32, cu(ds), cu(q), form="boxes")
    assert np.array_equal(gc.cpu().numpy(), oc) and np.array_equal(gi.cpu().numpy(), oi)
    h = R.nn_sort(cu(ds))
    gi, gc = R.query_ball_point(0.12, 32, cu(ds), cu(q), form="boxes", sorted1=h.buf)
    assert np.array_equal(gc.cpu().numpy(), oc) and np.array_equal(gi.cpu().numpy(), oi)


def test_boxes_radius_boundary_is_in_distance_domain(orc):
    """sqrt_rn(d2) straddling the radius by an ulp (as tests/test_gpu_sampling_grouping.py for the scan kernels)."""
    from rfnet_amd import _raw as R
    rng = np.random.RandomState(4)
    r = np.float32(0.1)
    dirs = rng.randn(1, 4096, 3)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    scale = r * (1 + (rng.rand(1, 4096, 1) - 0.5) * 4e-7)
    ds = (dirs * scale).astype(np.float32)
    q = np.zeros((1, 1, 3), np.float32)
    oi, oc = orc.query_ball_point(float(r), 64, ds, q)
    gi, gc = R.query_ball_point(float(r), 64, cu(ds), cu(q), form="boxes")
    assert np.array_equal(gc.cpu().numpy(), oc) and np.array_equal(gi.cpu().numpy(), oi)


def test_boxes_domain_and_workspace_errors():
    from rfnet_amd import _raw as R
    from rfnet_amd._lib import lib
    assert lib.rf_queryballpoint_boxes_workspace_bytes(2, 63) == 0
    assert lib.rf_queryballpoint_boxes_workspace_bytes(2, 65537) == 0
    x = torch.rand(1, 32, 3, device="cuda")
    with pytest.raises(ValueError):
        R.query_ball_point(0.1, 8, x, x, form="boxes")
    # below the auto threshold the scan kernel is taken and gives the same result as the boxed one above it
    y = torch.rand(1, 4096, 3, device="cuda")
    a = R.query_ball_point(0.1, 16, y, y[:, :64].contiguous(), form="auto")
    s = R.query_ball_point(0.1, 16, y, y[:, :64].contiguous(), form="scan")
    assert torch.equal(a[0], s[0]) and torch.equal(a[1], s[1])
