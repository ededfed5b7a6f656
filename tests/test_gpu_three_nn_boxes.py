"""GPU parity of the boxed three_nn (interpolate.hip three_nn_boxes_kernel, rf_threenn_boxes): dist and idx bit-exact against
the oracle (oracle/rfops_oracle.c restating threenn_cpu, tf_ops/interpolation/tf_interpolate.cpp:60-103) and against the scan
kernel -- ragged sizes, fewer than three known points, exact ties (duplicates, lattices: the earlier index wins), non-finite
coordinates on either side, clouds without extent, sort handles from the caller, and the rule that picks the form."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def cu(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def _both(R, u, k, **kw):
    bd, bi = R.three_nn(cu(u), cu(k), form="boxes", **kw)
    sd, si = R.three_nn(cu(u), cu(k), form="scan")
    return bd.cpu().numpy(), bi.cpu().numpy(), sd.cpu().numpy(), si.cpu().numpy()


def _same(a, b):
    """bit-equal, +inf and NaN included"""
    return np.array_equal(np.asarray(a).view(np.int32), np.asarray(b).view(np.int32))


@pytest.mark.parametrize("b,n,m", [
    (1, 1, 1), (2, 1, 70), (2, 70, 1), (2, 64, 2), (3, 300, 3), (2, 65, 64), (2, 1000, 2500), (3, 4097, 129),
    (2, 16384, 1024), (1, 20000, 300), (2, 3000, 9000), (1, 65536, 100), (1, 100, 65536),
])
def test_boxes_match_oracle_and_scan(orc, b, n, m):
    from rfnet_amd import _raw as R
    rng = np.random.RandomState(n + m)
    u = rng.rand(b, n, 3).astype(np.float32)
    k = rng.rand(b, m, 3).astype(np.float32)
    if m > 10:
        k[:, m // 2:m // 2 + 5] = k[:, :5]  # duplicates: the earlier index wins
    if n > 4 and m > 4:
        u[:, :4] = k[:, :4]  # distance 0
    bd, bi, sd, si = _both(R, u, k)
    od, oi = orc.three_nn(u, k)
    assert np.array_equal(bi, oi) and _same(bd, od)
    assert np.array_equal(si, oi) and _same(sd, od)


@pytest.mark.parametrize("kind", ["lattice", "dup", "clustered", "flat", "one_spot", "randn_vs_uniform"])
def test_boxes_degenerate_clouds(orc, kind):
    """Exact ties by the thousand, ties in the sort keys, empty boxes, a cloud with no extent on one axis or on all three,
    sets that barely overlap."""
    from rfnet_amd import _raw as R
    rng = np.random.RandomState(len(kind))
    b, n, m = 2, 5000, 1500
    u, k = rng.rand(b, n, 3), rng.rand(b, m, 3)
    if kind == "lattice":
        u, k = rng.randint(0, 7, size=(b, n, 3)) / 4.0, rng.randint(0, 7, size=(b, m, 3)) / 4.0
    elif kind == "dup":
        k[:, m // 3:] = k[:, : m - m // 3]
        u[:, : m // 2] = k[:, : m // 2]
    elif kind == "clustered":
        k[:, : m // 2] = 0.5 + 1e-4 * rng.randn(b, m // 2, 3)
        u[:, : n // 2] = 0.5 + 1e-3 * rng.randn(b, n // 2, 3)
    elif kind == "flat":
        k[..., 2] = 0.25
        u[..., 0] = -3.0
    elif kind == "one_spot":
        k[:] = 0.125
    else:
        u = rng.randn(b, n, 3) * 3.0
    u, k = u.astype(np.float32), k.astype(np.float32)
    bd, bi, sd, si = _both(R, u, k)
    od, oi = orc.three_nn(u, k)
    assert np.array_equal(bi, oi) and _same(bd, od), kind
    assert np.array_equal(si, oi) and _same(sd, od), kind


def test_boxes_non_finite_coordinates(orc):
    """A NaN or infinite coordinate makes every distance of that point NaN or +inf: never inserted (tf_interpolate.cpp:78-93,
    all three comparisons fail) -- an unknown point like that gets (+inf, 0) three times, a known point like that is nobody's
    neighbour; the sort's boxes leave such points out and the other points' results are what they are without them."""
    from rfnet_amd import _raw as R
    rng = np.random.RandomState(5)
    b, n, m = 2, 3000, 900
    u = rng.rand(b, n, 3).astype(np.float32)
    k = rng.rand(b, m, 3).astype(np.float32)
    u[0, 5, 1] = np.nan
    u[0, 77, 0] = np.inf
    u[1, 2999, 2] = -np.inf
    k[0, 3, 0] = np.nan
    k[0, 100, 2] = np.inf
    k[1, 0] = np.nan
    k[1, 450, 1] = -np.inf
    bd, bi, sd, si = _both(R, u, k)
    od, oi = orc.three_nn(u, k)
    assert np.array_equal(bi, oi) and _same(bd, od)
    assert np.array_equal(si, oi) and _same(sd, od)
    assert np.all(np.isinf(bd[0, 5])) and np.all(bi[0, 5] == 0) and np.all(np.isinf(bd[0, 77])) and np.all(bi[1, 2999] == 0)
    assert not np.isin(bi[0], [3, 100]).any() and not np.isin(bi[1, :2999], [0, 450]).any()
    # huge coordinates: squared distances overflow to +inf and are never inserted either
    u2, k2 = u.copy(), k.copy()
    u2[np.isnan(u2) | np.isinf(u2)] = 0.5
    k2[np.isnan(k2) | np.isinf(k2)] = 0.5
    k2[0, :890] *= 3e19
    bd, bi, sd, si = _both(R, u2, k2)
    od, oi = orc.three_nn(u2, k2)
    assert np.array_equal(bi, oi) and _same(bd, od)
    assert np.array_equal(si, oi) and _same(sd, od)


@pytest.mark.parametrize("which", ["known all NaN", "known all +inf", "unknown all NaN", "known: one finite point"])
def test_boxes_sets_without_a_usable_point(orc, which):
    """Every box of the set empty (no finite point to bound): nothing is ever inserted -- (+inf, 0) three times, as the scan and
    the reference leave it; with ONE finite known point that one is everybody's first neighbour and the other two slots stay unfilled."""
    from rfnet_amd import _raw as R
    rng = np.random.RandomState(17)
    b, n, m = 2, 1500, 700
    u = rng.rand(b, n, 3).astype(np.float32)
    k = rng.rand(b, m, 3).astype(np.float32)
    if which == "known all NaN":
        k[:] = np.nan
    elif which == "known all +inf":
        k[:] = np.inf
    elif which == "unknown all NaN":
        u[:] = np.nan
    else:
        keep = k[:, 333].copy()
        k[:] = np.nan
        k[:, 333] = keep
    bd, bi, sd, si = _both(R, u, k)
    od, oi = orc.three_nn(u, k)
    assert np.array_equal(bi, oi) and _same(bd, od), which
    assert np.array_equal(si, oi) and _same(sd, od), which
    if which == "known: one finite point":
        assert np.all(bi[..., 0] == 333) and np.all(np.isinf(bd[..., 1:])) and np.all(bi[..., 1:] == 0)
    else:
        assert np.all(np.isinf(bd)) and np.all(bi == 0)


def test_boxes_on_caller_handles_and_the_auto_rule(orc):
    from rfnet_amd import _raw as R
    rng = np.random.RandomState(9)
    b, n, m = 3, 6000, 2000
    u = rng.randn(b, n, 3).astype(np.float32)
    k = rng.randn(b, m, 3).astype(np.float32)
    tu, tk = cu(u), cu(k)
    od, oi = orc.three_nn(u, k)
    h1, h2 = R.nn_sort(tu), R.nn_sort(tk)
    for kw in ({"sorted1": h1.buf}, {"sorted2": h2.buf}, {"sorted1": h1.buf, "sorted2": h2.buf}):
        d, i = R.three_nn(tu, tk, form="boxes", **kw)
        assert np.array_equal(i.cpu().numpy(), oi) and _same(d.cpu().numpy(), od), sorted(kw)
    # the public op: "auto" takes the boxed kernel from TN_BOXES_MIN_PAIRS pairs on -- same results either way
    from tf_ops.interpolation.tf_interpolate import three_nn
    big_u = cu(rng.rand(8, 16384, 3).astype(np.float32))
    big_k = cu(rng.rand(8, 2048, 3).astype(np.float32))
    assert 8 * 16384 * 2048 >= R.TN_BOXES_MIN_PAIRS
    d, i = three_nn(big_u, big_k)
    sd, si = R.three_nn(big_u, big_k, form="scan")
    assert torch.equal(d, sd) and torch.equal(i, si)
    d, i = three_nn(u, k)  # numpy in -> numpy out, below the threshold
    assert isinstance(d, np.ndarray) and np.array_equal(i, oi) and _same(d, od)


def test_boxes_c_abi_contract():
    """rf_threenn_boxes: workspace size 0 outside the domain, RF_EWORKSPACE on a short workspace, RF_EINVAL on a misaligned
    one or on sizes it does not take, nothing written in those cases; b = 0 / n = 0 are no-ops."""
    from rfnet_amd import _lib
    lib = _lib.lib
    RF_OK, RF_EINVAL, RF_EWORKSPACE = 0, -1, -2  # include/rfops.h
    assert lib.rf_threenn_boxes_workspace_bytes(2, 100, 0) == 0
    assert lib.rf_threenn_boxes_workspace_bytes(2, 65537, 10) == 0
    assert lib.rf_threenn_boxes_workspace_bytes(0, 100, 10) == 0
    b, n, m = 2, 500, 300
    need = lib.rf_threenn_boxes_workspace_bytes(b, n, m)
    assert need > 0
    u, k = torch.rand(b, n, 3, device="cuda"), torch.rand(b, m, 3, device="cuda")
    dist = torch.full((b, n, 3), -7.0, device="cuda")
    idx = torch.full((b, n, 3), -7, dtype=torch.int32, device="cuda")
    ws = torch.empty(need + 64, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    call = lambda ws_ptr, nbytes, bb=b, nn=n, mm=m: lib.rf_threenn_boxes(bb, nn, mm, p(u), p(k), None, None, p(dist), p(idx), ws_ptr,
                                                                       nbytes, ctypes.c_void_p(s))
    assert call(p(ws), need - 1) == RF_EWORKSPACE
    assert call(ctypes.c_void_p(ws.data_ptr() + 4), need) == RF_EINVAL
    assert call(p(ws), need, mm=0) == RF_EINVAL
    assert call(None, need) == RF_EINVAL
    torch.cuda.synchronize()
    assert bool((dist == -7.0).all()) and bool((idx == -7).all())
    assert call(p(ws), need, bb=0) == RF_OK and call(p(ws), need, nn=0) == RF_OK
    assert call(p(ws), need) == RF_OK
    torch.cuda.synchronize()
    from rfnet_amd import _raw as R
    sd, si = R.three_nn(u, k, form="scan")
    assert torch.equal(dist, sd) and torch.equal(idx, si)
