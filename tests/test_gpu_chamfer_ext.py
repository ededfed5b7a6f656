"""GPU: the composite Chamfer entry points (include/rfops.h "one direction, sorted-cloud handles,
one-call step" + the fused loss / merge_layer ops) against the oracle and against the plain
operator chain they replace.  Bit-exact where the plain ops are (dist / idx), glue tolerance
(rel 1e-5) for the TensorFlow-side arithmetic restated in numpy float64 over oracle outputs."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def cu(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def clouds(seed, b, n, m, kind="randn"):
    rng = np.random.RandomState(seed)
    if kind == "randn":
        return rng.randn(b, n, 3).astype(np.float32), rng.randn(b, m, 3).astype(np.float32)
    if kind == "dup":  # resample_pcd-style duplicates: exact ties
        a = rng.rand(b, max(n // 3, 1), 3).astype(np.float32)
        c = rng.rand(b, max(m // 3, 1), 3).astype(np.float32)
        ia = rng.randint(0, a.shape[1], (b, n))
        ic = rng.randint(0, c.shape[1], (b, m))
        return (np.take_along_axis(a, ia[..., None], 1), np.take_along_axis(c, ic[..., None], 1))
    raise ValueError(kind)


# shapes that take the dense one-direction sweep (small), the culled one (large), in-place and packed
DIR_SHAPES = [(2, 100, 300), (3, 513, 64), (2, 2048, 512), (4, 3000, 1024), (2, 4096, 4096), (1, 1500, 9000)]


@pytest.mark.parametrize("b,n,m", DIR_SHAPES)
@pytest.mark.parametrize("kind", ["randn", "dup"])
def test_one_direction_is_bit_identical(orc, b, n, m, kind):
    from rfnet_amd import _raw as R
    a, c = clouds(b * 1000 + n, b, n, m, kind)
    e = orc.nn_distance(a, c)
    d1, i1, d2, i2 = R.nn_distance_dir(cu(a), cu(c), True, False)
    assert d2 is None and i2 is None
    assert np.array_equal(d1.cpu().numpy(), e[0]) and np.array_equal(i1.cpu().numpy(), e[1])
    d1, i1, d2, i2 = R.nn_distance_dir(cu(a), cu(c), False, True)
    assert d1 is None and i1 is None
    assert np.array_equal(d2.cpu().numpy(), e[2]) and np.array_equal(i2.cpu().numpy(), e[3])
    both = R.nn_distance_dir(cu(a), cu(c), True, True)
    for got, exp in zip(both, e):
        assert np.array_equal(got.cpu().numpy(), exp)


@pytest.mark.parametrize("b,n,m", [(2, 64, 1024), (3, 3000, 1024), (2, 16384, 2048), (1, 20000, 5000)])
def test_sorted_handles_reused_across_calls(orc, b, n, m):
    """One rf_nn_sort per cloud, several sweeps: same bits as nn_distance every time, either
    direction alone included, and the handle of one cloud pairs with different partners."""
    from rfnet_amd import _raw as R
    a, c = clouds(7 + n, b, n, m)
    c2 = clouds(8 + n, b, n, m)[1]
    ha, hc, hc2 = R.nn_sort(cu(a)), R.nn_sort(cu(c)), R.nn_sort(cu(c2))
    for partner, h in ((c, hc), (c2, hc2), (c, hc)):
        e = orc.nn_distance(a, partner)
        got = R.nn_distance_sorted(ha, h)
        for g, x in zip(got, e):
            assert np.array_equal(g.cpu().numpy(), x)
        d1, i1, d2, i2 = R.nn_distance_sorted(ha, h, True, False)
        assert d2 is None and np.array_equal(i1.cpu().numpy(), e[1]) and np.array_equal(d1.cpu().numpy(), e[0])
        d1, i1, d2, i2 = R.nn_distance_sorted(ha, h, False, True)
        assert d1 is None and np.array_equal(i2.cpu().numpy(), e[3]) and np.array_equal(d2.cpu().numpy(), e[2])
    with pytest.raises(ValueError):
        R.nn_distance_sorted(ha, R.nn_sort(cu(np.concatenate([c, c[:1]], 0))))  # batch mismatch


def test_chamfer_step_one_call_equals_two_ops(orc):
    from rfnet_amd import _raw as R
    for (b, n, m) in ((2, 300, 700), (4, 2048, 4096)):
        a, c = clouds(11, b, n, m)
        rng = np.random.RandomState(3)
        gd1, gd2 = rng.rand(b, n).astype(np.float32), rng.rand(b, m).astype(np.float32)
        plan = R.ChamferStep(b, n, m, "cuda")
        ta, tc, tg1, tg2 = cu(a), cu(c), cu(gd1), cu(gd2)
        for _ in range(2):  # the plan's buffers are reused
            d1, i1, d2, i2, g1, g2 = plan(ta, tc, tg1, tg2)
        e = orc.nn_distance(a, c)
        for got, exp in zip((d1, i1, d2, i2), e):
            assert np.array_equal(got.cpu().numpy(), exp)
        r1, r2 = R.nn_distance_grad(ta, tc, tg1, i1, tg2, i2)
        # (LDS float atomics: the summation order of a scatter is not fixed run to run)
        assert torch.allclose(g1, r1, rtol=1e-5, atol=1e-5 * float(r1.abs().max()))
        assert torch.allclose(g2, r2, rtol=1e-5, atol=1e-5 * float(r2.abs().max()))
        o1, o2 = orc.nn_distance_grad(a, c, gd1, e[1], gd2, e[3])
        assert np.allclose(g1.cpu().numpy(), o1, rtol=1e-5, atol=1e-5 * np.abs(o1).max())
        assert np.allclose(g2.cpu().numpy(), o2, rtol=1e-5, atol=1e-5 * np.abs(o2).max())


def _loss_np(e, want1=True, want2=True):
    l1 = np.sqrt(e[0].astype(np.float64)).mean(1) if want1 else np.zeros(e[0].shape[0])
    l2 = np.sqrt(e[2].astype(np.float64)).mean(1) if want2 else np.zeros(e[0].shape[0])
    return np.stack([l1, l2], 1)


@pytest.mark.parametrize("b,n,m", [(3, 200, 500), (2, 3000, 1024), (2, 4096, 4096)])
def test_chamfer_loss_forward_backward(orc, b, n, m):
    """loss (b,2) = per-sample mean sqrt(dist); backward = NnDistanceGrad with
    gd = gl/N * 0.5/sqrt(d) -- both against numpy float64 over the oracle's outputs."""
    from rfnet_amd import _raw as R
    a, c = clouds(21 + n, b, n, m)
    e = orc.nn_distance(a, c)
    rng = np.random.RandomState(5)
    gl = rng.rand(b, 2).astype(np.float32) + 0.5
    for (w1, w2) in ((True, True), (True, False), (False, True)):
        loss, d1, i1, d2, i2 = R.chamfer_loss(cu(a), cu(c), None, None, w1, w2)
        assert np.allclose(loss.cpu().numpy(), _loss_np(e, w1, w2), rtol=2e-6, atol=1e-9)
        if w1:
            assert np.array_equal(d1.cpu().numpy(), e[0]) and np.array_equal(i1.cpu().numpy(), e[1])
        if w2:
            assert np.array_equal(d2.cpu().numpy(), e[2]) and np.array_equal(i2.cpu().numpy(), e[3])
        g1, g2 = R.chamfer_loss_grad(cu(a), cu(c), d1, i1, d2, i2, cu(gl))
        gd1 = (gl[:, :1] / n * 0.5 / np.sqrt(e[0].astype(np.float64))).astype(np.float32) * (1 if w1 else 0)
        gd2 = (gl[:, 1:] / m * 0.5 / np.sqrt(e[2].astype(np.float64))).astype(np.float32) * (1 if w2 else 0)
        o1, o2 = orc.nn_distance_grad(a, c, gd1, e[1], gd2, e[3])
        assert np.allclose(g1.cpu().numpy(), o1, rtol=1e-4, atol=1e-5 * np.abs(o1).max())
        assert np.allclose(g2.cpu().numpy(), o2, rtol=1e-4, atol=1e-5 * np.abs(o2).max())
    # with sorted handles (one, the other, both): same outputs
    ha, hc = R.nn_sort(cu(a)), R.nn_sort(cu(c))
    ref = R.chamfer_loss(cu(a), cu(c))
    for (s1, s2) in ((ha, None), (None, hc), (ha, hc)):
        got = R.chamfer_loss(cu(a), cu(c), s1, s2)
        for g, x in zip(got, ref):
            assert torch.equal(g, x)


def test_fused_glue_equals_the_op_chain(orc):
    """glue.chamfer_big / fidelity_loss / merge_layer on the fused ops vs the same formulas written on
    the plain ops with torch autograd (what round 1 shipped): values and gradients."""
    from rfnet_amd import glue
    from rfnet_amd.tf_ops.CD.tf_nndistance import nn_distance
    a, c = clouds(31, 2, 1500, 2600)
    ta, tc = cu(a).requires_grad_(True), cu(c).requires_grad_(True)
    loss, idx1 = glue.chamfer_big(ta, tc)
    (loss * 3.0).backward()
    ra, rc = cu(a).requires_grad_(True), cu(c).requires_grad_(True)
    d1, i1, d2, _ = nn_distance(ra, rc)
    ref = (torch.sqrt(d1).mean() + torch.sqrt(d2).mean()) / 2
    (ref * 3.0).backward()
    assert abs(float(loss.detach()) - float(ref.detach())) < 2e-6 * float(ref.detach()) and torch.equal(idx1, i1)
    for got, exp in ((ta.grad, ra.grad), (tc.grad, rc.grad)):
        assert torch.allclose(got, exp, rtol=1e-4, atol=1e-5 * float(exp.abs().max()))
    # fidelity: direction 1 only
    ta2, tc2 = cu(a).requires_grad_(True), cu(c).requires_grad_(True)
    fl = glue.fidelity_loss(ta2, tc2)
    fl.backward()
    ra2, rc2 = cu(a).requires_grad_(True), cu(c).requires_grad_(True)
    torch.sqrt(nn_distance(ra2, rc2)[0]).mean().backward()
    assert torch.allclose(ta2.grad, ra2.grad, rtol=1e-4, atol=1e-5 * float(ra2.grad.abs().max()))
    assert torch.allclose(tc2.grad, rc2.grad, rtol=1e-4, atol=1e-5 * float(rc2.grad.abs().max()))


@pytest.mark.parametrize("n,m", [(3000, 64), (3000, 1024), (3000, 16384)])
def test_merge_layer_fused_vs_reference_formula(orc, n, m):
    from rfnet_amd import glue
    rng = np.random.RandomState(n + m)
    raw = (rng.rand(2, n, 3) - 0.5).astype(np.float32)
    new = (rng.rand(2, m, 3) - 0.5).astype(np.float32)
    dec = np.float32(0.07)
    # numpy float64 of the reference's formula (vv_recon.py:132-139) over the oracle's idx2
    i2 = orc.nn_distance(raw, new)[3]
    g = np.take_along_axis(raw, i2[..., None].astype(np.int64), 1).astype(np.float64)
    diff = g - new
    ratio = np.exp(-(diff * diff).sum(-1, keepdims=True) / (1e-8 + float(dec) ** 2))
    exp = new + ratio * diff
    hraw = glue.SortedCloud(cu(raw))
    for handle in (None, hraw):
        traw, tnew = cu(raw).requires_grad_(True), cu(new).requires_grad_(True)
        tdec = torch.tensor([dec], device="cuda", requires_grad=True)
        out = glue.merge_layer(traw, tnew, tdec, sorted_raw=handle)
        assert np.allclose(out.detach().cpu().numpy(), exp, rtol=1e-5, atol=1e-6)
        w = cu(rng.randn(2, m, 3).astype(np.float32))
        (out * w).sum().backward()
        uraw, unew = cu(raw).requires_grad_(True), cu(new).requires_grad_(True)
        udec = torch.tensor([dec], device="cuda", requires_grad=True)
        (glue.merge_layer_unfused(uraw, unew, udec) * w).sum().backward()
        assert torch.allclose(tnew.grad, unew.grad, rtol=1e-4, atol=1e-5 * float(unew.grad.abs().max()))
        assert torch.allclose(traw.grad, uraw.grad, rtol=1e-4, atol=1e-5 * float(uraw.grad.abs().max()))
        assert torch.allclose(tdec.grad, udec.grad, rtol=1e-3, atol=1e-4 * float(udec.grad.abs().max()) + 1e-6)


def test_queryball_device_radius_and_abi(orc):
    """rf_queryballpoint_dev (the reference's signature: radius is a device pointer, tf_grouping.cpp:67,
    93-95) gives the same rows as the by-value entry and the oracle, including radii whose square is
    a 1-ulp boundary case."""
    from rfnet_amd import _raw as R
    from rfnet_amd._lib import lib
    rng = np.random.RandomState(0)
    pts = rng.rand(2, 2000, 3).astype(np.float32)
    q = pts[:, :300].copy()
    for r in (0.1, 0.05, float(np.float32(0.3)), float(np.nextafter(np.float32(0.1), np.float32(1))), 1e-21, 1e-19, 5.0):
        oi, oc = orc.query_ball_point(np.float32(r), 16, pts, q)
        gi, gc = R.query_ball_point(r, 16, cu(pts), cu(q))
        rt = torch.tensor([r], dtype=torch.float32, device="cuda")
        di, dc = R.query_ball_point(rt, 16, cu(pts), cu(q))
        assert np.array_equal(gc.cpu().numpy(), oc) and np.array_equal(dc.cpu().numpy(), oc)
        has = oc > 0
        assert np.array_equal(gi.cpu().numpy()[has], oi[has]) and np.array_equal(di.cpu().numpy()[has], oi[has])
    # raw ABI: NULL radius pointer is an argument error
    a, c = cu(pts), cu(q)
    idx = torch.zeros(2, 300, 16, dtype=torch.int32, device="cuda")
    cnt = torch.zeros(2, 300, dtype=torch.int32, device="cuda")
    assert lib.rf_queryballpoint_dev(2, 2000, 300, None, 16, p(a), p(c), p(idx), p(cnt), None) == -1
    assert lib.rf_device_check() == 0


def test_raw_abi_direction_and_sorted_contracts():
    from rfnet_amd._lib import lib
    b, n, m = 2, 1500, 1100
    a, c = [cu(x) for x in clouds(1, b, n, m)]
    d1 = torch.empty(b, n, device="cuda"); i1 = torch.empty(b, n, dtype=torch.int32, device="cuda")
    need = lib.rf_nn_distance_dir_workspace_bytes(b, n, m, 1, 0)
    ws = torch.empty(max(need, 1), dtype=torch.uint8, device="cuda")
    # no direction / missing outputs of a wanted direction / short workspace
    assert lib.rf_nn_distance_dir(b, n, m, p(a), p(c), p(d1), p(i1), None, None, p(ws), need, None, 0, 0) == -1
    assert lib.rf_nn_distance_dir(b, n, m, p(a), p(c), None, p(i1), None, None, p(ws), need, None, 1, 0) == -1
    assert lib.rf_nn_distance_dir(b, n, m, p(a), p(c), p(d1), p(i1), None, None, p(ws), need - 1, None, 1, 0) == -2
    assert lib.rf_nn_distance_dir(b, n, m, p(a), p(c), p(d1), p(i1), None, None, p(ws), need, None, 1, 0) == 0
    # sorted handles: size query is pure, short buffer is refused, > 65536 points unsupported
    sb = lib.rf_nn_sort_bytes(b, n)
    assert sb > 0 and lib.rf_nn_sort_bytes(b, 70000) == 0
    h = torch.empty(sb, dtype=torch.uint8, device="cuda")
    assert lib.rf_nn_sort(b, n, p(a), p(h), sb - 1, None) == -2
    assert lib.rf_nn_sort(b, n, p(a), p(h), sb, None) == 0
    assert lib.rf_nn_distance_sorted(b, n, m, p(h), None, p(d1), p(i1), None, None, None) == -1
    torch.cuda.synchronize()
