"""GPU parity: the scatter-add gradients of group_point / three_interpolate on their sorted-slots route (scatter_rows.hip: the slots
counting-sorted by destination row, every row written once from sums in double) against the CPU oracle and against the
reference-shaped atomic route of the same library, at shapes past the route's threshold (2^22 gradient elements or 2^19 slots).
Tolerance: rel 1e-5 with an abs floor of 1e-6 of the row scale (atomically accumulated grads)."""
import numpy as np
import pytest
import torch

from conftest import assert_rel

pytestmark = pytest.mark.gpu


def cu(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


@pytest.mark.parametrize("b,n,m,ns,c", [(4, 16384, 1024, 32, 64), (5, 5000, 900, 17, 61), (8, 40000, 3000, 8, 48), (2, 70001, 1200, 64, 32), (2, 163000, 1100, 32, 64),
                                        (12, 1000, 1000, 40, 12), (4, 16384, 16384, 1, 67), (32, 16384, 16384, 1, 3)])
def test_group_point_grad_sorted_slots(orc, b, n, m, ns, c):
    from rfnet_amd import _raw as R
    from rfnet_amd._lib import lib
    rng = np.random.RandomState(n + m + c)
    pts = np.zeros((b, n, c), np.float32)
    # indices as a ball query leaves them: ascending runs, the first hit repeated to the end of a short row; some rows all one index
    idx = np.sort(rng.randint(0, n, size=(b, m, ns)), -1).astype(np.int32)
    short = rng.rand(b, m) < 0.3
    cut = rng.randint(1, ns + 1, size=(b, m))
    for q in range(ns):
        sel = short & (q >= cut)
        idx[..., q] = np.where(sel, idx[..., 0], idx[..., q])
    idx[:, : min(m, 5)] = 7 % n  # a popular point: hundreds of slots on one row
    go = rng.randn(b, m, ns, c).astype(np.float32)
    assert lib.rf_grouppoint_grad_workspace_bytes(b, n, c, m, ns) > 0, "shape is below the sorted-slots threshold: not the route under test"
    got = R.group_point_grad(cu(pts), cu(idx), cu(go)).cpu().numpy()
    want = orc.group_point_grad(pts, idx, go)
    scale = max(1.0, float(np.abs(want).max()))
    assert_rel(got, want, 1e-5, 1e-6 * scale, what="sorted slots vs oracle")
    atomic = R.group_point_grad(cu(pts), cu(idx), cu(go), form="atomic").cpu().numpy()
    assert_rel(got, atomic, 1e-5, 1e-6 * scale, what="sorted slots vs atomics")
    # rows nobody names are exact zeros (they are written, not left from a fill)
    named = np.zeros((b, n), bool)
    for i in range(b):
        named[i, idx[i].ravel()] = True
    assert not got[~named].any()
    # a second call on the same scratch gives the same bits (sums in double: order-free to fp32 rounding)
    again = R.group_point_grad(cu(pts), cu(idx), cu(go)).cpu().numpy()
    assert np.array_equal(got, again)


def test_group_point_grad_sorted_slots_ignores_out_of_range_indices(orc):
    """The reference adds at whatever address an index names; here a slot whose index is outside [0, n) adds to no row."""
    from rfnet_amd import _raw as R
    from rfnet_amd._lib import lib
    rng = np.random.RandomState(3)
    b, n, m, ns, c = 4, 9000, 900, 32, 64
    assert lib.rf_grouppoint_grad_workspace_bytes(b, n, c, m, ns) > 0  # (the atomic route would follow such an index out of its tensor)
    idx = rng.randint(0, n, size=(b, m, ns)).astype(np.int32)
    go = rng.randn(b, m, ns, c).astype(np.float32)
    bad = rng.rand(b, m, ns) < 0.01
    idx_bad = np.where(bad, np.where(rng.rand(b, m, ns) < 0.5, -3, n + 11), idx).astype(np.int32)
    got = R.group_point_grad(cu(np.zeros((b, n, c), np.float32)), cu(idx_bad), cu(go)).cpu().numpy()
    want = orc.group_point_grad(np.zeros((b, n, c), np.float32), idx, np.where(bad[..., None], 0, go).astype(np.float32))
    assert_rel(got, want, 1e-5, 1e-5)


@pytest.mark.parametrize("b,n,m,c", [(2, 16384, 4096, 64), (4, 9001, 2500, 61), (2, 65536, 16384, 16), (8, 16384, 16384, 12), (2, 30000, 40000, 32)])
def test_three_interpolate_grad_sorted_slots(orc, b, n, m, c):
    from rfnet_amd import _raw as R
    from rfnet_amd._lib import lib
    rng = np.random.RandomState(n + m + c)
    pts = np.zeros((b, m, c), np.float32)
    idx = rng.randint(0, m, size=(b, n, 3)).astype(np.int32)
    idx[:, ::7, 1] = idx[:, ::7, 0]  # the same known point twice in one triple (m < 3, coincident points)
    w = rng.rand(b, n, 3).astype(np.float32)
    go = rng.randn(b, n, c).astype(np.float32)
    assert lib.rf_threeinterpolate_grad_workspace_bytes(b, n, c, m) > 0, "not the route under test"
    got = R.three_interpolate_grad(cu(pts), cu(idx), cu(w), cu(go)).cpu().numpy()
    want = orc.three_interpolate_grad(pts, idx, w, go)
    scale = max(1.0, float(np.abs(want).max()))
    assert_rel(got, want, 1e-5, 1e-6 * scale, what="sorted slots vs oracle")
    inline = R.three_interpolate_grad(cu(pts), cu(idx), cu(w), cu(go), form="inline").cpu().numpy()
    assert_rel(got, inline, 1e-5, 1e-6 * scale, what="sorted slots vs atomics")


def test_sorted_slots_abi_contract():
    """rf_grouppoint_grad_ws / rf_threeinterpolate_grad_ws: 0 workspace bytes below the threshold (NULL accepted there); with NULL or
    too little scratch above it the call still completes, on the atomic route; the autograd functions use the scratch form."""
    from rfnet_amd._lib import lib
    from tf_ops.grouping.tf_grouping import group_point
    assert lib.rf_grouppoint_grad_workspace_bytes(2, 100, 3, 10, 4) == 0
    assert lib.rf_threeinterpolate_grad_workspace_bytes(2, 100, 8, 50) == 0      # fits the LDS tile
    assert lib.rf_threeinterpolate_grad_workspace_bytes(8, 16384, 64, 1024) == 0  # tile again (m * 8 <= 16384)
    b, n, m, ns, c = 2, 8192, 1024, 32, 64
    need = lib.rf_grouppoint_grad_workspace_bytes(b, n, c, m, ns)
    assert need > 0
    rng = np.random.RandomState(0)
    idx = torch.from_numpy(rng.randint(0, n, size=(b, m, ns)).astype(np.int32)).cuda()
    go = torch.randn(b, m, ns, c, device="cuda")
    ref = torch.zeros(b, n, c, device="cuda")
    ref.view(b * n, c).index_add_(0, (idx.long() + torch.arange(b, device="cuda").view(b, 1, 1) * n).view(-1), go.view(-1, c))
    for ws_bytes in (0, need // 2):
        out = torch.full((b, n, c), float("nan"), device="cuda")
        ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device="cuda")
        st = lib.rf_grouppoint_grad_ws(b, n, c, m, ns, go.data_ptr(), idx.data_ptr(), out.data_ptr(), ws.data_ptr() if ws_bytes else None,
                                       ws_bytes, None)
        torch.cuda.synchronize()
        assert st == 0 and torch.allclose(out, ref, rtol=1e-5, atol=1e-5)
    pts = torch.randn(b, n, c, device="cuda", requires_grad=True)
    (group_point(pts, idx) * go).sum().backward()
    assert torch.allclose(pts.grad, ref, rtol=1e-5, atol=1e-5)
