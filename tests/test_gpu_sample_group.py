"""rf_sample_and_group (sample_group.hip): BASELINE.json configs[2]'s chain as one call -- every output bit-identical to
farthest_point_sample -> gather_point -> query_ball_point -> group_point run as separate ops (each of which is pinned
against the oracle elsewhere), with and without the auxiliary stream, at the configuration's own size, on ragged sizes, with
the radius on the device, with ties, and under HIP-graph capture."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def cu(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def _chain(R, npoint, r, ns, x):
    fi = R.farthest_point_sample(npoint, x)
    nx = R.gather_point(x, fi)
    gi, cnt = R.query_ball_point(r, ns, x, nx, form="scan")
    gx = R.group_point(x, gi)
    return fi, nx, gi, cnt, gx


@pytest.mark.parametrize("b,n,npoint,ns,r", [(32, 16384, 1024, 32, 0.1), (3, 3000, 64, 16, 0.2), (2, 65, 65, 8, 0.5),
                                              (2, 20000, 100, 64, 0.05), (5, 8192, 512, 1, 0.03), (2, 4097, 33, 33, 2.0)])
def test_one_call_equals_the_four_ops(b, n, npoint, ns, r):
    from rfnet_amd import _raw as R
    rng = np.random.RandomState(b + n + npoint)
    x = cu(rng.random_sample((b, n, 3)).astype(np.float32))
    want = _chain(R, npoint, r, ns, x)
    got = R.sample_and_group(npoint, r, ns, x)
    aux = torch.cuda.Stream()
    got2 = R.sample_and_group(npoint, r, ns, x, aux_stream=aux)
    torch.cuda.synchronize()
    for name, w, g, g2 in zip(("fps_idx", "new_xyz", "idx", "pts_cnt", "grouped_xyz"), want, got, got2):
        assert torch.equal(w, g), name
        assert torch.equal(w, g2), name + " (aux stream)"


def test_one_call_ties_device_radius_and_tiny_radius(orc):
    from rfnet_amd import _raw as R
    rng = np.random.RandomState(9)
    x = cu((rng.randint(0, 12, size=(4, 5000, 3)) / 11.0).astype(np.float32))  # a lattice: ties everywhere
    want = _chain(R, 200, 0.15, 32, x)
    got = R.sample_and_group(200, torch.tensor([0.15], device="cuda"), 32, x)
    for w, g in zip(want, got):
        assert torch.equal(w, g)
    # against the oracle directly, small
    y = rng.random_sample((2, 700, 3)).astype(np.float32)
    fi, nx, gi, cnt, gx = [t.cpu().numpy() for t in R.sample_and_group(50, 0.2, 16, cu(y))]
    ofi = orc.farthest_point_sample(50, y)
    assert np.array_equal(fi, ofi)
    onx = np.take_along_axis(y, ofi[..., None].astype(np.int64), 1)
    assert np.array_equal(nx, onx)
    oi, oc = orc.query_ball_point(0.2, 16, y, onx)
    assert np.array_equal(gi, oi) and np.array_equal(cnt, oc)
    assert np.array_equal(gx, orc.group_point(y, oi))
    # a radius inside the 1e-20 clamp: every ball empty -> pts_cnt 0, rows defined (index 0), grouped = point 0
    fi, nx, gi, cnt, gx = R.sample_and_group(50, 1e-21, 16, cu(y))
    assert int(cnt.sum()) == 0 and int(gi.abs().sum()) == 0
    assert torch.equal(gx, cu(y)[:, :1, None, :].expand(-1, 50, 16, -1))


def test_one_call_replays_from_a_hip_graph():
    from rfnet_amd import _raw as R
    from rfnet_amd._lib import check, lib
    rng = np.random.RandomState(3)
    b, n, npoint, ns = 8, 16384, 256, 32
    x = cu(rng.random_sample((b, n, 3)).astype(np.float32))
    want = _chain(R, npoint, 0.1, ns, x)
    fi = torch.empty(b, npoint, dtype=torch.int32, device="cuda")
    nx = torch.empty(b, npoint, 3, device="cuda")
    gi = torch.empty(b, npoint, ns, dtype=torch.int32, device="cuda")
    cnt = torch.empty(b, npoint, dtype=torch.int32, device="cuda")
    gx = torch.empty(b, npoint, ns, 3, device="cuda")
    wsz = lib.rf_sample_and_group_workspace_bytes(b, n)
    ws = torch.empty(wsz, dtype=torch.uint8, device="cuda")
    aux = torch.cuda.Stream()

    def call():
        s = torch.cuda.current_stream()
        check(lib.rf_sample_and_group(b, n, npoint, 0.1, None, ns, x.data_ptr(), fi.data_ptr(), nx.data_ptr(), gi.data_ptr(),
                                      cnt.data_ptr(), gx.data_ptr(), ws.data_ptr(), wsz, s.cuda_stream, aux.cuda_stream),
              "rf_sample_and_group")

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        call()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        call()
    for t in (fi, nx, gi, cnt, gx):
        t.zero_()
    g.replay()
    torch.cuda.synchronize()
    for w, got in zip(want, (fi, nx, gi, cnt, gx)):
        assert torch.equal(w, got)
