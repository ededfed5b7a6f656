"""GPU: rf_chamfer_step on the culled path -- the sweep leaves {winner's sorted position, upstream gradient}
records and the backward runs in SORTED index space (nnp_grad_sorted_kernel, nn_pruned.hip).  Forward outputs
bit-exact against the oracle; gradients against the oracle's NnDistanceGrad restatement
(tf_ops/CD/tf_nndistance.cpp:126-163 via oracle/rfops_oracle.c) at the gradient tolerance (rel 1e-5 + 1e-5 of
the largest term: fp32 add order of a scatter) and against the original-order kernel (rf_nn_distance_grad)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def cu(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def make(kind, seed, b, n, m):
    rng = np.random.RandomState(seed)
    if kind == "randn":
        return rng.randn(b, n, 3).astype(np.float32), rng.randn(b, m, 3).astype(np.float32)
    if kind == "uniform":
        return (rng.rand(b, n, 3) - 0.5).astype(np.float32), (rng.rand(b, m, 3) - 0.5).astype(np.float32)
    if kind == "dup":  # resample_pcd-style duplicates on both sides: exact ties, several queries per winner
        a = rng.rand(b, max(n // 3, 1), 3).astype(np.float32)
        c = rng.rand(b, max(m // 5, 1), 3).astype(np.float32)
        return (np.take_along_axis(a, rng.randint(0, a.shape[1], (b, n))[..., None], 1),
                np.take_along_axis(c, rng.randint(0, c.shape[1], (b, m))[..., None], 1))
    if kind == "lattice":  # symmetric configurations: ties between DIFFERENT points across blocks
        ga = np.stack(np.meshgrid(*[np.arange(16)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
        a = np.stack([ga[rng.randint(0, len(ga), n)] for _ in range(b)])
        c = np.stack([ga[rng.randint(0, len(ga), m)] + 0.5 for _ in range(b)]).astype(np.float32)
        return a, c
    if kind == "clustered":
        ca = rng.randn(b, 8, 3)
        a = ca[np.arange(b)[:, None], rng.randint(0, 8, (b, n))] + 0.01 * rng.randn(b, n, 3)
        c = ca[np.arange(b)[:, None], rng.randint(0, 8, (b, m))] + 0.05 * rng.randn(b, m, 3)
        return a.astype(np.float32), c.astype(np.float32)
    if kind == "collapsed":  # the untrained network's output: the large cloud sits on ~120 spots (near-copies, no exact ties);
        # the sort flags it as crowded and the sweep's few-groups direction takes the shared-group path for it
        a = (rng.rand(b, n, 3) - 0.5).astype(np.float32)
        spots = (rng.rand(b, 120, 3) - 0.5)
        sid = np.where(rng.rand(b, m) < 0.95, (np.arange(m) * 120 // m)[None], rng.randint(0, 120, (b, m)))  # coherent runs
        c = spots[np.arange(b)[:, None], sid] + 6e-6 * rng.randn(b, m, 3)
        return a, c.astype(np.float32)
    if kind == "same":  # the same cloud on both sides: every winner at distance 0, one source per destination
        a = rng.randn(b, n, 3).astype(np.float32)
        return a, (a[:, rng.permutation(n)[:m]] if m <= n else np.concatenate([a] * (m // n + 1), 1)[:, :m].copy())
    raise ValueError(kind)


def run_step(a, c, gd1, gd2):
    from rfnet_amd import _raw as R
    b, n, m = a.shape[0], a.shape[1], c.shape[1]
    plan = R.ChamferStep(b, n, m, "cuda")
    ta, tc, tg1, tg2 = cu(a), cu(c), cu(gd1), cu(gd2)
    for _ in range(2):  # the plan's buffers and workspace are reused
        out = plan(ta, tc, tg1, tg2)
    torch.cuda.synchronize()
    return [t.cpu().numpy() for t in out], (ta, tc, tg1, tg2)


def check(orc, a, c, seed=3):
    from rfnet_amd import _raw as R
    from rfnet_amd._lib import lib
    b, n, m = a.shape[0], a.shape[1], c.shape[1]
    assert lib.rf_chamfer_step_workspace_bytes(b, n, m) > lib.rf_nn_distance_workspace_bytes(b, n, m), \
        "shape does not take the sorted-space step"
    rng = np.random.RandomState(seed)
    gd1 = (rng.rand(b, n) + 0.25).astype(np.float32) * rng.choice([-1, 1], (b, n)).astype(np.float32)
    gd2 = (rng.rand(b, m) + 0.25).astype(np.float32)
    (d1, i1, d2, i2, g1, g2), (ta, tc, tg1, tg2) = run_step(a, c, gd1, gd2)
    e = orc.nn_distance(a, c)
    for got, exp, name in zip((d1, i1, d2, i2), e, ("dist1", "idx1", "dist2", "idx2")):
        assert np.array_equal(got, exp), name
    o1, o2 = orc.nn_distance_grad(a, c, gd1, e[1], gd2, e[3])
    assert np.allclose(g1, o1, rtol=1e-5, atol=1e-5 * np.abs(o1).max()), np.abs(g1 - o1).max()
    assert np.allclose(g2, o2, rtol=1e-5, atol=1e-5 * np.abs(o2).max()), np.abs(g2 - o2).max()
    r1, r2 = R.nn_distance_grad(ta, tc, tg1, cu(i1), tg2, cu(i2))
    assert np.allclose(g1, r1.cpu().numpy(), rtol=1e-5, atol=1e-5 * np.abs(o1).max())
    assert np.allclose(g2, r2.cpu().numpy(), rtol=1e-5, atol=1e-5 * np.abs(o2).max())


# every launch shape of sort / sweep / backward: shared groups (few queries), one-wave groups, split sort
# (> 8192 points), the non-register sort (> 16384 points), ragged sizes (padding inside and behind the set)
SHAPES = [(4, 2048, 4096), (8, 4096, 4096), (3, 3000, 16384), (2, 16384, 16384), (3, 20000, 17000), (14, 1100, 9001),
          (8, 1024, 2500)]


@pytest.mark.parametrize("b,n,m", SHAPES)
def test_step_randn(orc, b, n, m):
    a, c = make("randn", 100 + n, b, n, m)
    check(orc, a, c)


@pytest.mark.parametrize("kind", ["uniform", "dup", "lattice", "clustered", "same", "collapsed"])
@pytest.mark.parametrize("b,n,m", [(4, 2048, 4096), (3, 3000, 16384)])
def test_step_distributions(orc, kind, b, n, m):
    a, c = make(kind, 7, b, n, m)
    check(orc, a, c)


def test_step_c2_full_size(orc):
    """BASELINE.json configs[1] itself: B=32, 2048 vs 16384, randn seed 100, upstream gradients of ones."""
    rng = np.random.RandomState(100)
    a = rng.randn(32, 2048, 3).astype(np.float32)
    c = rng.randn(32, 16384, 3).astype(np.float32)
    gd1, gd2 = np.ones((32, 2048), np.float32), np.ones((32, 16384), np.float32)
    (d1, i1, d2, i2, g1, g2), _ = run_step(a, c, gd1, gd2)
    e = orc.nn_distance(a, c)
    for got, exp in zip((d1, i1, d2, i2), e):
        assert np.array_equal(got, exp)
    o1, o2 = orc.nn_distance_grad(a, c, gd1, e[1], gd2, e[3])
    assert np.allclose(g1, o1, rtol=1e-5, atol=1e-5 * np.abs(o1).max())
    assert np.allclose(g2, o2, rtol=1e-5, atol=1e-5 * np.abs(o2).max())
    # size-independent properties: the two gradients of one Chamfer sum to zero per sample (every term enters
    # one set with +, the other with -), and a point nobody chose carries its own term only
    tot = g1.astype(np.float64).sum(1) + g2.astype(np.float64).sum(1)
    assert np.abs(tot).max() < 1e-2 * np.abs(o2).max()
    chosen = np.zeros((32, 16384), bool)
    np.put_along_axis(chosen, i1.astype(np.int64), True, 1)
    own = 2.0 * (c - np.take_along_axis(a, i2.astype(np.int64)[..., None], 1))
    assert np.allclose(g2[~chosen], own[~chosen], rtol=1e-6, atol=1e-7)


def test_step_nonfinite_policy_matches_the_two_op_path():
    """A NaN query gets (NaN, index 0) in the forward; its gradient terms then go
    where NnDistanceGrad sends them for index 0 -- the point with ORIGINAL index 0 of the other set."""
    from rfnet_amd import _raw as R
    a, c = make("randn", 5, 3, 2048, 4096)
    a[1, 77, 1] = np.nan
    c[2, 4000, 0] = np.nan
    gd1 = np.ones(a.shape[:2], np.float32)
    gd2 = np.ones(c.shape[:2], np.float32)
    (d1, i1, d2, i2, g1, g2), (ta, tc, tg1, tg2) = run_step(a, c, gd1, gd2)
    assert np.isnan(d1[1, 77]) and i1[1, 77] == 0 and np.isnan(d2[2, 4000]) and i2[2, 4000] == 0
    r1, r2 = R.nn_distance_grad(ta, tc, tg1, cu(i1), tg2, cu(i2))
    r1, r2 = r1.cpu().numpy(), r2.cpu().numpy()
    assert np.array_equal(np.isnan(g1), np.isnan(r1)) and np.array_equal(np.isnan(g2), np.isnan(r2))
    assert np.isnan(g1[1, 77]).any() and np.isnan(g2[1, 0]).any()      # own term, and the scatter into point 0
    assert np.isnan(g2[2, 4000]).any() and np.isnan(g1[2, 0]).any()
    ok1, ok2 = ~np.isnan(r1), ~np.isnan(r2)
    assert np.allclose(g1[ok1], r1[ok1], rtol=1e-5, atol=1e-4) and np.allclose(g2[ok2], r2[ok2], rtol=1e-5, atol=1e-4)


def test_small_shapes_keep_the_two_op_step(orc):
    """Below the culled sweep's sizes rf_chamfer_step is the dense forward + the original-order backward."""
    from rfnet_amd._lib import lib
    assert lib.rf_chamfer_step_workspace_bytes(2, 300, 700) == lib.rf_nn_distance_workspace_bytes(2, 300, 700)


def test_step_accepts_the_forward_workspace_size():
    """A caller that sizes the workspace with rf_nn_distance_workspace_bytes (what rounds 1-2 documented) on a shape that
    takes the culled sweep gets the two ops back to back -- same results, not RF_EWORKSPACE (chamfer_ext.hip)."""
    from rfnet_amd import _raw as R
    from rfnet_amd._lib import lib
    b, n, m = 4, 2048, 4096
    small, big = lib.rf_nn_distance_workspace_bytes(b, n, m), lib.rf_chamfer_step_workspace_bytes(b, n, m)
    assert 0 < small < big
    a, c = make("randn", 5, b, n, m)
    gd1, gd2 = np.ones((b, n), np.float32), np.ones((b, m), np.float32)
    ta, tc, tg1, tg2 = cu(a), cu(c), cu(gd1), cu(gd2)
    ref = R.ChamferStep(b, n, m, "cuda")(ta, tc, tg1, tg2)
    ref = [t.clone() for t in ref]
    out = [torch.empty_like(t) for t in ref]
    ws = torch.empty(small, dtype=torch.uint8, device="cuda")
    st = lib.rf_chamfer_step(b, n, m, ta.data_ptr(), tc.data_ptr(), tg1.data_ptr(), tg2.data_ptr(), *[t.data_ptr() for t in out],
                             ws.data_ptr(), small, torch.cuda.current_stream().cuda_stream)
    assert st == 0
    torch.cuda.synchronize()
    for x, y in zip(out[:4], ref[:4]):
        assert torch.equal(x, y)
    for x, y in zip(out[4:], ref[4:]):
        assert torch.allclose(x, y, rtol=1e-5, atol=1e-5 * float(y.abs().max()))
