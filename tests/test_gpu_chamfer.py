"""GPU parity: nn_distance forward/backward (HIP, through the C ABI) vs the CPU oracle.

Bars: idx bit-exact and dist bit-exact vs the oracle (same pinned fma chain); vs the
reference's CPU bodies (golden `ref_*`) idx equal and dist within 1e-5 relative (north_star);
gradients within rel 1e-5 / abs 1e-6 (fp32 atomics: summation order differs).
"""
import numpy as np
import pytest
import torch

from conftest import assert_rel

pytestmark = pytest.mark.gpu


def _run(xyz1, xyz2):
    from tf_ops.CD.tf_nndistance import nn_distance
    out = nn_distance(torch.from_numpy(xyz1).cuda(), torch.from_numpy(xyz2).cuda())
    return [t.cpu().numpy() for t in out]


def _check_vs_oracle(orc, a, c):
    got = _run(a, c)
    exp = orc.nn_distance(a, c)
    for g, e, name in zip(got, exp, ("dist1", "idx1", "dist2", "idx2")):
        assert g.dtype == e.dtype and g.shape == e.shape
        assert np.array_equal(g, e), f"{name}: {np.sum(g != e)} mismatches of {g.size}"
    return got


def test_c1_config_golden(orc, golden):
    g = golden("nn_distance_c1")
    got = _check_vs_oracle(orc, g["xyz1"], g["xyz2"])
    assert np.array_equal(got[1], g["ref_idx1"]) and np.array_equal(got[3], g["ref_idx2"])
    assert_rel(got[0], g["ref_dist1"], 1e-5)
    assert_rel(got[2], g["ref_dist2"], 1e-5)


@pytest.mark.parametrize("tag", ["a", "b", "dup"])
def test_ragged_and_ties_golden(orc, golden, tag):
    g = golden("nn_distance_ragged")
    got = _check_vs_oracle(orc, g[f"{tag}_xyz1"], g[f"{tag}_xyz2"])
    assert np.array_equal(got[1], g[f"{tag}_ref_idx1"])
    assert np.array_equal(got[3], g[f"{tag}_ref_idx2"])
    assert_rel(got[0], g[f"{tag}_ref_dist1"], 1e-5)


@pytest.mark.parametrize("b,n,m", [(1, 1, 1), (2, 7, 5), (3, 1, 300), (2, 300, 1), (3, 256, 1024),
                                   (2, 1025, 4097), (1, 5000, 300), (5, 64, 3000), (2, 3000, 16384),
                                   (33, 100, 17)])
def test_random_shapes(orc, b, n, m):
    rng = np.random.RandomState(b * 1000 + n + m)
    a = rng.randn(b, n, 3).astype(np.float32)
    c = rng.randn(b, m, 3).astype(np.float32)
    _check_vs_oracle(orc, a, c)


def test_many_exact_ties(orc):
    rng = np.random.RandomState(1)
    grid = rng.randint(0, 4, size=(2, 2000, 3)).astype(np.float32)  # lattice: masses of ties
    _check_vs_oracle(orc, grid[:, :900], grid[:, 900:])
    same = np.ones((1, 700, 3), np.float32)  # all distances 0: every idx must be 0
    got = _check_vs_oracle(orc, same, same[:, :333])
    assert (got[1] == 0).all() and (got[3] == 0).all()


def test_large_coordinates_and_inf(orc):
    rng = np.random.RandomState(3)
    a = (rng.randn(1, 300, 3) * 1e18).astype(np.float32)
    c = (rng.randn(1, 200, 3) * 1e18).astype(np.float32)
    a[0, :5] = 3e38
    c[0, :3] = -3e38  # d2 overflows to +inf for these pairs
    _check_vs_oracle(orc, a, c)


def test_c2_config_full_size(orc):
    """BASELINE.json configs[1] (B=32, 2048 vs 16384): oracle on two batch elements, then
    size-independent properties on all 32."""
    rng = np.random.RandomState(100)
    a = rng.randn(32, 2048, 3).astype(np.float32)
    c = rng.randn(32, 16384, 3).astype(np.float32)
    d1, i1, d2, i2 = _run(a, c)
    for bi in (0, 31):
        e = orc.nn_distance(a[bi:bi + 1], c[bi:bi + 1])
        assert np.array_equal(d1[bi], e[0][0]) and np.array_equal(i1[bi], e[1][0])
        assert np.array_equal(d2[bi], e[2][0]) and np.array_equal(i2[bi], e[3][0])
    # property 1: the reported distance is the distance to the reported neighbour
    nb = np.take_along_axis(c, i1[..., None].astype(np.int64), 1)
    assert_rel(d1, ((nb - a).astype(np.float64) ** 2).sum(-1), 1e-6)
    nb2 = np.take_along_axis(a, i2[..., None].astype(np.int64), 1)
    assert_rel(d2, ((nb2 - c).astype(np.float64) ** 2).sum(-1), 1e-6)
    # property 2: mutual consistency -- d2[idx1[j]] <= d1[j] and d1[idx2[k]] <= d2[k]
    assert (np.take_along_axis(d2, i1.astype(np.int64), 1) <= d1).all()
    assert (np.take_along_axis(d1, i2.astype(np.int64), 1) <= d2).all()
    assert i1.min() >= 0 and i1.max() < 16384 and i2.min() >= 0 and i2.max() < 2048


def test_north_star_size_16384_sq(orc):
    """32 x 16384 x 16384: oracle on a slice of queries of one batch element + properties."""
    rng = np.random.RandomState(5)
    a = rng.randn(32, 16384, 3).astype(np.float32)
    c = rng.randn(32, 16384, 3).astype(np.float32)
    d1, i1, d2, i2 = _run(a, c)
    e = orc.nn_distance(a[7:8, 1000:1400], c[7:8])
    assert np.array_equal(d1[7, 1000:1400], e[0][0]) and np.array_equal(i1[7, 1000:1400], e[1][0])
    e = orc.nn_distance(a[19:20], c[19:20, 16000:16384])
    assert np.array_equal(d2[19, 16000:], e[2][0]) and np.array_equal(i2[19, 16000:], e[3][0])
    assert (np.take_along_axis(d2, i1.astype(np.int64), 1) <= d1).all()
    assert (np.take_along_axis(d1, i2.astype(np.int64), 1) <= d2).all()
    # identical clouds: every point is its own nearest neighbour at distance 0
    d1, i1, d2, i2 = _run(a[:2], a[:2])
    assert (d1 == 0).all() and (d2 == 0).all()
    assert np.array_equal(i1[0], np.arange(16384)) and np.array_equal(i2[1], np.arange(16384))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_grad_golden_and_oracle(orc, golden, tag):
    from tf_ops.CD.tf_nndistance import nn_distance_grad
    g = golden("nn_distance_ragged")
    args = (g[f"{tag}_xyz1"], g[f"{tag}_xyz2"], g[f"{tag}_gd1"], g[f"{tag}_ref_idx1"],
            g[f"{tag}_gd2"], g[f"{tag}_ref_idx2"])
    g1, g2 = nn_distance_grad(*args)  # numpy in -> numpy out (staged through the GPU)
    assert isinstance(g1, np.ndarray)
    # fp32 atomics sum the scatter contributions (up to ~50 per point here, of both signs) in
    # arrival order: the error bar scales with the magnitude of the terms, not of their sum
    a1 = 1e-5 * np.abs(g[f"{tag}_ref_grad1"]).max()
    a2 = 1e-5 * np.abs(g[f"{tag}_ref_grad2"]).max()
    assert_rel(g1, g[f"{tag}_ref_grad1"], 1e-5, a1, what="grad_xyz1 vs reference CPU op")
    assert_rel(g2, g[f"{tag}_ref_grad2"], 1e-5, a2, what="grad_xyz2 vs reference CPU op")
    o1, o2 = orc.nn_distance_grad(*args)
    assert_rel(g1, o1, 1e-5, a1)
    assert_rel(g2, o2, 1e-5, a2)


def test_autograd_matches_op_gradient(orc):
    from tf_ops.CD.tf_nndistance import nn_distance
    rng = np.random.RandomState(9)
    a = rng.randn(3, 400, 3).astype(np.float32)
    c = rng.randn(3, 1500, 3).astype(np.float32)
    ta = torch.from_numpy(a).cuda().requires_grad_(True)
    tc = torch.from_numpy(c).cuda().requires_grad_(True)
    d1, i1, d2, i2 = nn_distance(ta, tc)
    w1 = torch.from_numpy(rng.rand(3, 400).astype(np.float32)).cuda()
    w2 = torch.from_numpy(rng.rand(3, 1500).astype(np.float32)).cuda()
    ((d1 * w1).sum() + (d2 * w2).sum()).backward()
    o1, o2 = orc.nn_distance_grad(a, c, w1.cpu().numpy(), i1.cpu().numpy(), w2.cpu().numpy(),
                                  i2.cpu().numpy())
    assert_rel(ta.grad.cpu().numpy(), o1, 1e-5, 1e-5 * np.abs(o1).max())
    assert_rel(tc.grad.cpu().numpy(), o2, 1e-5, 1e-5 * np.abs(o2).max())
    # the reference bench's loss (tf_nndistance.py:50): reduce_sum(dist1)+reduce_sum(dist2)
    # has gradient 2*(a - nn(a)) summed with the scatter from the other direction
    assert not i1.requires_grad


def test_pc_distance_alias_and_errors():
    from pc_distance.tf_nndistance import nn_distance as nn2
    from tf_ops.CD.tf_nndistance import nn_distance
    assert nn2 is nn_distance
    x = torch.zeros(2, 5, 3, device="cuda")
    with pytest.raises(ValueError, match="NnDistance only accepts 3d point set xyz2"):
        nn_distance(x, torch.zeros(2, 5, 2, device="cuda"))
    with pytest.raises(ValueError, match="same batch size"):
        nn_distance(x, torch.zeros(3, 5, 3, device="cuda"))


def test_large_clouds_65536(orc):
    """Beyond any config size: 65536 vs 32768 points (oracle on slices, mutual-consistency property)."""
    rng = np.random.RandomState(11)
    a = rng.randn(1, 65536, 3).astype(np.float32)
    c = rng.randn(1, 32768, 3).astype(np.float32)
    d1, i1, d2, i2 = _run(a, c)
    e = orc.nn_distance(a[:, 60000:60300], c)
    assert np.array_equal(d1[0, 60000:60300], e[0][0]) and np.array_equal(i1[0, 60000:60300], e[1][0])
    e = orc.nn_distance(a, c[:, 100:300])
    assert np.array_equal(d2[0, 100:300], e[2][0]) and np.array_equal(i2[0, 100:300], e[3][0])
    assert (np.take_along_axis(d2, i1.astype(np.int64), 1) <= d1).all()
    assert (np.take_along_axis(d1, i2.astype(np.int64), 1) <= d2).all()
