"""GPU: the rest of the reference's import surface (next row f3): auction_match and select_top_k,
bit-exact against the oracle; plus properties (permutation, optimality gap of the auction)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def cu(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


@pytest.mark.parametrize("b,n", [(3, 1), (2, 16), (2, 100), (3, 512), (2, 1000), (2, 1024), (2, 2048)])
def test_auction_match_vs_oracle(orc, b, n):
    from tf_ops.emd.tf_auctionmatch import auction_match
    rng = np.random.RandomState(n)
    a = rng.randn(b, n, 3).astype(np.float32)
    # the reference's own smoke setup (tf_auctionmatch.py:33-50): xyz2 = roll(xyz1 + 0.01 noise)
    c = np.roll(a + 0.01 * rng.randn(b, n, 3).astype(np.float32), 5, axis=1)
    ml, mr = auction_match(cu(a), cu(c))
    ol, orr = orc.auction_match(a, c)
    assert np.array_equal(mr.cpu().numpy(), orr) and np.array_equal(ml.cpu().numpy(), ol)
    mr = mr.cpu().numpy()
    for i in range(b):
        assert sorted(mr[i].tolist()) == list(range(n))  # a permutation
        assert np.array_equal(ml.cpu().numpy()[i][mr[i]], np.arange(n))


def test_auction_match_4096_and_optimality(orc):
    from scipy.optimize import linear_sum_assignment
    from tf_ops.emd.tf_auctionmatch import auction_match
    rng = np.random.RandomState(4096)
    a = rng.rand(1, 4096, 3).astype(np.float32)
    c = rng.rand(1, 4096, 3).astype(np.float32)  # unrelated clouds: a hard assignment
    ml, mr = auction_match(cu(a), cu(c))
    ol, orr = orc.auction_match(a, c)
    assert np.array_equal(mr.cpu().numpy(), orr) and np.array_equal(ml.cpu().numpy(), ol)
    d = np.sqrt(((a[0][:, None] - c[0][None]) ** 2).sum(-1))
    r, cc = linear_sum_assignment(d)
    got = d[orr[0], np.arange(4096)].sum()
    assert got >= d[r, cc].sum() - 1e-3 and got <= d[r, cc].sum() * 1.05  # near-optimal


def test_auction_match_rejects_undefined_sizes():
    from tf_ops.emd.tf_auctionmatch import auction_match
    x = torch.zeros(1, 1500, 3, device="cuda")
    with pytest.raises(ValueError, match="reads out of bounds"):
        auction_match(x, x)
    with pytest.raises(ValueError, match="at most 4096"):
        auction_match(torch.zeros(1, 5000, 3, device="cuda"), torch.zeros(1, 5000, 3, device="cuda"))


@pytest.mark.parametrize("b,m,n,k", [(1, 1, 1, 1), (2, 5, 50, 5), (2, 130, 700, 16), (1, 3, 16384, 8), (2, 4, 33, 40)])
def test_select_top_k_vs_oracle(orc, b, m, n, k):
    from tf_ops.grouping.tf_grouping import select_top_k
    rng = np.random.RandomState(n + k)
    d = rng.rand(b, m, n).astype(np.float32)
    d[:, :, n // 2:n // 2 + 3] = d[:, :, :3][..., : min(3, n - n // 2)] if n >= 6 else d[:, :, n // 2:n // 2 + 3]
    idx, val = select_top_k(k, cu(d))
    oi, ov = orc.select_top_k(k, d)
    assert np.array_equal(idx.cpu().numpy(), oi) and np.array_equal(val.cpu().numpy(), ov)
    kk = min(k, n)
    assert np.array_equal(ov[..., :kk], np.sort(d, -1)[..., :kk])
    # selection sort with swaps is NOT stable among equal values (a swap can carry an equal
    # element past another), so indices are checked through the values they point at
    assert np.array_equal(np.take_along_axis(d, oi.astype(np.int64), -1), ov)
    assert (np.sort(oi, -1) == np.arange(n)).all()  # every row of idx is a permutation


@pytest.mark.parametrize("b,n,m", [(1, 1, 5), (2, 3, 7), (3, 40, 100), (2, 255, 64), (2, 257, 64), (3, 4099, 500), (2, 8192, 300), (2, 8193, 300),
                                   (1, 16384, 2000), (2, 24577, 100), (2, 32768, 100), (2, 50001, 1000)])
def test_prob_sample_vs_oracle(orc, b, n, m):
    from rfnet_amd import _raw as R
    from tf_ops.sampling.tf_sampling import prob_sample
    rng = np.random.RandomState(n)
    p = rng.rand(b, n).astype(np.float32)
    if n > 10:
        p[:, 3] = 0.0
        p[:, n // 2] = -0.0  # signed zeros: x + 0.0 is not the identity on the sign bit
    r = rng.rand(b, m).astype(np.float32)
    got = prob_sample(cu(p), cu(r)).cpu().numpy()
    exp, cs = orc.prob_sample(p, r)
    assert np.array_equal(got, exp)
    # the cumulative sums themselves, bit for bit (the association order of the blocked scan)
    got2, gcs = R.prob_sample(cu(p), cu(r), return_cumsum=True)
    assert np.array_equal(gcs.cpu().numpy().view(np.uint32), np.ascontiguousarray(cs).view(np.uint32))
    assert np.array_equal(got2.cpu().numpy(), exp)
    # inverse CDF: index i is drawn with probability p[i] / sum(p)
    assert (got >= 0).all() and (got < n).all()
