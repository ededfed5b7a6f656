"""GPU parity, randomised: many seeded random shapes per op against the CPU oracle.

The shape lists of the other test files are hand-picked around the tiling boundaries; this file
draws shapes at random (log-uniform sizes, ragged batches, duplicated points, lattice ties) so
that every launch-plan branch (in-place / packed clouds, candidate splits with and without the
in-workgroup merge, tile sizes and source slices of the backward, LDS-staged and direct ball
query, register-resident FPS variants) is hit by some case.  Bars as in the per-op files:
bit-exact for indices and Chamfer distances, fp32 tolerance for accumulated gradients.
"""
import os

import numpy as np
import pytest
import torch

from conftest import assert_rel, strict_bar_report

pytestmark = pytest.mark.gpu

# RF_FUZZ_SCALE=10 runs ten times the seeds of every test below (an extended parity run; the default suite is what it was)
FUZZ_SCALE = max(1, int(os.environ.get("RF_FUZZ_SCALE", "1")))


def cu(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def _logint(rng, lo, hi):
    return int(round(np.exp(rng.uniform(np.log(lo), np.log(hi)))))


def _cloud(rng, b, n, kind):
    if kind == 0:
        return rng.randn(b, n, 3).astype(np.float32)
    if kind == 1:  # lattice: masses of exact ties
        return rng.randint(0, 5, size=(b, n, 3)).astype(np.float32)
    x = rng.rand(b, n, 3).astype(np.float32)  # duplicated points, as resample_pcd produces
    if n > 3:
        x[:, n // 2:] = x[:, : n - n // 2]
    return x


@pytest.mark.parametrize("seed", range(80 * FUZZ_SCALE))
def test_fuzz_nn_distance_and_grad(orc, seed):
    from rfnet_amd import _raw
    rng = np.random.RandomState(1000 + seed)
    b = rng.randint(1, 6)
    n, m = _logint(rng, 1, 6000), _logint(rng, 1, 6000)
    if seed % 8 == 0:  # sizes that fit the tiling: the in-place path
        n, m = 512 * rng.randint(1, 5), 32 * rng.randint(1, 40)
    kind = seed % 3
    a, c = _cloud(rng, b, n, kind), _cloud(rng, b, m, kind)
    got = [t.cpu().numpy() for t in _raw.nn_distance(cu(a), cu(c))]
    exp = orc.nn_distance(a, c)
    for g, e, name in zip(got, exp, ("dist1", "idx1", "dist2", "idx2")):
        assert np.array_equal(g, e), f"seed {seed} b={b} n={n} m={m} {name}: {np.sum(g != e)} mismatches"
    gd1, gd2 = rng.randn(b, n).astype(np.float32), rng.randn(b, m).astype(np.float32)
    g1, g2 = _raw.nn_distance_grad(cu(a), cu(c), cu(gd1), cu(exp[1]), cu(gd2), cu(exp[3]))
    o1, o2 = orc.nn_distance_grad(a, c, gd1, exp[1], gd2, exp[3])
    scale = 1e-5 * max(1.0, float(np.abs(o1).max()), float(np.abs(o2).max()))
    assert_rel(g1.cpu().numpy(), o1, 1e-5, scale, what=f"seed {seed} grad1")
    assert_rel(g2.cpu().numpy(), o2, 1e-5, scale, what=f"seed {seed} grad2")


@pytest.mark.parametrize("seed", range(32 * FUZZ_SCALE))
def test_fuzz_fps_gather(orc, seed):
    from rfnet_amd import _raw
    rng = np.random.RandomState(2000 + seed)
    b = rng.randint(1, 5)
    n = _logint(rng, 1, 9000)
    npoint = rng.randint(1, min(n, 300) + 1)
    x = _cloud(rng, b, n, seed % 3)
    idx = _raw.farthest_point_sample(npoint, cu(x))
    oi = orc.farthest_point_sample(npoint, x)
    assert np.array_equal(idx.cpu().numpy(), oi), f"seed {seed} b={b} n={n} npoint={npoint}"
    out = _raw.gather_point(cu(x), idx)
    assert np.array_equal(out.cpu().numpy(), orc.gather_point(x, oi))


@pytest.mark.parametrize("seed", range(32 * FUZZ_SCALE))
def test_fuzz_query_ball_group(orc, seed):
    from rfnet_amd import _raw
    rng = np.random.RandomState(3000 + seed)
    b = rng.randint(1, 4)
    n, m = _logint(rng, 1, 4000), _logint(rng, 1, 300)
    ns = [1, 8, 32, 64, 65, 200][seed % 6]
    r = float(rng.uniform(0.02, 0.6))
    ds = rng.rand(b, n, 3).astype(np.float32)
    q = rng.rand(b, m, 3).astype(np.float32)
    k = min(n, m) // 2
    q[:, :k] = ds[:, :k]  # queries on dataset points: d = 0 hits, clamp path
    idx, cnt = _raw.query_ball_point(r, ns, cu(ds), cu(q))
    oi, oc = orc.query_ball_point(r, ns, ds, q, fill=0)
    assert np.array_equal(cnt.cpu().numpy(), oc), f"seed {seed}"
    assert np.array_equal(idx.cpu().numpy(), oi), f"seed {seed} b={b} n={n} m={m} ns={ns} r={r}"
    pts = rng.randn(b, n, 5).astype(np.float32)
    grp = _raw.group_point(cu(pts), idx)
    assert np.array_equal(grp.cpu().numpy(), orc.group_point(pts, oi))


@pytest.mark.parametrize("seed", range(40 * FUZZ_SCALE))
def test_fuzz_query_ball_boxes_and_one_call(orc, seed):
    """The boxed ball query (every dataset size from 64 up, whatever the auto rule would pick) and the one-call
    sample-and-group on random shapes, cloud kinds (uniform / lattice ties / duplicated halves / one tight cluster) and radii
    from "a few hits" to "every ball holds the cloud": idx, pts_cnt against the oracle; the one call against the four ops."""
    from rfnet_amd import _raw
    rng = np.random.RandomState(7000 + seed)
    b = rng.randint(1, 5)
    n, m = _logint(rng, 64, 20000), _logint(rng, 1, 400)
    ns = [1, 7, 32, 64][seed % 4]
    kind = seed % 4
    if kind == 0:
        ds = rng.rand(b, n, 3)
    elif kind == 1:
        ds = rng.randint(0, 7, size=(b, n, 3)) / 6.0
    elif kind == 2:
        ds = rng.rand(b, n, 3)
        ds[:, n // 2:] = ds[:, : n - n // 2]
    else:
        ds = rng.rand(b, n, 3)
        ds[:, : n // 3] = 0.3 + 1e-3 * rng.randn(b, n // 3, 3)
    ds = ds.astype(np.float32)
    q = rng.rand(b, m, 3).astype(np.float32)
    k = min(n, m) // 2
    q[:, :k] = ds[:, rng.permutation(n)[:k]]
    r = float(np.exp(rng.uniform(np.log(0.01), np.log(1.5))))
    idx, cnt = _raw.query_ball_point(r, ns, cu(ds), cu(q), form="boxes")
    oi, oc = orc.query_ball_point(r, ns, ds, q, fill=0)
    assert np.array_equal(cnt.cpu().numpy(), oc), f"seed {seed} b={b} n={n} m={m} ns={ns} r={r} kind={kind}"
    assert np.array_equal(idx.cpu().numpy(), oi), f"seed {seed} b={b} n={n} m={m} ns={ns} r={r} kind={kind}"
    npoint = rng.randint(1, min(n, 200) + 1)
    x = cu(ds)
    fi = _raw.farthest_point_sample(npoint, x)
    nx = _raw.gather_point(x, fi)
    gi, gc = _raw.query_ball_point(r, ns, x, nx, form="scan")
    gx = _raw.group_point(x, gi)
    one = _raw.sample_and_group(npoint, r, ns, x, aux_stream=torch.cuda.Stream() if seed % 2 else None)
    torch.cuda.synchronize()
    for name, w, g in zip(("fps_idx", "new_xyz", "idx", "pts_cnt", "grouped_xyz"), (fi, nx, gi, gc, gx), one):
        assert torch.equal(w, g), f"seed {seed} {name} b={b} n={n} npoint={npoint} ns={ns} r={r} kind={kind}"


@pytest.mark.parametrize("seed", range(24 * FUZZ_SCALE))
def test_fuzz_three_nn_interpolate(orc, seed):
    from rfnet_amd import _raw
    rng = np.random.RandomState(4000 + seed)
    b = rng.randint(1, 4)
    n, m = _logint(rng, 1, 3000), _logint(rng, 1, 2000)
    a, c = _cloud(rng, b, n, seed % 3), _cloud(rng, b, m, seed % 3)
    d, i = _raw.three_nn(cu(a), cu(c))
    od, oi = orc.three_nn(a, c)
    assert np.array_equal(i.cpu().numpy(), oi), f"seed {seed} b={b} n={n} m={m}"
    assert np.array_equal(d.cpu().numpy(), od)
    db, ib = _raw.three_nn(cu(a), cu(c), form="boxes")  # over sorted copies of the two sets: the same bits
    assert torch.equal(ib, i) and torch.equal(db.view(torch.int32), d.view(torch.int32)), f"boxes: seed {seed} b={b} n={n} m={m}"
    if m >= 3:
        ch = rng.randint(1, 20)
        pts = rng.randn(b, m, ch).astype(np.float32)
        w = rng.rand(b, n, 3).astype(np.float32)
        out = _raw.three_interpolate(cu(pts), i, cu(w))
        assert np.array_equal(out.cpu().numpy(), orc.three_interpolate(pts, oi, w))


def _match_close(got, exp, what):
    """`match` bar for random inputs.  The annealing schedule is ill-conditioned in a few entries:
    a clamp (`min(remainR/(t+1e-9), 1)`, `max(0, remain - t)`) that flips on a last-bit difference of
    a row sum moves some mass between neighbouring entries.  Measured on seed 6 below (3 x 274 x 274):
    the fp32 ORACLE itself is up to 1.9e-4 away from a float64 evaluation of the same schedule in
    1-6 entries per sample, the GPU up to 3.6e-4 in 1-4 entries, oracle vs GPU 5.1e-4 in 8 of 75 076
    entries -- while every row/column sum agrees to 1e-6.  So: >= 99.9 % of the entries inside
    abs 1e-6 + rel 1e-4, every entry inside 2e-3 of a unit mass, and all marginals inside 1e-5.
    (With RF_FUZZ_SCALE=60, 11 of 960 seeds of the EMD chain test leave the marginal bar by up to 2e-4 -- clamp flips that move
    mass between columns -- identically on the round-4 kernels; their cost stays within 3.5e-7 of the oracle's:
    profiles/r05_soak.txt.)"""
    got, exp = np.asarray(got, np.float64), np.asarray(exp, np.float64)
    strict_bar_report(what, got, exp)  # the un-relaxed bar, always reported
    err = np.abs(got - exp)
    frac = np.mean(err <= 1e-6 + 1e-4 * np.abs(exp))
    assert frac >= 0.999, f"{what}: only {frac:.5f} of the entries inside abs 1e-6 + rel 1e-4"
    assert err.max() <= 2e-3, f"{what}: max abs err {err.max():.3e}"
    assert_rel(got.sum(1), exp.sum(1), 1e-5, 1e-5, what=what + " column sums")
    assert_rel(got.sum(2), exp.sum(2), 1e-5, 1e-5, what=what + " row sums")


@pytest.mark.parametrize("seed", range(16 * FUZZ_SCALE))
def test_fuzz_emd_chain_and_fused(orc, seed):
    from rfnet_amd import _raw
    rng = np.random.RandomState(5000 + seed)
    b = rng.randint(1, 4)
    n, m = _logint(rng, 2, 700), _logint(rng, 2, 700)
    if seed % 3 == 0:
        m = n  # the model's use (equal sizes)
    a = (rng.random_sample((b, n, 3)) - 0.5).astype(np.float32)
    c = (rng.random_sample((b, m, 3)) - 0.5).astype(np.float32)
    om = orc.approx_match(a, c)
    match = _raw.approx_match(cu(a), cu(c))
    gm = match.cpu().numpy()
    _match_close(gm, om, f"seed {seed} n={n} m={m} match")
    ocost = orc.match_cost(a, c, om)
    assert_rel(_raw.match_cost(cu(a), cu(c), match).cpu().numpy(), ocost, 1e-5, what="cost")
    # the fused op against the oracle's match_cost / match_cost_grad evaluated ON THE GPU's OWN
    # match: isolates the fused kernel from the ill-conditioning of the match itself
    fcost, g1, g2 = _raw.earth_mover(cu(a), cu(c), with_grad=True)
    assert_rel(fcost.cpu().numpy(), ocost, 1e-5, what=f"seed {seed} n={n} m={m} fused cost")
    assert_rel(fcost.cpu().numpy(), orc.match_cost(a, c, gm), 1e-5, what="fused cost vs oracle on gpu match")
    o1, o2 = orc.match_cost_grad(a, c, gm)
    assert_rel(g1.cpu().numpy(), o1, 1e-4, 1e-5 * max(1, m // n), what="fused grad1")
    assert_rel(g2.cpu().numpy(), o2, 1e-4, 1e-5 * max(1, n // m), what="fused grad2")


@pytest.mark.parametrize("seed", range(24 * FUZZ_SCALE))
def test_fuzz_match_cost_grad_both_forms(orc, seed):
    """match_cost_grad over random shapes on both sides of its dispatch rule (approxmatch.hip mcg_launch: whole rows per workgroup
    when n % 4 == 0 and the 1024-k blocks are at least 3/4 alive, the LDS-tile form otherwise), arbitrary non-negative `match`,
    duplicated points (coincident pairs: rsq(max(d2, 1e-20)) * 0), against the oracle."""
    from rfnet_amd import _raw
    rng = np.random.RandomState(7000 + seed)
    b = rng.randint(1, 4)
    n = _logint(rng, 200, 3300)
    if seed % 3:
        n = max(4, n // 4 * 4)
    m = _logint(rng, 2, 1500)
    a = _cloud(rng, b, n, 0 if seed % 4 else 2) * np.float32(0.3)
    c = _cloud(rng, b, m, 0) * np.float32(0.3)
    if seed % 4 == 0 and m > 2:
        c[:, 1] = a[:, 0]
    mt = (rng.random_sample((b, m, n)) ** 6).astype(np.float32) / np.float32(n)
    g1, g2 = _raw.match_cost_grad(cu(a), cu(c), cu(mt))
    o1, o2 = orc.match_cost_grad(a, c, mt)
    assert_rel(g1.cpu().numpy(), o1, 1e-4, 2e-6 * max(1, m // 256), what=f"seed {seed} {b}x{n}x{m} grad1")
    assert_rel(g2.cpu().numpy(), o2, 1e-4, 2e-6 * max(1, n // 256), what=f"seed {seed} {b}x{n}x{m} grad2")


@pytest.mark.parametrize("seed", range(10 * FUZZ_SCALE))
def test_fuzz_scatter_gradients_sorted_slots(orc, seed):
    """group_point_grad / three_interpolate_grad on their sorted-slots route (scatter_rows.hip) at random shapes past the route's
    threshold: against the oracle and against the atomic route; slot counts around the register-resident limit of the sort (32768
    slots per sample), key spaces cut over 1..8 workgroups, channel counts with and without the 4-wide form, indices with long runs
    on one row."""
    from rfnet_amd import _raw
    from rfnet_amd._lib import lib
    rng = np.random.RandomState(7000 + seed)
    if seed % 2 == 0:
        n = _logint(rng, 64, 60000)
        c = int(rng.choice([3, 8, 13, 32, 64, 67]))
        ns = int(rng.choice([1, 8, 32, 64]))
        m = _logint(rng, 64, 40000 // ns + 64)
        b = max(1, int(np.ceil((1 << 22) / (m * ns * c))) + rng.randint(0, 3))
        b = min(b, 64)
        if b * m * ns * c < (1 << 22) and b * m * ns < (1 << 19):
            pytest.skip("below the route's threshold")
        idx = rng.randint(0, n, size=(b, m, ns)).astype(np.int32)
        if seed % 4 == 0:
            idx[:, : m // 3] = rng.randint(0, min(n, 5), size=(b, m // 3, ns))  # a few popular rows
        go = rng.randn(b, m, ns, c).astype(np.float32)
        assert lib.rf_grouppoint_grad_workspace_bytes(b, n, c, m, ns) > 0
        pts = np.zeros((b, n, c), np.float32)
        got = _raw.group_point_grad(cu(pts), cu(idx), cu(go)).cpu().numpy()
        want = orc.group_point_grad(pts, idx, go)
        scale = max(1.0, float(np.abs(want).max()))
        assert np.abs(got - want).max() <= 1e-5 * scale + 1e-6, f"seed {seed} b={b} n={n} m={m} ns={ns} c={c}"
        return
    n = _logint(rng, 3000, 40000)   # unknown points
    m = _logint(rng, 2100, 40000)   # known points: beyond the LDS tile
    c = int(rng.choice([8, 13, 32, 64]))
    b = min(32, max(1, int(np.ceil((1 << 22) / (n * 3 * c))) + rng.randint(0, 2)))
    if lib.rf_threeinterpolate_grad_workspace_bytes(b, n, c, m) == 0:
        pytest.skip("not the sorted-slots route")
    idx = rng.randint(0, m, size=(b, n, 3)).astype(np.int32)
    w = rng.rand(b, n, 3).astype(np.float32)
    go = rng.randn(b, n, c).astype(np.float32)
    pts = np.zeros((b, m, c), np.float32)
    got = _raw.three_interpolate_grad(cu(pts), cu(idx), cu(w), cu(go)).cpu().numpy()
    want = orc.three_interpolate_grad(pts, idx, w, go)
    scale = max(1.0, float(np.abs(want).max()))
    assert np.abs(got - want).max() <= 1e-5 * scale + 1e-6, f"seed {seed} b={b} n={n} m={m} c={c}"
