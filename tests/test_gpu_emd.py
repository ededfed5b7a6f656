"""GPU parity: approx_match / match_cost / match_cost_grad vs the CPU oracle.

Tolerances: the hardware exp2/rsq approximations differ from the oracle's
exp2f / 1/sqrtf, and row sums are accumulated per column segment, so
   match entries: abs 1e-6 + rel 1e-4;   cost: rel 1e-5 (north_star);
   gradients:     rel 1e-4 + abs 1e-5 of the row scale.
"""
import numpy as np
import pytest
import torch

from conftest import assert_rel, strict_bar_report

pytestmark = pytest.mark.gpu


def cu(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


@pytest.mark.parametrize("tag", ["sq", "rag"])
def test_emd_golden(golden, tag):
    from pc_distance.tf_approxmatch import approx_match, match_cost, match_cost_grad
    g = golden("emd")
    a, c = cu(g[f"{tag}_xyz1"]), cu(g[f"{tag}_xyz2"])
    match = approx_match(a, c)
    assert tuple(match.shape) == g[f"{tag}_cuda_match_mn"].shape  # (b, m, n)
    assert_rel(match.cpu().numpy(), g[f"{tag}_cuda_match_mn"], 1e-4, 1e-6, what="match")
    cost = match_cost(a, c, match)
    assert_rel(cost.cpu().numpy(), g[f"{tag}_cuda_cost"], 1e-5, what="cost")
    # on the oracle's own match, so only the cost kernel is compared
    cost2 = match_cost(a, c, cu(g[f"{tag}_cuda_match_mn"]))
    assert_rel(cost2.cpu().numpy(), g[f"{tag}_cuda_cost"], 1e-5, what="cost on oracle match")
    g1, g2 = match_cost_grad(a, c, cu(g[f"{tag}_cuda_match_mn"]))
    assert_rel(g1.cpu().numpy(), g[f"{tag}_cuda_grad1"], 1e-4, 1e-5, what="grad1")
    assert_rel(g2.cpu().numpy(), g[f"{tag}_cuda_grad2"], 1e-4, 1e-5, what="grad2")


@pytest.mark.parametrize("b,n,m", [(1, 1, 1), (2, 5, 3), (2, 64, 64), (3, 100, 1000), (2, 1024, 1024),
                                   (2, 777, 130), (1, 130, 777)])
def test_emd_random_shapes(orc, b, n, m):
    from pc_distance.tf_approxmatch import approx_match, match_cost, match_cost_grad
    rng = np.random.RandomState(n * 3 + m)
    a = (rng.random_sample((b, n, 3)) - 0.5).astype(np.float32)
    c = (rng.random_sample((b, m, 3)) - 0.5).astype(np.float32)
    om = orc.approx_match(a, c)
    match = approx_match(cu(a), cu(c))
    assert_rel(match.cpu().numpy(), om, 1e-4, 1e-6, what="match")
    assert_rel(match_cost(cu(a), cu(c), match).cpu().numpy(), orc.match_cost(a, c, om), 1e-5)
    g1, g2 = match_cost_grad(cu(a), cu(c), cu(om))
    o1, o2 = orc.match_cost_grad(a, c, om)
    assert_rel(g1.cpu().numpy(), o1, 1e-4, 1e-5)
    assert_rel(g2.cpu().numpy(), o2, 1e-4, 1e-5)


def test_c4_config_full_size(orc):
    """BASELINE.json configs[3]: B=32, 2048 vs 2048, reference 10-level schedule.  Oracle on one
    batch element (1.3e8 exp evaluations), properties on all 32."""
    from pc_distance.tf_approxmatch import approx_match, match_cost
    rng = np.random.RandomState(100)
    a = (rng.random_sample((32, 2048, 3)) - 0.5).astype(np.float32)
    c = (rng.random_sample((32, 2048, 3)) - 0.5).astype(np.float32)
    ta, tc = cu(a), cu(c)
    match = approx_match(ta, tc)
    cost = match_cost(ta, tc, match)
    om = orc.approx_match(a[3:4], c[3:4])
    # 4.2M entries, exp arguments down to -16384*d2: the 1-ulp differences between v_exp_f32
    # and the oracle's exp2f (and the segment-wise summation order) are amplified by |argument|
    # and by the clamps of the ten-level annealing, so a handful of entries (~1e-6 of them)
    # move by up to ~1e-4 of a unit mass.  Bar at this size: every entry within 2e-4 absolute
    # (entries are masses in [0,1]), >= 99.99 % within the small-case bar abs 1e-6 + rel 1e-4,
    # and the cost (what the loss uses) within 1e-5 relative.
    gm = match[3:4].cpu().numpy()
    strict_bar_report("C4 match, sample 3 (32x2048x2048)", gm, om)  # the un-relaxed bar, always reported
    assert np.abs(gm - om).max() < 2e-4
    tight = np.abs(gm - om) <= 1e-6 + 1e-4 * np.abs(om)
    # the un-relaxed bar's own count, held where rounds 4 and 5 measured it (21 of 4 194 304 outside = 99.99950 % inside, with
    # and without round 5's squared level weights in am_match): a change that widens it fails here, not in a warning
    assert int((~tight).sum()) <= 32, f"{int((~tight).sum())} entries outside abs 1e-6 + rel 1e-4 (21 in rounds 4-5)"
    assert_rel(cost[3:4].cpu().numpy(), orc.match_cost(a[3:4], c[3:4], om), 1e-5, what="cost[3]")
    # doubly stochastic (n == m): every point ships and receives unit mass
    rows = match.sum(1).cpu().numpy()
    cols = match.sum(2).cpu().numpy()
    assert_rel(rows, np.ones_like(rows), 1e-3)
    assert_rel(cols, np.ones_like(cols), 1e-3)
    assert (match >= 0).all()
    # EMD cost per point of two uniform clouds in the unit cube is O(n^-1/3): sanity band
    per_point = cost.cpu().numpy() / 2048
    assert (per_point > 0.02).all() and (per_point < 0.2).all()


def test_extended_schedule_50_levels(orc):
    """BASELINE's '50 Sinkhorn iters' has no reference counterpart: each of the 10
    reference levels repeated 5x, checked against the oracle on the same schedule."""
    from pc_distance.tf_approxmatch import approx_match_levels
    lv = np.repeat(orc.default_levels(), 5)
    rng = np.random.RandomState(50)
    a = (rng.random_sample((2, 200, 3)) - 0.5).astype(np.float32)
    c = (rng.random_sample((2, 200, 3)) - 0.5).astype(np.float32)
    got = approx_match_levels(cu(a), cu(c), lv.tolist())
    assert_rel(got.cpu().numpy(), orc.approx_match(a, c, levels=lv), 1e-4, 1e-6)


@pytest.mark.parametrize("nlv, kind", [(13, "irregular"), (20, "irregular"), (30, "pairs"), (37, "irregular"), (50, "fives"),
                                       (50, "irregular"), (64, "irregular"), (40, "eights")])
def test_any_schedule_is_materialised_in_one_pass(orc, nlv, kind):
    """Schedules other than the reference's ten levels go through am_match_any_kernel: every level in one pass over `match`, a
    repeated multiplier reusing its weight, a row's levels behind its last live one not formed (1, 2, 3 or 4 groups of sixteen
    levels; runs of five as their own instantiation).  Against the oracle on the same schedule, above the small-cloud kernel's
    size, with ragged sizes; rounds 1-5 took sixteen levels per launch and read the tensor back in between."""
    from pc_distance.tf_approxmatch import approx_match_levels, match_cost
    rng = np.random.RandomState(1000 + nlv)
    base = np.asarray(orc.default_levels(), np.float32)
    if kind == "fives":
        lv = np.repeat(base, 5)
    elif kind == "pairs":
        lv = np.repeat(np.concatenate([base, base[-5:]]), 2)
    elif kind == "eights":
        lv = np.repeat(base[[0, 2, 4, 6, 9]], 8)
    else:  # runs of 1..4 of a descending ladder that ends at 0
        ladder = np.abs(np.concatenate([base[:-1], base[:-1] * 0.5, base[:-1] * 0.3]))
        lv, i = [], 0
        while len(lv) < nlv - 1:
            lv += [-float(ladder[i % len(ladder)])] * int(rng.randint(1, 5))
            i += 1
        lv = np.asarray(sorted(lv[:nlv - 1]) + [0.0], np.float32)
    assert len(lv) == nlv
    b, n, m = 2, 300 + nlv, 417
    a = (rng.random_sample((b, n, 3)) - 0.5).astype(np.float32)
    c = (rng.random_sample((b, m, 3)) - 0.5).astype(np.float32)
    ta, tc = cu(a), cu(c)
    got = approx_match_levels(ta, tc, lv.tolist())
    want = orc.approx_match(a, c, levels=lv)
    g = got.cpu().numpy()
    bad = np.abs(g - want) > 1e-6 + 1e-4 * np.abs(want)
    assert bad.mean() <= 1e-3, f"{int(bad.sum())} of {bad.size} entries outside the strict bar"  # (the ill-conditioned few: test_oracle_golden.py)
    np.testing.assert_allclose(g.sum(1), want.sum(1), rtol=0, atol=2e-4)
    assert_rel(match_cost(ta, tc, got).cpu().numpy(), orc.match_cost(a, c, want), 1e-5, what=f"cost, {nlv} levels ({kind})")


def test_extended_schedule_50_levels_c4_size(orc):
    """SURVEY 8(d) C4's secondary run at ITS size: the 50-level schedule (10 reference levels x 5) at 2048 vs 2048, one
    sample against the oracle on the same schedule (6.3e8 exp evaluations on the host), marginals on all of the
    batch that was run.  Same bars as the 10-level C4 test: repeating a sharp level five times amplifies last-bit
    differences no further than the reference schedule itself does (the level's clamps saturate)."""
    from pc_distance.tf_approxmatch import approx_match_levels, match_cost
    lv = np.repeat(orc.default_levels(), 5)
    rng = np.random.RandomState(100)
    a = (rng.random_sample((4, 2048, 3)) - 0.5).astype(np.float32)
    c = (rng.random_sample((4, 2048, 3)) - 0.5).astype(np.float32)
    ta, tc = cu(a), cu(c)
    match = approx_match_levels(ta, tc, lv.tolist())
    om = orc.approx_match(a[1:2], c[1:2], levels=lv)
    gm = match[1:2].cpu().numpy()
    strict_bar_report("C4 match, 50-level schedule, sample 1 (2048x2048)", gm, om)
    assert np.abs(gm - om).max() < 2e-4
    assert (np.abs(gm - om) <= 1e-6 + 1e-4 * np.abs(om)).mean() > 0.9999
    cost = match_cost(ta, tc, match)
    assert_rel(cost[1:2].cpu().numpy(), orc.match_cost(a[1:2], c[1:2], om), 1e-5, what="cost[1], 50 levels")
    rows, cols = match.sum(1).cpu().numpy(), match.sum(2).cpu().numpy()
    assert_rel(rows, np.ones_like(rows), 1e-3)
    assert_rel(cols, np.ones_like(cols), 1e-3)
    assert (match >= 0).all()


def test_match_cost_autograd(orc):
    """earth_mover's use (vv_recon.py:392-399): cost.backward() scales the op gradients by
    grad_cost[:,None,None] and gives no gradient to match (tf_approxmatch.py:44-50)."""
    from pc_distance.tf_approxmatch import approx_match, match_cost
    rng = np.random.RandomState(12)
    a = (rng.random_sample((2, 300, 3)) - 0.5).astype(np.float32)
    c = (rng.random_sample((2, 300, 3)) - 0.5).astype(np.float32)
    ta, tc = cu(a).requires_grad_(True), cu(c).requires_grad_(True)
    match = approx_match(ta, tc)
    assert not match.requires_grad
    cost = match_cost(ta, tc, match)
    scale = torch.tensor([0.5, -2.0], device="cuda")
    (cost * scale).sum().backward()
    o1, o2 = orc.match_cost_grad(a, c, match.cpu().numpy())
    s = scale.cpu().numpy()[:, None, None]
    assert_rel(ta.grad.cpu().numpy(), o1 * s, 1e-4, 1e-5)
    assert_rel(tc.grad.cpu().numpy(), o2 * s, 1e-4, 1e-5)


def test_eval_size_16384_sq_single_sample():
    """The evaluation shape of the reference (vv_recon.py:485: earth_mover at 16384 x 16384, match =
    1 GiB per sample): one sample, size-independent properties only (the oracle would need 8e9 exp)."""
    from pc_distance.tf_approxmatch import approx_match, match_cost, match_cost_grad
    rng = np.random.RandomState(16384)
    a = cu((rng.random_sample((1, 16384, 3)) - 0.5).astype(np.float32))
    c = cu((rng.random_sample((1, 16384, 3)) - 0.5).astype(np.float32))
    match = approx_match(a, c)
    assert tuple(match.shape) == (1, 16384, 16384)
    rows, cols = match.sum(1).cpu().numpy(), match.sum(2).cpu().numpy()
    assert_rel(rows, np.ones_like(rows), 2e-3)
    assert_rel(cols, np.ones_like(cols), 2e-3)
    cost = float(match_cost(a, c, match)[0])
    assert 0.01 < cost / 16384 < 0.1  # EMD per point of two uniform clouds, O(n^-1/3)
    # linearity of match_cost in match, and consistency of the gradient with the cost
    half = float(match_cost(a, c, match * 0.5)[0])
    assert abs(half - 0.5 * cost) < 1e-5 * cost
    g1, g2 = match_cost_grad(a, c, match)
    # sum_k grad1[k] = -sum_l grad2[l] (every pair contributes +v to one and -v to the other)
    s1, s2 = g1.sum(1).cpu().numpy(), g2.sum(1).cpu().numpy()
    assert np.allclose(s1, -s2, atol=1e-2)


# ---- row f1: earth_mover fused (cost + MatchCostGrad without materialising match) -------------
@pytest.mark.parametrize("b,n,m", [(2, 5, 3), (2, 64, 64), (2, 256, 256), (3, 100, 1000), (2, 1024, 1024),
                                   (2, 777, 130), (1, 130, 777), (1, 2048, 2048), (2, 300, 257)])
def test_earth_mover_fused_vs_oracle(orc, b, n, m):
    """The fused op against the oracle's three-op chain approx_match -> match_cost / match_cost_grad."""
    from rfnet_amd import _raw
    rng = np.random.RandomState(7 * n + m)
    a = (rng.random_sample((b, n, 3)) - 0.5).astype(np.float32)
    c = (rng.random_sample((b, m, 3)) - 0.5).astype(np.float32)
    om = orc.approx_match(a, c)
    ocost = orc.match_cost(a, c, om)
    o1, o2 = orc.match_cost_grad(a, c, om)
    cost = _raw.earth_mover(cu(a), cu(c))
    assert_rel(cost.cpu().numpy(), ocost, 1e-5, what="fused cost")
    cost_g, g1, g2 = _raw.earth_mover(cu(a), cu(c), with_grad=True)
    assert_rel(cost_g.cpu().numpy(), ocost, 1e-5, what="fused cost (grad variant)")
    # The fused gradients are built on the GPU's own match entries (hardware exp2: each within
    # rel 1e-4 of the oracle's, the `match` tolerance above), not on the oracle's match as in
    # test_emd_random_shapes.  A row's entries sum to its mass (multiL for grad1 rows, multiR for
    # grad2 rows) and multiply unit vectors, so the inherited error bound is 1e-4 * mass.
    massL, massR = max(1, m // n), max(1, n // m)
    assert_rel(g1.cpu().numpy(), o1, 1e-4, 1e-4 * massL, what="fused grad1")
    assert_rel(g2.cpu().numpy(), o2, 1e-4, 1e-4 * massR, what="fused grad2")


def test_earth_mover_fused_equals_chain_full_size():
    """C4 size: fused op == this library's own approx_match -> match_cost -> match_cost_grad chain
    on all 32 samples (both compute match entries with the same fma chain; only summation order
    differs)."""
    from rfnet_amd import _raw
    from pc_distance.tf_approxmatch import approx_match, match_cost, match_cost_grad
    rng = np.random.RandomState(100)
    a = cu((rng.random_sample((32, 2048, 3)) - 0.5).astype(np.float32))
    c = cu((rng.random_sample((32, 2048, 3)) - 0.5).astype(np.float32))
    match = approx_match(a, c)
    cost = match_cost(a, c, match)
    r1, r2 = match_cost_grad(a, c, match)
    del match
    fcost, g1, g2 = _raw.earth_mover(a, c, with_grad=True)
    assert_rel(fcost.cpu().numpy(), cost.cpu().numpy(), 1e-5, what="cost")
    assert_rel(g1.cpu().numpy(), r1.cpu().numpy(), 1e-4, 1e-5, what="grad1")
    assert_rel(g2.cpu().numpy(), r2.cpu().numpy(), 1e-4, 1e-5, what="grad2")


def test_earth_mover_cost_autograd_matches_chain():
    from pc_distance.tf_approxmatch import approx_match, match_cost, earth_mover_cost
    rng = np.random.RandomState(3)
    a0 = (rng.random_sample((2, 512, 3)) - 0.5).astype(np.float32)
    c0 = (rng.random_sample((2, 512, 3)) - 0.5).astype(np.float32)
    w = cu(np.array([0.5, 2.0], np.float32))
    a, c = cu(a0).requires_grad_(), cu(c0).requires_grad_()
    (match_cost(a, c, approx_match(a, c)) * w).sum().backward()
    a2, c2 = cu(a0).requires_grad_(), cu(c0).requires_grad_()
    (earth_mover_cost(a2, c2) * w).sum().backward()
    assert_rel(a2.grad.cpu().numpy(), a.grad.cpu().numpy(), 1e-4, 1e-5)
    assert_rel(c2.grad.cpu().numpy(), c.grad.cpu().numpy(), 1e-4, 1e-5)


def test_earth_mover_abi_errors():
    from rfnet_amd import _lib
    import ctypes as C
    lib = _lib.lib
    x = torch.zeros(1, 300, 3, device="cuda")
    cost = torch.zeros(1, device="cuda")
    g = torch.zeros(1, 300, 3, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    need = lib.rf_earth_mover_workspace_bytes(1, 300, 300)
    ws = torch.zeros(need, dtype=torch.uint8, device="cuda")
    assert lib.rf_earth_mover(1, 300, 300, p(x), p(x), p(cost), None, None, p(ws), need - 4, None) == -2
    assert lib.rf_earth_mover(1, 300, 300, p(x), p(x), p(cost), p(g), None, p(ws), need, None) == -1
    assert lib.rf_earth_mover(-1, 300, 300, p(x), p(x), p(cost), None, None, p(ws), need, None) == -1
    assert lib.rf_earth_mover(0, 300, 300, None, None, None, None, None, None, 0, None) == 0


@pytest.mark.parametrize("kind", ["box", "corners", "same", "near"])
@pytest.mark.parametrize("b,n,m", [(2, 1500, 2500), (1, 3000, 1025), (1, 600, 5121)])
def test_earth_mover_cost_by_column_classes(kind, b, n, m):
    """The cost-only form walks set 2's columns by the CLASS of their last live level (emd_class_count / emd_pack_cols_sorted /
    emd_fused_cls kernels): clouds that put every column in one class -- two clusters in opposite corners (nothing matches before the
    broad levels: all columns live to the end), a cloud against itself (everything is used up at the sharpest level), a jittered copy --
    and ragged sizes around the pack's 1024-column workgroups, against this library's own chain approx_match -> match_cost."""
    from rfnet_amd import _raw
    from pc_distance.tf_approxmatch import approx_match, match_cost
    rng = np.random.RandomState(n + m)
    if kind == "box":
        a, c = rng.random_sample((b, n, 3)) - 0.5, rng.random_sample((b, m, 3)) - 0.5
    elif kind == "corners":
        a, c = rng.random_sample((b, n, 3)) * 0.1 - 0.5, rng.random_sample((b, m, 3)) * 0.1 + 0.4
    else:
        base = rng.random_sample((b, max(n, m), 3)) - 0.5
        a, c = base[:, :n].copy(), base[:, :m].copy()
        if kind == "near":
            c += rng.normal(0, 2e-3, c.shape)
    ta, tc = cu(a.astype(np.float32)), cu(c.astype(np.float32))
    fused = _raw.earth_mover(ta, tc).cpu().numpy()
    chain = match_cost(ta, tc, approx_match(ta, tc)).cpu().numpy()
    assert np.isfinite(fused).all()
    assert_rel(fused, chain, 1e-5, 1e-6, what=f"cost-only earth_mover vs chain ({kind} {b}x{n}x{m})")
    swept = _raw.earth_mover(ta, tc, mode="swept").cpu().numpy()
    assert_rel(swept, chain, 1e-5, 1e-6, what=f"swept route ({kind})")


def test_earth_mover_fused_eval_size_16384():
    """The evaluation-size EMD (16384 vs 16384, recon_test.py / vv_recon.py:28 EVAL_SIZE micro-batches):
    the reference needs 1 GiB of match per sample; the fused op none.  Checked against this
    library's own materialising chain on one sample, and for batch independence on four."""
    from rfnet_amd import _raw
    from pc_distance.tf_approxmatch import approx_match, match_cost
    rng = np.random.RandomState(11)
    a = cu((rng.random_sample((4, 16384, 3)) - 0.5).astype(np.float32))
    c = cu((rng.random_sample((4, 16384, 3)) - 0.5).astype(np.float32))
    fused = _raw.earth_mover(a, c).cpu().numpy()
    assert np.isfinite(fused).all() and (fused > 0).all()
    match = approx_match(a[1:2], c[1:2])
    chain = match_cost(a[1:2], c[1:2], match).cpu().numpy()
    del match
    assert_rel(fused[1:2], chain, 1e-5, what="fused vs chain at 16384^2")
    solo = _raw.earth_mover(a[2:3].contiguous(), c[2:3].contiguous()).cpu().numpy()
    assert_rel(fused[2:3], solo, 1e-6, what="batch independence")


def test_exp2_is_exactly_zero_below_the_cull_argument():
    """What the skipping sweeps of the sharp levels rest on (approxmatch.hip kSkipArg): v_exp_f32(x) == +0 for every
    x <= -160, bit for bit, so a skipped pair would have added fma(0, s, acc) = acc."""
    from rfnet_amd._lib import check, lib
    x = np.concatenate([-np.linspace(160.0, 400.0, 200001), -np.logspace(np.log10(160.0), 30.0, 20000),
                        [-160.0, -np.inf, -3.0e38]]).astype(np.float32)
    tx = cu(x)
    ty = torch.empty_like(tx)
    check(lib.rf_probe_exp2(tx.data_ptr(), ty.data_ptr(), tx.numel(), None), "rf_probe_exp2")
    y = ty.cpu().numpy()
    assert (y.view(np.uint32) == 0).all(), "v_exp_f32 returned a non-zero (or -0) below -160"
    # and it is an exponential above: spot values
    v = cu(np.array([0.0, -1.0, -10.0, -126.0, 3.0], np.float32))
    o = torch.empty_like(v)
    check(lib.rf_probe_exp2(v.data_ptr(), o.data_ptr(), v.numel(), None), "rf_probe_exp2")
    assert np.allclose(o.cpu().numpy(), np.exp2(np.array([0.0, -1.0, -10.0, -126.0, 3.0])), rtol=1e-6)


@pytest.mark.parametrize("b,n,m", [(1, 4096, 4096), (1, 4096, 5000), (2, 6000, 4100)])
def test_earth_mover_culled_sharp_levels(orc, b, n, m):
    """rf_earth_mover on clouds of >= 4096 points: cost against the oracle's chain (approx_match -> match_cost, tf_approxmatch.cu
    restated) within 1e-5, with and without gradients, and the unfused chain.  (The name is rounds 3-5's, when the cost-only form ran
    its three sharpest levels CULLED over sorted copies, sums in sorted order; round 6 took that route out -- a soak found its cost
    1e-5 .. 3e-5 off the oracle's on rare large shapes: tools/experiments/emd_cull_route.patch.txt.)"""
    from rfnet_amd import _raw as R
    rng = np.random.RandomState(n + m)
    a = (rng.random_sample((b, n, 3)) - 0.5).astype(np.float32)
    c = (rng.random_sample((b, m, 3)) - 0.5).astype(np.float32)
    om = orc.approx_match(a, c)
    oc = orc.match_cost(a, c, om)
    assert_rel(R.earth_mover(cu(a), cu(c)).cpu().numpy(), oc, 1e-5, what="fused earth_mover cost")
    cost, g1, g2 = R.earth_mover(cu(a), cu(c), with_grad=True)
    assert_rel(cost.cpu().numpy(), oc, 1e-5, what="cost (with_grad)")
    o1, o2 = orc.match_cost_grad(a, c, om)
    multiL, multiR = (1.0, float(n // m)) if n >= m else (float(m // n), 1.0)
    # (at these sizes -- twice to three times test_earth_mover_fused's -- the
    # fast-exp noise of single match entries reaches 1.2e-4 of a unit mass in one component of 36000)
    assert_rel(g1.cpu().numpy(), o1, 1e-4, 2e-4 * multiL, what="grad1")
    assert_rel(g2.cpu().numpy(), o2, 1e-4, 2e-4 * multiR, what="grad2")
    # the dense pipeline on the same clouds gives the same cost to fp32 summation noise
    match = R.approx_match(cu(a), cu(c))
    assert_rel(R.match_cost(cu(a), cu(c), match).cpu().numpy(), oc, 1e-5, what="dense chain cost")


def test_sharp_level_sweeps_with_sorted_rows_keep_every_bit():
    """From 6e7 pairs per sweep on, the dense sweeps of the sharp levels take their rows in the clouds' spatial order and
    skip, after the distance, the columns whose weights are exactly 0 for every row of the wave (approxmatch.hip
    am_rowk_kernel SKIP).  Row order enters no sum and column order is untouched, so `match` must not change by one bit: the
    same clouds as a batch of one (4.2e6 pairs: plain sweeps) and replicated into a batch of sixteen (6.7e7: sorted rows) --
    also with a ragged pair of sizes, and for the fused earth_mover cost and gradients.  (What a sample's later levels sum over --
    the live columns, am_compact_kernel -- is decided by the clouds' sizes alone, never by the batch.)"""
    from rfnet_amd import _raw as R
    rng = np.random.RandomState(77)
    for n, m in ((2048, 2048), (3000, 1400)):
        reps = -(-60000000 // (n * m)) + 1
        a = (rng.random_sample((1, n, 3)) - 0.5).astype(np.float32)
        c = (rng.random_sample((1, m, 3)) - 0.5).astype(np.float32)
        one = R.approx_match(cu(a), cu(c))
        many = R.approx_match(cu(np.repeat(a, reps, 0)), cu(np.repeat(c, reps, 0)))
        for i in range(reps):
            assert torch.equal(many[i], one[0]), (n, m, i)
        c1, g1, h1 = R.earth_mover(cu(a), cu(c), with_grad=True)
        cm, gm, hm = R.earth_mover(cu(np.repeat(a, reps, 0)), cu(np.repeat(c, reps, 0)), with_grad=True)
        assert torch.equal(cm, c1.expand(reps)), (n, m)
        # (the gradients are atomic sums over workgroups: equal to fp32 summation noise, not bit for bit)
        assert torch.allclose(gm[0], g1[0], rtol=1e-5, atol=1e-6) and torch.allclose(hm[-1], h1[0], rtol=1e-5, atol=1e-6)
        # DIFFERENT clouds per sample (level 0's sweeps list, per wave and sample, the columns level 1's will visit -- am_rowk_kernel
        # MASK: a list read for the wrong sample would go unnoticed in a batch of copies), clustered so that the lists differ a lot
        A = (rng.random_sample((reps, n, 3)) - 0.5).astype(np.float32) * np.linspace(0.2, 1.0, reps, dtype=np.float32)[:, None, None]
        C = (rng.random_sample((reps, m, 3)) - 0.5).astype(np.float32) * np.linspace(1.0, 0.3, reps, dtype=np.float32)[:, None, None]
        many = R.approx_match(cu(A), cu(C))
        for i in range(reps):
            assert torch.equal(many[i], R.approx_match(cu(A[i:i + 1]), cu(C[i:i + 1]))[0]), (n, m, i)


@pytest.mark.parametrize("scale,what", [(1.0, "unit cube"), (4.0, "clouds four times the unit cube"), (float("nan"), "a NaN coordinate in one sample")])
def test_broad_levels_by_expansion_and_its_refusal(orc, scale, what):
    """A large batch (8.4e7 pairs: sorted rows at the sharp levels, live-column sweeps from the third level on) against the oracle's
    chain on its first and last sample; with clouds four times the unit cube (every level sixteen-fold sharper: the strays are the
    sharp levels' cancellations, 2485 of 8e6 on every build since round 4); with a NaN in ANOTHER sample of the batch (that sample is
    garbage by contract, the others must be what they are without the NaN next door).  (The name is round 5's, when the three
    broadest levels of such a batch came from a Taylor expansion that refused the last two cases: tools/experiments/emd_fgt_route.patch.txt.)"""
    from pc_distance.tf_approxmatch import approx_match, match_cost
    rng = np.random.RandomState(31)
    B, N = 21, 2000  # 8.4e7 pairs: past the expansion's threshold (approxmatch.hip FGT_MIN_PAIRS = 8e7)
    a = (rng.random_sample((B, N, 3)) - 0.5).astype(np.float32)
    c = (rng.random_sample((B, N, 3)) - 0.5).astype(np.float32)
    pick = [0, B - 1]
    if np.isnan(scale):
        a2 = a.copy()
        a2[1, 7, 2] = np.nan  # sample 1 is garbage by contract; the others must be what they are without the NaN next door
        got = approx_match(cu(a2), cu(c))[pick].cpu().numpy()
        om = orc.approx_match(a[pick], c[pick])
        bad = np.abs(got - om) > 1e-6 + 1e-4 * np.abs(om)
        assert int(bad.sum()) <= 32 and np.abs(got - om).max() < 2e-4, int(bad.sum())
        return
    a, c = a * np.float32(scale), c * np.float32(scale)
    om = orc.approx_match(a[pick], c[pick])
    got_all = approx_match(cu(a), cu(c))
    got = got_all[pick].cpu().numpy()
    strict_bar_report(f"2000^2 x {scale} ({what})", got, om)
    bad = np.abs(got - om) > 1e-6 + 1e-4 * np.abs(om)
    assert int(bad.sum()) <= (32 if scale == 1.0 else 2800), f"{int(bad.sum())} of {bad.size} entries outside abs 1e-6 + rel 1e-4 ({what})"
    assert_rel(match_cost(cu(a), cu(c), got_all).cpu().numpy()[pick], orc.match_cost(a[pick], c[pick], om), 1e-5, what="cost")


@pytest.mark.parametrize("b,n,m", [(2, 768, 37), (3, 1028, 100), (1, 2048, 17), (2, 1536, 1003), (1, 4096, 2500), (2, 1024, 16),
                                   (1, 1800, 4100), (2, 2048, 2048), (1, 3000, 64)])
def test_match_cost_grad_whole_row_form_shapes(orc, b, n, m):
    """match_cost_grad's whole-row form (approxmatch.hip mcg_rows_kernel: n % 4 == 0 and at least 3/4 of its 1024-k blocks alive,
    m >= 16, 16-byte aligned operands; 3 x 1028 x 100 is a shape that stays with the tile form) over ragged shapes -- dead lanes behind n, a partial last block of rows, l-ranges cut for LDS, one and several
    k-blocks -- with an arbitrary non-negative `match` (the op is linear in it: nothing needs it to be a transport plan),
    against the oracle; and the same call through views whose storage is NOT 16-byte aligned (the tile form takes those):
    same gradients to summation noise."""
    from rfnet_amd import _raw as R
    rng = np.random.RandomState(b * 1000 + n + m)
    a = (rng.random_sample((b, n, 3)) - 0.5).astype(np.float32)
    c = (rng.random_sample((b, m, 3)) - 0.5).astype(np.float32)
    mt = (rng.random_sample((b, m, n)) ** 8).astype(np.float32) / np.float32(n)
    c[0, 3] = a[0, 5]  # a coincident pair: rsq(max(d2, 1e-20)) * 0
    g1, g2 = R.match_cost_grad(cu(a), cu(c), cu(mt))
    o1, o2 = orc.match_cost_grad(a, c, mt)
    assert_rel(g1.cpu().numpy(), o1, 1e-4, 2e-6, what="grad1")
    assert_rel(g2.cpu().numpy(), o2, 1e-4, 2e-6, what="grad2")
    # misaligned storage: one float into a larger buffer
    buf = torch.empty(mt.size + 1, dtype=torch.float32, device="cuda")
    view = buf[1:].view(b, m, n)
    view.copy_(cu(mt))
    assert view.data_ptr() % 16 != 0
    h1, h2 = R.match_cost_grad(cu(a), cu(c), view)
    assert torch.allclose(h1, g1, rtol=1e-4, atol=2e-6) and torch.allclose(h2, g2, rtol=1e-4, atol=2e-6)


def test_broad_levels_by_expansion_ragged_sizes(orc):
    """A large batch with clouds of different sizes (multiL / multiR != 1; the live sets of the two sides' sweeps of different
    capacities, last chunks that are mostly padding): 24 x 2400 x 1500 = 8.6e7 pairs, first and last sample against the
    oracle's chain; the fused earth_mover cost on the same route."""
    from pc_distance.tf_approxmatch import approx_match, match_cost
    from rfnet_amd import _raw as R
    rng = np.random.RandomState(47)
    B, N, M = 24, 2400, 1500
    a = (rng.random_sample((B, N, 3)) - 0.5).astype(np.float32)
    c = (rng.random_sample((B, M, 3)) - 0.5).astype(np.float32)
    pick = [0, B - 1]
    om = orc.approx_match(a[pick], c[pick])
    got_all = approx_match(cu(a), cu(c))
    got = got_all[pick].cpu().numpy()
    strict_bar_report("2400 x 1500", got, om)
    bad = np.abs(got - om) > 1e-6 + 1e-4 * np.abs(om)
    assert int(bad.sum()) <= 32 and np.abs(got - om).max() < 2e-4, int(bad.sum())
    oc = orc.match_cost(a[pick], c[pick], om)
    assert_rel(match_cost(cu(a), cu(c), got_all).cpu().numpy()[pick], oc, 1e-5, what="cost")
    assert_rel(R.earth_mover(cu(a), cu(c)).cpu().numpy()[pick], oc, 1e-5, what="fused cost")


# ----------------------------------------------------------------------------- the pinned route (rf_approxmatch_mode / rf_earth_mover_mode)
@pytest.mark.parametrize("B,N", [(21, 2000), (32, 2048)])
def test_swept_route_is_batch_invariant(B, N):
    """The reference's kernel loops over the samples independently (tf_approxmatch.cu:13): sample i's match does not depend on the
    batch it is called in.  RF_EMD_AUTO picks routes and launch shapes by the size of the whole batch (both sizes here are past the
    sorted-row and expansion thresholds, a batch of one is not); RF_EMD_SWEPT pins them:
        approx_match(a, c)[i] == approx_match(a[i:i+1], c[i:i+1])   bit for bit
    for every i, and for slices of other lengths (the shards of an R-way split)."""
    from rfnet_amd import _raw as R
    rng = np.random.RandomState(B + N)
    a = (rng.random_sample((B, N, 3)) - 0.5).astype(np.float32)
    c = (rng.random_sample((B, N, 3)) - 0.5).astype(np.float32)
    ta, tc = cu(a), cu(c)
    full = R.approx_match(ta, tc, mode="swept")
    for i in range(B):
        one = R.approx_match(ta[i:i + 1], tc[i:i + 1], mode="swept")
        assert torch.equal(one[0], full[i]), f"sample {i} alone differs from the batch"
        del one
    for lo, hi in ((0, B // 2), (B // 2, B), (3, 8)):
        part = R.approx_match(ta[lo:hi], tc[lo:hi], mode="swept")
        assert torch.equal(part, full[lo:hi]), (lo, hi)
        del part
    # ... and the pinned route is the op: same bars as the default route against it (both sit within tolerance of the oracle)
    auto = R.approx_match(ta, tc)
    d = (auto - full).abs()
    assert float(d.max()) < 2e-4 and int((d > 1e-6 + 1e-4 * full.abs()).sum()) <= 64 * B
    assert_rel(R.match_cost(ta, tc, auto).cpu().numpy(), R.match_cost(ta, tc, full).cpu().numpy(), 1e-5, what="cost: auto vs swept")


def test_swept_earth_mover_is_identical_over_shard_splits():
    """SURVEY 8(d) C5 "cross-GPU loss equality": the per-sample EMD of B = 32 samples computed as R in {1, 2, 4, 8} contiguous shards (run
    one after the other on this GPU, as R ranks would each run theirs) -- identical vectors, bit for bit, on the pinned route;
    rfnet_amd.shard.emd_per_sample is that route."""
    from rfnet_amd import _raw as R
    from rfnet_amd import shard
    rng = np.random.RandomState(5)
    for n, m in ((2048, 2048), (1024, 1024), (700, 1100)):
        B = 32
        a = cu((rng.random_sample((B, n, 3)) - 0.5).astype(np.float32))
        c = cu((rng.random_sample((B, m, 3)) - 0.5).astype(np.float32))
        whole = R.earth_mover(a, c, mode="swept")
        for parts in (2, 4, 8):
            pieces = []
            for r in range(parts):
                lo, hi = shard.shard_bounds(B, r, parts)
                pieces.append(R.earth_mover(a[lo:hi], c[lo:hi], mode="swept"))
            assert torch.equal(torch.cat(pieces), whole), (n, m, parts)
        assert torch.equal(shard.emd_per_sample(a, c), whole / float(n))
        halves = torch.cat([shard.emd_per_sample(a[:16], c[:16]), shard.emd_per_sample(a[16:], c[16:])])
        assert torch.equal(halves, whole / float(n))
        assert_rel(R.earth_mover(a, c).cpu().numpy(), whole.cpu().numpy(), 1e-5, what="auto vs swept cost")


def test_emd_mode_abi_errors():
    from rfnet_amd import _raw as R
    from rfnet_amd._lib import lib
    a = cu(np.zeros((1, 8, 3), np.float32))
    with pytest.raises(ValueError):
        R.approx_match(a, a, mode="nope")
    assert lib.rf_approxmatch_mode_workspace_bytes(1, 600, 600, 0, 7) == 0
    m = torch.empty(1, 8, 8, device="cuda")
    ws = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    assert lib.rf_approxmatch_mode(1, 8, 8, a.data_ptr(), a.data_ptr(), m.data_ptr(), None, 0, ws.data_ptr(), ws.numel(), None, 7) != 0
    assert lib.rf_earth_mover_mode(1, 8, 8, a.data_ptr(), a.data_ptr(), m.data_ptr(), None, None, ws.data_ptr(), ws.numel(), None, -1) != 0


@pytest.mark.parametrize("kind", ["opposite_corners", "corner_vs_filled", "filled_vs_corner", "filled"])
def test_live_column_sweeps_on_lopsided_clouds(orc, kind):
    """From the third level on the sweeps run over the LIVE columns and rows of set 2 only (remainR exactly +0 drops a column out of
    every later sum: am_compact_kernel, am_p2_live_kernel).  How many are live depends on the clouds: on two that fill the same
    box nearly all columns are used up by the sharp levels (C4: 3 % left at level -1); two clusters in opposite corners match
    nothing until the broad levels, so every column stays live to the end; a partial shape against a complete one sits in
    between.  All of them against the oracle at the C4 bars, the cost at rel 1e-5, the fused cost, and against the pinned swept
    route (which sweeps every column at every level)."""
    from rfnet_amd import _raw as R
    rng = np.random.RandomState(11)
    B, N = 2, 1536
    filled = lambda: (rng.random_sample((B, N, 3)) - 0.5).astype(np.float32)
    corner = lambda s: np.clip(s * 0.45 + 0.03 * rng.randn(B, N, 3), -0.5, 0.5).astype(np.float32)
    a, c = {"opposite_corners": (corner(1.0), corner(-1.0)), "corner_vs_filled": (corner(1.0), filled()),
            "filled_vs_corner": (filled(), corner(-1.0)), "filled": (filled(), filled())}[kind]
    om = orc.approx_match(a, c)
    oc = orc.match_cost(a, c, om)
    got = R.approx_match(cu(a), cu(c))
    gm = got.cpu().numpy()
    strict_bar_report(f"live-column sweeps, {kind}", gm, om)
    bad = np.abs(gm - om) > 1e-6 + 1e-4 * np.abs(om)
    assert int(bad.sum()) <= 32 and np.abs(gm - om).max() < 2e-4, f"{kind}: {int(bad.sum())} strays, max {np.abs(gm - om).max():.2e}"
    # marginals: all but a few within 1e-5, none beyond the 2e-4 a clamp flip moves (tests/test_oracle_golden.py::test_match_bar_is_ill_conditioned)
    for ax in (1, 2):
        dm = np.abs(gm.sum(ax) - om.sum(ax))
        assert int((dm > 1e-5 + 1e-5 * np.abs(om.sum(ax))).sum()) <= 8 and dm.max() < 2e-4, (kind, ax, float(dm.max()))
    assert_rel(R.match_cost(cu(a), cu(c), got).cpu().numpy(), oc, 1e-5, what=f"{kind}: cost")
    assert_rel(R.earth_mover(cu(a), cu(c)).cpu().numpy(), oc, 1e-5, what=f"{kind}: fused cost")
    swept = R.approx_match(cu(a), cu(c), mode="swept").cpu().numpy()
    sb = np.abs(swept - om) > 1e-6 + 1e-4 * np.abs(om)
    # leaving the dead columns out must not make the route the less faithful of the two by more than a handful of clamp flips
    assert int(bad.sum()) <= int(sb.sum()) + 16, (int(bad.sum()), int(sb.sum()))


@pytest.mark.parametrize("n,m", [(512, 512), (513, 700), (2048, 640), (640, 2048), (5000, 3000)])
def test_live_column_sweeps_shapes(orc, n, m):
    """The live-column route at the edges of its domain (it starts at 512 points per cloud): ragged sizes, multiL / multiR != 1,
    a set 2 larger or smaller than set 1, more entries than one thread of the packing kernels holds in registers."""
    from rfnet_amd import _raw as R
    rng = np.random.RandomState(n + m)
    b = 2
    a = (rng.random_sample((b, n, 3)) - 0.5).astype(np.float32)
    c = (rng.random_sample((b, m, 3)) - 0.5).astype(np.float32)
    om = orc.approx_match(a[:1], c[:1])
    got = R.approx_match(cu(a), cu(c))
    gm = got[:1].cpu().numpy()
    bad = np.abs(gm - om) > 1e-6 + 1e-4 * np.abs(om)
    # (2048 x 640 is one of the ill-conditioned cases: one clamp flip at a sharp level moves 6e-4 of a unit mass and leaves 329 entries
    # outside the strict bar -- on the pinned swept route exactly as here: tools/experiments/emd_ragged_strays.py.  So the bar is the
    # swept route's own count, the fuzz tests' 2e-3 of a unit mass, and the cost.)
    swept = R.approx_match(cu(a), cu(c), mode="swept")[:1].cpu().numpy()
    sb = np.abs(swept - om) > 1e-6 + 1e-4 * np.abs(om)
    assert int(bad.sum()) <= max(32, int(sb.sum()) + 16) and np.abs(gm - om).max() < 2e-3, (int(bad.sum()), int(sb.sum()), float(np.abs(gm - om).max()))
    oc = orc.match_cost(a[:1], c[:1], om)
    assert_rel(R.match_cost(cu(a), cu(c), got).cpu().numpy()[:1], oc, 1e-5, what="cost")
    assert_rel(R.earth_mover(cu(a), cu(c)).cpu().numpy()[:1], oc, 1e-5, what="fused cost")
    # the fused gradients against MatchCostGrad on this route's own match (the oracle's match is a clamp flip away in the ill-conditioned case)
    cg, g1, g2 = R.earth_mover(cu(a), cu(c), with_grad=True)
    o1, o2 = R.match_cost_grad(cu(a), cu(c), got)
    assert_rel(g1.cpu().numpy(), o1.cpu().numpy(), 1e-4, 1e-4, what="fused grad1")
    assert_rel(g2.cpu().numpy(), o2.cpu().numpy(), 1e-4, 1e-4, what="fused grad2")
