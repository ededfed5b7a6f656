"""GPU: the loss/geometry glue (first 'next' row) against numpy compositions of
oracle outputs, with the reference's formulas (vv_recon.py:67-83,132-139,171-193,381-419)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def cu(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def _chamfer_np(orc, a, c):
    d1, i1, d2, _ = orc.nn_distance(a, c)
    return (np.sqrt(d1).mean(dtype=np.float64) + np.sqrt(d2).mean(dtype=np.float64)) / 2, i1


def test_chamfer_fidelity_earth_mover(orc):
    from rfnet_amd import glue
    rng = np.random.RandomState(0)
    a = (rng.rand(3, 512, 3) - 0.5).astype(np.float32)
    c = (rng.rand(3, 512, 3) - 0.5).astype(np.float32)
    loss, idx1 = glue.chamfer_big(cu(a), cu(c))
    exp, ei = _chamfer_np(orc, a, c)
    assert abs(float(loss) - exp) < 1e-6 * exp + 1e-7 and np.array_equal(idx1.cpu().numpy(), ei)
    d1 = orc.nn_distance(a, c)[0]
    assert abs(float(glue.fidelity_loss(cu(a), cu(c))) - np.sqrt(d1).mean(dtype=np.float64)) < 1e-6
    om = orc.approx_match(a, c)
    exp_emd = (orc.match_cost(a, c, om) / 512.0).mean()
    assert abs(float(glue.earth_mover(cu(a), cu(c))) - exp_emd) < 1e-5 * exp_emd


def test_merge_layer_and_sampling_and_rechamfer(orc):
    from rfnet_amd import glue
    rng = np.random.RandomState(1)
    raw = rng.rand(2, 3000, 3).astype(np.float32)
    idx, new = glue.sampling(64, cu(raw))
    oi = orc.farthest_point_sample(64, raw)
    assert np.array_equal(idx.cpu().numpy(), oi)
    assert np.array_equal(new.cpu().numpy(), orc.gather_point(raw, oi))
    newpts = (new.cpu().numpy() + 0.05 * rng.randn(2, 64, 3)).astype(np.float32)
    out = glue.merge_layer(cu(raw), cu(newpts), 0.1).cpu().numpy()
    i2 = orc.nn_distance(raw, newpts)[3]
    g = np.take_along_axis(raw, i2[..., None].astype(np.int64), 1)
    diff = (g - newpts).astype(np.float32)
    ratio = np.exp(-(diff * diff).sum(-1, keepdims=True) / np.float32(1e-8 + 0.1 * 0.1))
    assert np.allclose(out, newpts + ratio * diff, rtol=1e-5, atol=1e-6)
    # gradient flows to newpts and (through group_point) to rawpts
    tr, tn = cu(raw).requires_grad_(True), cu(newpts).requires_grad_(True)
    glue.merge_layer(tr, tn, 0.1).sum().backward()
    assert tr.grad is not None and tn.grad is not None and float(tr.grad.abs().sum()) > 0
    # random-subset sampling shares one index set across the batch
    ridx, rxyz = glue.sampling(10, cu(raw), use_type='r')
    assert torch.equal(ridx[0], ridx[1]) and torch.equal(rxyz, cu(raw)[:, ridx[0].long()])
    gt = rng.rand(2, 1024, 3).astype(np.float32)
    pred = (gt + 0.01 * rng.randn(2, 1024, 3)).astype(np.float32)
    exp = np.mean([_chamfer_np(orc, pred[:, i * 128:(i + 1) * 128].copy(), gt[:, i * 128:(i + 1) * 128].copy())[0]
                   for i in range(8)])
    assert abs(float(glue.re_chamfer(cu(gt), cu(pred))) - exp) < 1e-6
    cens = rng.rand(2, 64, 3).astype(np.float32)
    outmat = (0.1 * rng.randn(2, 64, 16, 3)).astype(np.float32)
    d2 = orc.nn_distance(cens, raw)[2]
    exp = max(0.0, float((outmat ** 2).sum(-1).mean()) - 0.4 * float(d2.mean(dtype=np.float64)))
    assert abs(float(glue.zero_groupnear(cu(cens), cu(raw), cu(outmat))) - exp) < 1e-6


def test_chamfer_loss_training_step_gradient(orc):
    """chamfer_big.backward(): d/dxyz of mean sqrt(dist) chains 0.5/sqrt(d) into NnDistanceGrad."""
    from rfnet_amd import glue
    rng = np.random.RandomState(2)
    a = rng.rand(2, 300, 3).astype(np.float32)
    c = rng.rand(2, 400, 3).astype(np.float32)
    ta, tc = cu(a).requires_grad_(True), cu(c).requires_grad_(True)
    glue.chamfer_big(ta, tc)[0].backward()
    d1, i1, d2, i2 = orc.nn_distance(a, c)
    gd1 = (0.5 / np.sqrt(d1) / d1.size / 2).astype(np.float32)
    gd2 = (0.5 / np.sqrt(d2) / d2.size / 2).astype(np.float32)
    g1, g2 = orc.nn_distance_grad(a, c, gd1, i1, gd2, i2)
    assert np.allclose(ta.grad.cpu().numpy(), g1, rtol=1e-4, atol=1e-7 * np.abs(g1).max() + 1e-9)
    assert np.allclose(tc.grad.cpu().numpy(), g2, rtol=1e-4, atol=1e-4 * np.abs(g2).max())
