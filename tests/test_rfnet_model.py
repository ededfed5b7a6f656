"""The RFNet generator graph (second 'next' row).

CPU: the parameter inventory of rfnet_amd.rfnet.RFNet equals, name for name and shape for shape,
the variable list of the reference's checkpoint index (tests/golden/rfnet_variables.json, produced
from /root/reference/bestrecord/model-229999.index by tools/read_tf_index.py; data only).
CPU: the float64 restatement of the graph (oracle/rfnet_oracle.py) consumes exactly that inventory.
GPU: forward + training loss against the restatement oracle fed the same seeded weights (B=2, all
four outputs and every loss term, rel 1e-4, indices of FPS / merge_layer shared and their
agreement asserted); one forward/backward at C5's per-GPU size (B=32, 3000 -> 16384) on the HIP
ops -- shapes, finiteness, gradient reach, determinism.  An execution of the TF graph itself is
not available (no TensorFlow, no weight blob: SURVEY.md T2/T10)."""
import json
import os

import numpy as np
import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_parameter_inventory_matches_reference_checkpoint_index():
    from rfnet_amd.rfnet import RFNet
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "rfnet_variables.json")))
    # `subvar*` belong to the training script's loss section (vv_recon.py train()), not the graph
    ref = {k: v for k, v in ref.items() if not k.startswith("subvar")}
    mine = RFNet().tf_variables()
    assert sorted(mine) == sorted(ref)
    for k in ref:
        assert list(mine[k]) == list(ref[k]), k
    n_params = sum(int(np.prod(v)) for v in mine.values())
    assert n_params == sum(p.numel() for p in RFNet().parameters()) == 3827611


def test_sharing_quirk_shared_kernels_fresh_biases():
    from rfnet_amd.rfnet import RFNet
    net = RFNet()
    names = net.tf_variables()
    assert "cell/state0/weights" in names and "cell_1/state0/weights" not in names
    assert all(f"{s}/state0/Variable" in names for s in ("cell", "cell_1", "cell_2"))
    assert "decode_cell_1/points_out/Variable" in names and "decode_cell_1/points_out/weights" not in names


@pytest.mark.gpu
def test_forward_backward_on_hip_ops():
    from rfnet_amd import glue
    from rfnet_amd.rfnet import RFNet
    torch.manual_seed(0)
    net = RFNet().cuda()
    rng = np.random.RandomState(0)
    partial = torch.from_numpy((rng.rand(2, 3000, 3) - 0.5).astype(np.float32)).cuda()
    gt = torch.from_numpy((rng.rand(2, 16384, 3) - 0.5).astype(np.float32)).cuda()
    p1, p2, p3, pf = net(partial)
    assert tuple(p1.shape) == (2, 64, 3) and tuple(p2.shape) == (2, 1024, 3)
    assert tuple(p3.shape) == (2, 16384, 3) and tuple(pf.shape) == (2, 16384, 3)
    assert all(torch.isfinite(t).all() for t in (p1, p2, p3, pf))
    # the reference's main loss terms (vv_recon.py:484-492): CD at 16384^2 + EMD at 64^2 / 1024^2
    gt64 = glue.sampling(64, gt)[1]
    gt1024 = glue.sampling(1024, gt)[1]
    loss = glue.chamfer_big(pf, gt)[0] + glue.earth_mover(p1, gt64) + glue.earth_mover(p2, gt1024)
    loss.backward()
    assert torch.isfinite(loss)
    missing = [n for n, p in net.named_parameters() if p.grad is None]
    # full_process discards the state returned by the last refine layer, so nothing downstream of
    # the last point decoder's STATE path reaches the loss: the feat_refine branch of
    # refine_layer_final and the per-application biases of decode_cell's state layers (2nd call)
    def dead(n):
        return ("refine_layer_final__feat_refine" in n or
                (n.startswith("biases.decode_cell_1__state") and "state_trans" not in n))
    assert all(dead(n) for n in missing), [n for n in missing if not dead(n)]
    assert len(missing) == 6 + 2 + 32
    assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)
    # deterministic forward
    q = net(partial)
    assert all(torch.equal(a, b) for a, b in zip((p1, p2, p3, pf), q))


def _seeded_net(seed=0, bias_std=0.05):
    """Random-init weights as the reference initialises them, plus NON-zero biases and decline factors
    of a useful size, so that the bias plumbing (fresh biases per cell application) is under test."""
    from rfnet_amd.rfnet import RFNet
    torch.manual_seed(seed)
    net = RFNet()
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for p in net.biases.values():
            p.copy_(bias_std * torch.randn(p.shape, generator=g))
        for i, dn in enumerate(("decline_factor0", "decline_factor1", "decline_factor")):
            getattr(net, dn).fill_(0.05 + 0.03 * i)
    return net


def test_oracle_graph_consumes_the_reference_inventory(orc):
    """CPU: the float64 restatement runs on the product's tf_state_dict, touches every variable of the
    checkpoint inventory (except the ones full_process itself leaves dead) and produces the four
    outputs at the reference's sizes.  Small input (300 points): the graph is size-agnostic."""
    from oracle.rfnet_oracle import RFNetOracle
    net = _seeded_net()
    sd = net.tf_state_dict()
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "rfnet_variables.json")))
    assert sorted(sd) == sorted(k for k in ref if not k.startswith("subvar"))
    o = RFNetOracle(sd, orc)
    rng = np.random.RandomState(0)
    out = o.forward((rng.rand(1, 300, 3) - 0.5).astype(np.float32))
    assert out["points1"].shape == (1, 64, 3) and out["points2"].shape == (1, 1024, 3)
    assert out["points3"].shape == (1, 16384, 3) and out["points_final"].shape == (1, 16384, 3)
    assert all(np.isfinite(out[k]).all() for k in ("points1", "points2", "points3", "points_final"))
    unused = set(sd) - o.used
    # full_process never reads the state returned by the last refine layer / the last decode cell
    assert all(("refine_layer_final/feat_refine" in k) or k.startswith("decode_cell_1/state") for k in unused), unused


@pytest.mark.gpu
def test_forward_and_training_loss_match_the_restatement_oracle(orc):
    """Row f2 / C5 parity: RFNet on the HIP ops + fp32 library GEMMs vs the float64 restatement of
    full_process and of train()'s loss block, same weights, same inputs, B=2."""
    from oracle.rfnet_oracle import RFNetOracle
    from rfnet_amd.rfnet import training_loss
    net = _seeded_net().cuda()
    rng = np.random.RandomState(3)
    partial = (rng.rand(2, 3000, 3) - 0.5).astype(np.float32)
    gt = (rng.rand(2, 16384, 3) - 0.5).astype(np.float32)
    col, terms = {}, {}
    with torch.no_grad():
        outs = net(torch.from_numpy(partial).cuda(), collect=col)
        loss = training_loss(net, outs, col, torch.from_numpy(gt).cuda(), terms=terms)
    shared = {k: col[k].cpu().numpy() for k in ("fps32", "merge1", "merge2", "merge3")}
    o = RFNetOracle(net.tf_state_dict(), orc)
    ref = o.forward(partial, shared=shared)
    # the discrete choices: FPS runs on the input itself (identical), the merge layers' nearest
    # neighbours on fp32-vs-float64 network outputs (a few near-ties may flip)
    assert ref["agreement"]["fps32"] == 1.0
    for k in ("merge1", "merge2", "merge3"):
        assert ref["agreement"][k] > 0.995, ref["agreement"]

    def close(got, exp, what, rel=1e-4):
        got = got.detach().cpu().numpy().astype(np.float64)
        scale = np.abs(exp).max()
        err = np.abs(got - exp).max()
        assert err <= rel * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"

    for name, t in zip(("points1", "points2", "points3", "points_final"), outs):
        close(t, ref[name], name)
    close(col["points1"], ref["points1_pre"], "collection points1")
    close(col["points2"], ref["points2_pre"], "collection points2")
    close(col["refine_layer_final16384"], ref["refinemove3"], "refinemove3")
    close(col["decode_cell64"], ref["decode_move64"], "decode_cell64")
    close(col["decode_cell1024"], ref["decode_move1024"], "decode_cell1024")
    # the loss block on the ORACLE's own tensors, FPS indices of gt shared (same input -> identical anyway)
    sh = {"gt_fps64": terms["gt_fps64"].cpu().numpy(), "gt_fps1024": terms["gt_fps1024"].cpu().numpy()}
    assert np.array_equal(sh["gt_fps64"], orc.farthest_point_sample(64, gt))
    assert np.array_equal(sh["gt_fps1024"], orc.farthest_point_sample(1024, gt))
    rt = o.training_loss(ref, gt, shared=sh)
    for k in ("cd1", "cd2", "cd3", "cd4", "recd3", "moveloss", "loss_d1", "loss_d2", "loss_dec", "loss"):
        g, e = float(terms[k]), float(rt[k])
        assert abs(g - e) <= 2e-4 * abs(e) + 1e-7, f"{k}: {g} vs {e}"
    assert abs(float(loss) - rt["loss"]) <= 2e-4 * abs(rt["loss"])


@pytest.mark.gpu
def test_c5_size_forward_backward_properties():
    """BASELINE.json configs[4] at one GPU's share: B=32, 3000 -> 16384 points, the reference's full
    training loss, forward + backward.  Properties: shapes, finiteness, every live parameter gets a
    finite gradient, bit-identical forward on a second run, per-sample independence (sample 0 of
    the batch equals a B=1 run on it: the path shards by batch)."""
    from rfnet_amd.rfnet import training_loss
    net = _seeded_net().cuda()
    rng = np.random.RandomState(5)
    partial = torch.from_numpy((rng.rand(32, 3000, 3) - 0.5).astype(np.float32)).cuda()
    gt = torch.from_numpy((rng.rand(32, 16384, 3) - 0.5).astype(np.float32)).cuda()
    col = {}
    outs = net(partial, collect=col)
    assert [tuple(t.shape) for t in outs] == [(32, 64, 3), (32, 1024, 3), (32, 16384, 3), (32, 16384, 3)]
    loss = training_loss(net, outs, col, gt)
    loss.backward()
    assert torch.isfinite(loss) and all(torch.isfinite(t).all() for t in outs)
    missing = [n for n, p in net.named_parameters() if p.grad is None]
    def dead(n):
        return ("refine_layer_final__feat_refine" in n or
                (n.startswith("biases.decode_cell_1__state") and "state_trans" not in n))
    assert all(dead(n) for n in missing), [n for n in missing if not dead(n)]
    assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)
    # gt preparation on a side stream (overlapping the forward) == the reference's two separate FPS runs
    from rfnet_amd import glue
    from rfnet_amd.rfnet import GroundTruth
    g = GroundTruth(gt, 64, 1024, overlap=True).join()
    i64, p64 = glue.sampling(64, gt)
    i1024, p1024 = glue.sampling(1024, gt)
    assert torch.equal(g.idx1, i64) and torch.equal(g.idx2, i1024) and torch.equal(g.gt1, p64) and torch.equal(g.gt2, p1024)
    with torch.no_grad():
        l_overlap = training_loss(net, outs, col, gt, prepared=GroundTruth(gt, 64, 1024, overlap=True))
    assert abs(float(l_overlap) - float(loss.detach())) <= 1e-6 * abs(float(loss.detach()))
    with torch.no_grad():
        again = net(partial)
        one = net(partial[:1].contiguous())
    assert all(torch.equal(a, b) for a, b in zip(outs, again))
    for a, b in zip(outs, one):  # same sample alone: only the GEMM batch size differs
        assert torch.allclose(a[:1], b, rtol=1e-4, atol=1e-5)


def test_split_concat_layer_equals_the_concatenated_layer():
    """CPU: RFNet.dcat (row blocks of the kernel applied to the un-concatenated parts, broadcast parts
    as one row per sample) == the reference's tile + concat + conv2d, for the layer shapes it is
    used on."""
    from rfnet_amd.rfnet import RFNet
    net = _seeded_net()
    g = torch.Generator().manual_seed(3)
    B, N = 2, 37
    cases = [("cell", "state0", [(N, 3), (1, 256)], 1), ("recover2", "recover20", [(1, 256), (N, 3)], 0),
             ("", "ini_featout0", [(N, 3), (1, 256), (1, 256)], 0), ("init_cell", "state0", [(32, 16), (1, 256)], 0),
             ("refine_layer2", "feat_refine0", [(N, 3), (N, 128), (1, 256)], 0),
             ("decode_cell", "basic_state0", [(N, 256), (N, 128)], 1), ("decode_cell", "state0", [(N, 256), (1, 256)], 0)]
    for scope, name, shapes, call in cases:
        parts = [torch.randn(B, n, c, generator=g) for n, c in shapes]
        npts = max(n for n, _ in shapes)
        full = torch.cat([p.expand(-1, npts, -1) for p in parts], -1)
        ref = net.d(scope, name, full, call=call)
        got = net.dcat(scope, name, parts, call=call)
        assert torch.allclose(got, ref, rtol=1e-5, atol=1e-5), (scope, name)


@pytest.mark.gpu
def test_fused_layer_tails_match_tensor_ops():
    """rf_point_affine (one-pass act(y + p @ w + r)) and the GEMM+ReLU-epilogue layer against the same
    layers written with plain tensor ops: values and every gradient."""
    from rfnet_amd import _raw
    from rfnet_amd.rfnet import _PointAffine, linear_relu
    g = torch.Generator(device="cuda").manual_seed(0)
    B, N, C = 3, 1000, 128
    for act in ("relu", "tanh", None):
        for has_y, kp, per_sample in ((True, 3, True), (False, 3, True), (True, 0, False), (True, 16, True), (False, 16, False)):
            y = torch.randn(B, N, C, device="cuda", generator=g, requires_grad=True) if has_y else None
            p = torch.randn(B, N, kp, device="cuda", generator=g, requires_grad=True) if kp else None
            w = torch.randn(kp, C, device="cuda", generator=g, requires_grad=True) if kp else None
            r = torch.randn((B, 1, C) if per_sample else (C,), device="cuda", generator=g, requires_grad=True)
            out = _PointAffine.apply(y, p, w, r, act)
            ref = r + (y if has_y else 0) + (p @ w if kp else 0)
            ref = torch.relu(ref) if act == "relu" else torch.tanh(ref) if act == "tanh" else ref
            assert torch.allclose(out, ref, rtol=1e-5, atol=1e-5)
            wgt = torch.randn(B, N, C, device="cuda", generator=g)
            leaves = [t for t in (y, p, w, r) if t is not None]
            got = torch.autograd.grad((out * wgt).sum(), leaves)
            exp = torch.autograd.grad((ref * wgt).sum(), leaves)
            for a, e in zip(got, exp):
                assert torch.allclose(a, e, rtol=1e-4, atol=1e-4 * float(e.abs().max())), (act, has_y, kp)
    x = torch.randn(B, N, 259, device="cuda", generator=g, requires_grad=True)
    w = torch.randn(259, 256, device="cuda", generator=g, requires_grad=True)
    b = torch.randn(256, device="cuda", generator=g, requires_grad=True)
    out, ref = linear_relu(x, w, b), torch.relu(x @ w + b)
    assert torch.allclose(out, ref, rtol=1e-4, atol=1e-3)
    got = torch.autograd.grad(out.sum(), (x, w, b))
    exp = torch.autograd.grad(ref.sum(), (x, w, b))
    for a, e in zip(got, exp):
        assert torch.allclose(a, e, rtol=1e-3, atol=1e-3 * float(e.abs().max()))
    assert _raw.lib.rf_point_affine_supported(130, 3) == 0 and _raw.lib.rf_point_affine_supported(128, 17) == 0


@pytest.mark.gpu
def test_maxpool_points_kernel():
    from rfnet_amd import _raw
    g = torch.Generator(device="cuda").manual_seed(1)
    for (b, n, c) in ((1, 1, 4), (2, 255, 64), (3, 3000, 256), (2, 16384, 128), (1, 4024, 384), (2, 257, 1024)):
        x = torch.randn(b, n, c, device="cuda", generator=g)
        assert torch.equal(_raw.maxpool_points(x), x.amax(1, keepdim=True))


@pytest.mark.gpu
def test_backward_kernels_match_tensor_ops():
    """Training-step backward pieces: rf_act_grad_colsum, rf_maxpool_points_idx and the split weight
    gradient against plain tensor ops."""
    from rfnet_amd import _raw
    from rfnet_amd.rfnet import _LinearAct, _MaxPool, _splits, _wgrad, maxpool_points
    g = torch.Generator(device="cuda").manual_seed(2)
    for (b, n, c) in ((1, 1, 4), (2, 255, 64), (3, 3000, 256), (2, 16384, 128), (5, 1031, 1024)):
        grad = torch.randn(b, n, c, device="cuda", generator=g)
        out = torch.randn(b, n, c, device="cuda", generator=g)
        out[0, 0, 0] = 0.0  # relu / leaky: zero is NOT positive
        for act, ref in (("relu", torch.where(out > 0, grad, torch.zeros_like(grad))),
                         ("tanh", grad * (1.0 - out * out)),
                         ("leaky_relu", torch.where(out > 0, grad, grad * 0.2)),
                         (None, grad)):
            got, sums = _raw.act_grad_colsum(grad, out, act)
            assert torch.equal(got, ref) if act != "tanh" else torch.allclose(got, ref, rtol=1e-6, atol=1e-7), act
            exp = ref.double().sum(1)
            assert torch.allclose(sums.double(), exp, rtol=1e-5, atol=1e-5 * float(exp.abs().max()) + 1e-6), (act, b, n, c)
            again = _raw.act_grad_colsum(grad, out, act)[1]
            assert torch.equal(sums, again), "fixed summation order"
        # in place
        g2 = grad.clone()
        got, _ = _raw.act_grad_colsum(g2, out, "relu", inplace=True)
        assert got.data_ptr() == g2.data_ptr() and torch.equal(g2, torch.where(out > 0, grad, torch.zeros_like(grad)))
        # pooling with indices
        x = torch.randn(b, n, c, device="cuda", generator=g)
        if n > 3:
            x[:, 3] = x[:, 1]  # ties: the lower point index wins
        val, idx = _raw.maxpool_points_idx(x)
        assert torch.equal(val, x.amax(1, keepdim=True))
        first = (x == val).float().argmax(1)  # first position of the maximum
        assert torch.equal(idx.long(), first)
        xr = x.clone().requires_grad_(True)
        up = torch.randn(b, 1, c, device="cuda", generator=g)
        (gin,) = torch.autograd.grad((_MaxPool.apply(xr) * up).sum(), xr)
        exp = torch.zeros_like(x).scatter_(1, first.unsqueeze(1), up)
        assert torch.equal(gin, exp)
    assert maxpool_points(torch.randn(2, 50, 8, device="cuda", requires_grad=True)).grad_fn is not None
    # weight gradient cut into row blocks (with a remainder block)
    for rows, cin, cout in ((524288, 3, 256), (96000, 128, 64), (32 * 4024, 64, 48), (2049 * 5, 16, 128), (1000, 7, 5)):
        x = torch.randn(rows, cin, device="cuda", generator=g)
        gr = torch.randn(rows, cout, device="cuda", generator=g)
        exp = (x.double().t() @ gr.double())
        got = _wgrad(x, gr)
        assert torch.allclose(got.double(), exp, rtol=1e-4, atol=1e-4 * float(exp.abs().max())), (rows, cin, cout)
    assert _splits(524288) == 128 and _splits(96000) == 32 and _splits(2048) == 1
    # the non-ReLU dense layer
    for act in (None, "tanh", "leaky_relu"):
        for cout in (128, 3):
            x = torch.randn(4000, 64, device="cuda", generator=g, requires_grad=True)
            w = torch.randn(64, cout, device="cuda", generator=g, requires_grad=True)
            bias = torch.randn(cout, device="cuda", generator=g, requires_grad=True)
            out = _LinearAct.apply(x, w, bias, act)
            ref = x @ w + bias
            ref = torch.tanh(ref) if act == "tanh" else torch.nn.functional.leaky_relu(ref, 0.2) if act == "leaky_relu" else ref
            assert torch.allclose(out, ref, rtol=1e-4, atol=1e-4)
            up = torch.randn(4000, cout, device="cuda", generator=g)
            got = torch.autograd.grad((out * up).sum(), (x, w, bias))
            exp = torch.autograd.grad((ref * up).sum(), (x, w, bias))
            for a, e in zip(got, exp):
                assert torch.allclose(a, e, rtol=1e-3, atol=1e-3 * float(e.abs().max())), (act, cout)


@pytest.mark.gpu
def test_row_sparse_pool_backward_equals_the_dense_backward():
    """_PooledChain (backward of an MLP chain that feeds only a max-pool, recomputed on the arg-max rows)
    against the ordinary dense backward of the same graph: outputs, loss and every parameter gradient."""
    from rfnet_amd.rfnet import training_loss
    rng = np.random.RandomState(3)
    partial = torch.from_numpy((rng.rand(2, 3000, 3) - 0.5).astype(np.float32)).cuda()
    gt = torch.from_numpy((rng.rand(2, 16384, 3) - 0.5).astype(np.float32)).cuda()
    net = _seeded_net(4).cuda()
    res = {}
    for sparse in (True, False):
        net.sparse_pool_backward = sparse
        net.zero_grad(set_to_none=True)
        collect = {}
        outs = net(partial, collect=collect)
        loss = training_loss(net, outs, collect, gt, 0.01)
        loss.backward()
        res[sparse] = (outs, float(loss), {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None})
    net.sparse_pool_backward = True
    for a, b in zip(res[True][0], res[False][0]):
        assert torch.equal(a, b)  # the forward is the same code
    assert res[True][1] == res[False][1]
    assert set(res[True][2]) == set(res[False][2])
    worst = ("", 0.0)
    for n, gs in res[True][2].items():
        gd = res[False][2][n]
        err = float((gs - gd).abs().max()) / (float(gd.abs().max()) + 1e-12)
        worst = max(worst, (n, err), key=lambda t: t[1])
    assert worst[1] < 2e-3, worst
