"""The RFNet generator graph (second 'next' row).

CPU: the parameter inventory of rfnet_amd.rfnet.RFNet equals, name for name and shape for shape,
the variable list of the reference's checkpoint index (tests/golden/rfnet_variables.json, produced
from /root/reference/bestrecord/model-229999.index by tools/read_tf_index.py; data only).
GPU: one forward/backward on the HIP ops -- shapes, finiteness, gradient reach.  Numerical parity
with the TF graph is not claimed (no TensorFlow, no weight blob: SURVEY.md T10)."""
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
