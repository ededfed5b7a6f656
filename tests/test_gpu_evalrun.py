"""GPU: the evaluation driver (recon_test.py:19-100 on this stack) on a small synthetic data set, eager
and with the forward captured into a HIP graph: same rows bit for bit, the reference's CSV schema,
per-category means, timing that skips the first models."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make_dataset(root, rng, cats=("02691156", "03001627"), per_cat=6):
    ids = []
    for c in cats:
        os.makedirs(os.path.join(root, "partial", c), exist_ok=True)
        os.makedirs(os.path.join(root, "complete", c), exist_ok=True)
        for k in range(per_cat):
            name = "m%02d" % k
            from rfnet_amd import evalio
            complete = (rng.rand(16384, 3) - 0.5).astype(np.float32)
            npart = int(rng.randint(600, 3500))  # some scans shorter than 3000 points: resample_pcd duplicates
            partial = complete[rng.permutation(16384)[:npart]]
            evalio.save_pcd(os.path.join(root, "partial", c, name + ".pcd"), partial)
            evalio.save_pcd(os.path.join(root, "complete", c, name + ".pcd"), complete)
            ids.append(f"{c}/{name}")
    with open(os.path.join(root, "test.list"), "w") as f:
        f.write("\n".join(ids))
    return ids


def test_evaluate_eager_equals_graph(tmp_path):
    from rfnet_amd import evalio, evalrun
    from rfnet_amd.rfnet import RFNet
    rng = np.random.RandomState(0)
    ids = _make_dataset(str(tmp_path), rng)
    torch.manual_seed(0)
    net = RFNet().cuda().eval()
    out = {}
    for mode in (False, True):
        res = evalrun.evaluate(net, str(tmp_path / "test.list"), str(tmp_path), str(tmp_path / f"res{int(mode)}"),
                               graph=mode, rng=np.random.RandomState(1), warm_models=2, save_pcd=mode)
        rows = evalio.read_results_csv(str(tmp_path / f"res{int(mode)}" / "results.csv"))
        assert [r[0] for r in rows] == ids and res["models"] == 12 and res["graph"] is mode
        assert all(np.isfinite(r[1]) and np.isfinite(r[2]) and r[1] > 0 for r in rows)
        assert set(res["per_category"]) == {"02691156", "03001627"}
        assert abs(res["average_cd"] - np.mean([r[1] for r in rows])) < 1e-12
        assert res["average_time_s"] > 0
        out[mode] = (rows, res)
    # replaying the captured graph runs the same kernels on the same data: identical rows
    assert out[False][0] == out[True][0]
    saved = evalio.read_pcd(str(tmp_path / "res1" / "pcds" / "02691156" / "m00.pcd"))
    assert saved.shape == (16384, 3)
    print("batch-1 completion: eager %.3f ms, HIP graph %.3f ms" % (out[False][1]["average_time_s"] * 1e3,
                                                                   out[True][1]["average_time_s"] * 1e3))
    assert out[True][1]["average_time_s"] < out[False][1]["average_time_s"]
