"""When do the clouds of ONE culled sweep launch finish?  (round 4, the sweep -> backward seam: a per-cloud hand-off inside the
launch only pays if clouds finish well apart.)  Needs the instrumented build: python tools/build_variant.py cloudend -DRFP_CLOUD_END=1,
then RFOPS_LIB=rfnet_amd/variants/librfops_cloudend.so python tools/experiments/cloud_end_times.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from rfnet_amd import _raw as R
rng = np.random.RandomState(100)
a = torch.from_numpy(rng.randn(32, 2048, 3).astype(np.float32)).cuda()
c = torch.from_numpy(rng.randn(32, 16384, 3).astype(np.float32)).cuda()
for rep in range(4):
    st = []
    R.nn_distance(a, c, mode="culled", stats=st)
    t = np.array(st[:32], dtype=np.float64) * 0.01  # 100 MHz ticks -> us
    t = t.max() - t
    o = np.sort(t)[::-1]
    print("run %d: cloud finished this many us BEFORE the launch's last workgroup (sorted): " % rep + " ".join("%.1f" % v for v in o))
    print("        median %.1f us, clouds finishing more than 10 us early: %d of 32" % (np.median(t), int((t > 10).sum())))
