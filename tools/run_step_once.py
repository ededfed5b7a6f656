"""rf_chamfer_step at one shape, a few calls: the target of rocprofv3 passes.
usage: python tools/run_step_once.py [b n m [iters]]"""
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from rfnet_amd import _raw  # noqa: E402

b, n, m = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 2048, 16384)
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
rng = np.random.RandomState(100)
a = torch.from_numpy(rng.randn(b, n, 3).astype(np.float32)).cuda()
c = torch.from_numpy(rng.randn(b, m, 3).astype(np.float32)).cuda()
g1, g2 = torch.ones(b, n, device="cuda"), torch.ones(b, m, device="cuda")
plan = _raw.ChamferStep(b, n, m, "cuda")
for _ in range(iters):
    out = plan(a, c, g1, g2)
torch.cuda.synchronize()
print(float(out[0].sum()), float(out[4].abs().sum()))
