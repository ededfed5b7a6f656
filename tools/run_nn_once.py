"""One Chamfer shape, a few calls: the target of rocprofv3 passes (tools/profile_nn.sh).
usage: python tools/run_nn_once.py <b> <n> <m> <mode> [iters] [kind]"""
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from rfnet_amd import _raw  # noqa: E402

b, n, m = (int(v) for v in sys.argv[1:4])
mode = sys.argv[4]
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
kind = sys.argv[6] if len(sys.argv) > 6 else "randn"
rng = np.random.RandomState(100)
gen = (lambda k: rng.randn(b, k, 3)) if kind == "randn" else (lambda k: rng.random_sample((b, k, 3)))
a = torch.from_numpy(gen(n).astype(np.float32)).cuda()
c = torch.from_numpy(gen(m).astype(np.float32)).cuda()
for _ in range(iters):
    out = _raw.nn_distance(a, c, mode=mode)
torch.cuda.synchronize()
print(float(out[0].sum()), float(out[2].sum()))
