"""C5 forward + loss, a few iterations: the target of a rocprofv3 --kernel-trace --stats pass."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from rfnet_amd import glue
from rfnet_amd.rfnet import RFNet
rng = np.random.RandomState(100)
torch.manual_seed(0)
net = RFNet().cuda()
partial = torch.from_numpy((rng.rand(32, 3000, 3) - 0.5).astype(np.float32)).cuda()
gt = torch.from_numpy((rng.rand(32, 16384, 3) - 0.5).astype(np.float32)).cuda()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for _ in range(iters):
    with torch.no_grad():
        p1, p2, p3, pf = net(partial)
        gt64, gt1024 = glue.sampling(64, gt)[1], glue.sampling(1024, gt)[1]
        loss = glue.chamfer_big(pf, gt)[0] + glue.earth_mover(p1, gt64) + glue.earth_mover(p2, gt1024)
torch.cuda.synchronize()
print(float(loss))
