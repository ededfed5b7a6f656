"""C5-sized RFNet training step (forward + loss + backward, no optimizer), a few iterations: the
target of a rocprofv3 --kernel-trace --stats pass.  argv: iters [full]  (full = vv_recon.py's whole
training loss through rfnet.training_loss instead of the three C5 terms)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from rfnet_amd import glue
from rfnet_amd.rfnet import GroundTruth, RFNet, training_loss
rng = np.random.RandomState(100)
torch.manual_seed(0)
net = RFNet().cuda()
partial = torch.from_numpy((rng.rand(32, 3000, 3) - 0.5).astype(np.float32)).cuda()
gt = torch.from_numpy((rng.rand(32, 16384, 3) - 0.5).astype(np.float32)).cuda()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
full = len(sys.argv) > 2 and sys.argv[2] == "full"


def step():
    net.zero_grad(set_to_none=True)
    if full:
        collect = {}
        outs = net(partial, collect=collect)
        loss = training_loss(net, outs, collect, gt, 0.01)
    else:
        p1, p2, p3, pf = net(partial)
        g = GroundTruth(gt, 64, 1024, overlap=False).join()
        loss = (glue.chamfer_big(gt, pf, sorted1=g.h_gt)[0] + glue.earth_mover(g.gt1, p1) + glue.earth_mover(g.gt2, p2))
    loss.backward()
    return loss


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    loss = step()
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / iters * 1e3:.3f} ms per training step (full={full}), loss {float(loss):.6f}")
