"""C5 forward + loss with the ground-truth preparation (FPS 16384->1024 + sorts) on a side stream
underneath the network forward, vs in line on the same stream."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from rfnet_amd import glue
from rfnet_amd.rfnet import GroundTruth, RFNet
rng = np.random.RandomState(500)
partial = torch.from_numpy((rng.rand(32, 3000, 3) - 0.5).astype(np.float32)).cuda()
gt = torch.from_numpy((rng.rand(32, 16384, 3) - 0.5).astype(np.float32)).cuda()
torch.manual_seed(0)
net = RFNet().cuda()
def step(overlap, where):
    with torch.no_grad():
        if where == "before":
            g = GroundTruth(gt, 64, 1024, overlap=overlap)
            outs = net(partial)
        else:
            outs = net(partial)
            g = GroundTruth(gt, 64, 1024, overlap=overlap)
        g.join()
        p1, p2, p3, pf = outs
        cd = glue.chamfer_per_sample(gt, pf, sorted1=g.h_gt)[0].mean(1)
        e1 = glue.earth_mover_cost(g.gt1, p1) / 64.0
        e2 = glue.earth_mover_cost(g.gt2, p2) / 1024.0
        return torch.stack([cd, e1, e2], 1)
def timeit(fn, n=12):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rep in range(2):
    print("in line (before forward):", timeit(lambda: step(False, "before")))
    print("side stream, under the forward:", timeit(lambda: step(True, "before")))
    print("in line (after forward):", timeit(lambda: step(False, "after")))
def fwd():
    with torch.no_grad(): return net(partial)
print("network forward alone:", timeit(fwd))
print("GroundTruth alone:", timeit(lambda: GroundTruth(gt, 64, 1024, overlap=False)))
