"""Experiment: the bench step (nn_distance + nn_distance_grad, 3 launches) replayed from a captured
HIP graph vs launched eagerly.  python tools/experiments/graph_step.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from rfnet_amd._raw import nn_distance, nn_distance_grad
dev = torch.device("cuda")
rng = np.random.RandomState(100)
B, N, M = 32, 2048, 16384
xyz1 = torch.from_numpy(rng.randn(B, N, 3).astype(np.float32)).to(dev)
xyz2 = torch.from_numpy(rng.randn(B, M, 3).astype(np.float32)).to(dev)
gd1, gd2 = torch.ones(B, N, device=dev), torch.ones(B, M, device=dev)
def step():
    d1, i1, d2, i2 = nn_distance(xyz1, xyz2)
    g1, g2 = nn_distance_grad(xyz1, xyz2, gd1, i1, gd2, i2)
    return d1, i1, d2, i2, g1, g2
def timeit(fn, k=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / k * 1e3
print("eager   %.4f ms/step" % timeit(step))
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    outs = step()
ref = step()
g.replay(); torch.cuda.synchronize()
print("graph == eager:", all(torch.equal(a, b) for a, b in zip(outs[:4], ref[:4])), float((outs[4] - ref[4]).abs().max()))
print("graph   %.4f ms/step" % timeit(g.replay))
print("eager   %.4f ms/step" % timeit(step))
