"""Batch-1 completion latency (recon_test.py's timed sess.run): eager vs the forward captured into one HIP graph."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from rfnet_amd.evalrun import GraphedForward
from rfnet_amd.rfnet import RFNet
torch.manual_seed(0)
net = RFNet().cuda().eval()
rng = np.random.RandomState(0)
for B in (1, 4, 32):
    x = torch.from_numpy((rng.rand(B, 3000, 3) - 0.5).astype(np.float32)).cuda()
    def timeit(fn, n=30):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            fn(); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    with torch.no_grad():
        e = timeit(lambda: net(x))
        g = GraphedForward(net, x)
        r = timeit(lambda: g(x))
        ref = net(x); got = g(x)
        same = all(torch.equal(a, b) for a, b in zip(ref, got))
    print(f"B={B}: eager {e:.3f} ms, HIP graph {r:.3f} ms, identical outputs {same}")
