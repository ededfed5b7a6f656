#!/usr/bin/env python3
"""What would rf_chamfer_step gain from running the batch as K independent sub-batches on K streams (fork / join by events)?
The step's three kernels each end in a tail at falling occupancy (tools/experiments/wave_timeline.py); sub-batches on separate
hardware queues fill each other's tails.  Here from Python with K plans of B / K clouds; inputs are slices of the same tensors.
usage: python tools/experiments/split_streams.py [B N M]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from rfnet_amd import _raw as R  # noqa: E402

B, N, M = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 2048, 16384)
rng = np.random.RandomState(100)
a = torch.from_numpy(rng.randn(B, N, 3).astype(np.float32)).cuda()
c = torch.from_numpy(rng.randn(B, M, 3).astype(np.float32)).cuda()
g1 = torch.ones(B, N, device="cuda")
g2 = torch.ones(B, M, device="cuda")
main = torch.cuda.current_stream()
ref = R.ChamferStep(B, N, M, "cuda")(a, c, g1, g2)
ref = [t.clone() for t in ref]
for K in (1, 2, 4):
    bs = B // K
    plans = [R.ChamferStep(bs, N, M, "cuda") for _ in range(K)]
    streams = [main] + [torch.cuda.Stream() for _ in range(K - 1)]
    sl = [slice(i * bs, (i + 1) * bs) for i in range(K)]
    ins = [(a[s], c[s], g1[s], g2[s]) for s in sl]
    fork = torch.cuda.Event()
    joins = [torch.cuda.Event() for _ in range(K - 1)]

    def step():
        if K > 1:
            fork.record(main)
        for i in range(1, K):
            streams[i].wait_event(fork)
            with torch.cuda.stream(streams[i]):
                plans[i](*ins[i])
                joins[i - 1].record(streams[i])
        plans[0](*ins[0])
        for e in joins:
            main.wait_event(e)

    for _ in range(20):
        step()
    torch.cuda.synchronize()
    outs = [torch.cat([p.dist1 for p in plans]), torch.cat([p.idx1 for p in plans]), torch.cat([p.dist2 for p in plans]),
            torch.cat([p.idx2 for p in plans])]
    ok = all(torch.equal(x, y) for x, y in zip(outs, ref[:4]))
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(200):
            step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 200 * 1e6)
    print(f"K={K}: {best:.1f} us per whole step (forward outputs equal to the single launch: {ok})")
