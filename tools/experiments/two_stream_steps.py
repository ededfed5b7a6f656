"""Independent Chamfer steps (fwd + bwd, C2) issued round-robin on 1 / 2 / 3 streams, each stream with its
own ChamferStep plan (outputs + workspace): how much of the step's serial structure (a 96-workgroup sort,
kernel boundaries) does a second independent step fill?  Throughput only -- the latency of ONE step does
not change."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from rfnet_amd._raw import ChamferStep

B, N, M = 32, 2048, 16384
rng = np.random.RandomState(100)
x1 = torch.from_numpy(rng.randn(B, N, 3).astype(np.float32)).cuda()
x2 = torch.from_numpy(rng.randn(B, M, 3).astype(np.float32)).cuda()
g1, g2 = torch.ones(B, N, device="cuda"), torch.ones(B, M, device="cuda")
ref = [t.clone() for t in ChamferStep(B, N, M, "cuda")(x1, x2, g1, g2)]
for ns in (1, 2, 3, 1, 2, 3):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    plans = [ChamferStep(B, N, M, "cuda") for _ in range(ns)]
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())

    def run(k):
        for i in range(k):
            with torch.cuda.stream(streams[i % ns]):
                plans[i % ns](x1, x2, g1, g2)

    run(20)
    torch.cuda.synchronize()
    K = 300
    t0 = time.perf_counter()
    run(K)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = all(torch.equal(a, b) for p in plans for a, b in zip((p.dist1, p.idx1, p.dist2, p.idx2), ref[:4]))
    print(f"{ns} stream(s): {dt / K * 1e3:.4f} ms per step, {B * N * M * K / dt:.3e} pairs/s, outputs identical {ok}", flush=True)
