"""ONE Chamfer step (fwd + bwd, C2, B = 32) with the batch cut into P parts that run on P side streams
(forked from and joined to the caller's stream by events): does intra-step concurrency pay like
pipelining independent steps does (two_stream_steps.py)?  Also as a captured HIP graph."""
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


def make(parts):
    bs = B // parts
    plans = [ChamferStep(bs, N, M, "cuda") for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    sl = [slice(i * bs, (i + 1) * bs) for i in range(parts)]
    ins = [(x1[s], x2[s], g1[s], g2[s]) for s in sl]

    def step():
        cur = torch.cuda.current_stream()
        if parts == 1:
            plans[0](*ins[0])
            return
        for st, pl, a in zip(streams, plans, ins):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                pl(*a)
        for st in streams:
            cur.wait_stream(st)
    return step, plans


for parts in (1, 2, 4, 1, 2, 4):
    step, plans = make(parts)
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    K = 300
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    got = [torch.cat([getattr(p, nm) for p in plans]) for nm in ("dist1", "idx1", "dist2", "idx2")]
    ok = all(torch.equal(a, b) for a, b in zip(got, ref[:4]))
    line = f"{parts} part(s): eager {dt / K * 1e3:.4f} ms per step ({B * N * M * K / dt:.3e} pairs/s), identical {ok}"
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            g.replay()
        torch.cuda.synchronize()
        dg = time.perf_counter() - t0
        got = [torch.cat([getattr(p, nm) for p in plans]) for nm in ("dist1", "idx1", "dist2", "idx2")]
        line += f"; HIP graph {dg / K * 1e3:.4f} ms, identical {all(torch.equal(a, b) for a, b in zip(got, ref[:4]))}"
    except Exception as exc:  # noqa: BLE001
        line += f"; graph failed: {type(exc).__name__}"
    print(line, flush=True)
