"""Does running the EMD level pipeline of two half-batches on two HIP streams (their launches fill each
other's ramps and tails) beat one full-batch chain?  C4: B=32, 2048 vs 2048."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from rfnet_amd import _raw as R
rng = np.random.RandomState(100)
u = torch.from_numpy((rng.random_sample((32, 2048, 3)) - 0.5).astype(np.float32)).cuda()
v = torch.from_numpy((rng.random_sample((32, 2048, 3)) - 0.5).astype(np.float32)).cuda()
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("one chain, B=32:", timeit(lambda: R.approx_match(u, v)))
print("fused earth_mover B=32:", timeit(lambda: R.earth_mover(u, v)))
for parts in (2, 4):
    streams = [torch.cuda.Stream() for _ in range(parts)]
    chunks = [(u[i::parts].contiguous(), v[i::parts].contiguous()) for i in range(parts)]
    def run():
        cur = torch.cuda.current_stream()
        ev = torch.cuda.Event(); ev.record(cur)
        outs = []
        for s, (a, c) in zip(streams, chunks):
            s.wait_event(ev)
            with torch.cuda.stream(s):
                outs.append(R.earth_mover(a, c))
        for s in streams:
            cur.wait_stream(s)
        return outs
    print(f"{parts} chains of B={32 // parts} on {parts} streams (earth_mover):", timeit(run))
    def run2():
        cur = torch.cuda.current_stream()
        outs = []
        for s, (a, c) in zip(streams, chunks):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                outs.append(R.approx_match(a, c))
        for s in streams:
            cur.wait_stream(s)
        return outs
    print(f"{parts} chains (approx_match):", timeit(run2))
