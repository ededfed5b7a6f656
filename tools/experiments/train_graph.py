"""The RFNet training step (forward + loss block + backward) captured into ONE HIP graph: is it right,
and what does it save?  Gradients of 4 replays on fresh inputs are compared with eager runs on the same
inputs -- once with the eager runs AFTER all replays, once INTERLEAVED (an eager GEMM -> reduction
between replays is what broke a captured forward on this torch / ROCm build)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from rfnet_amd.rfnet import RFNet, training_loss

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
rng = np.random.RandomState(0)
torch.manual_seed(0)
net = RFNet().cuda()


def fresh():
    return (torch.from_numpy((rng.rand(B, 3000, 3) - 0.5).astype(np.float32)).cuda(),
            torch.from_numpy((rng.rand(B, 16384, 3) - 0.5).astype(np.float32)).cuda())


def step(partial, gt):
    collect = {}
    outs = net(partial, collect=collect)
    loss = training_loss(net, outs, collect, gt, 0.01)
    loss.backward()
    return loss


def grads():
    return [p.grad.clone() for p in net.parameters() if p.grad is not None]


sp, sg = fresh()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        net.zero_grad(set_to_none=True)
        step(sp, sg)
torch.cuda.current_stream().wait_stream(side)
net.zero_grad(set_to_none=True)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    sloss = step(sp, sg)
static_grads = [p.grad for p in net.parameters() if p.grad is not None]
print("captured; params with grad:", len(static_grads), flush=True)


def compare(tag, a, b):
    worst = 0.0
    for x, y in zip(a, b):
        den = float(y.abs().max()) + 1e-12
        worst = max(worst, float((x - y).abs().max()) / den)
    print(f"{tag}: worst max-abs difference relative to the gradient's max {worst:.2e}", flush=True)
    return worst


for mode in ("eager after all replays", "eager interleaved"):
    inputs = [fresh() for _ in range(4)]
    got = []
    for i, (p, g) in enumerate(inputs):
        sp.copy_(p); sg.copy_(g)
        graph.replay()
        torch.cuda.synchronize()
        got.append(([x.clone() for x in static_grads], float(sloss)))
        if mode == "eager interleaved":
            keep = [x.clone() for x in static_grads]
            for q in net.parameters():
                q.grad = None
            el = step(p, g)
            compare(f"  [{mode}] replay {i} (loss {got[-1][1]:.6f} vs {float(el):.6f})", got[-1][0], grads())
            for q, k in zip([q for q in net.parameters() if q.grad is not None], static_grads):
                q.grad = k  # hand the static buffers back
    if mode == "eager after all replays":
        for i, (p, g) in enumerate(inputs):
            for q in net.parameters():
                q.grad = None
            el = step(p, g)
            compare(f"  [{mode}] replay {i} (loss {got[i][1]:.6f} vs {float(el):.6f})", got[i][0], grads())
        for q, k in zip([q for q in net.parameters() if q.grad is not None], static_grads):
            q.grad = k

torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    graph.replay()
torch.cuda.synchronize()
print(f"graph replay: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per training step (B={B})")
for q in net.parameters():
    q.grad = None
p, g = fresh()
for _ in range(2):
    net.zero_grad(set_to_none=True); step(p, g)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    net.zero_grad(set_to_none=True); step(p, g)
torch.cuda.synchronize()
print(f"eager:        {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per training step")
