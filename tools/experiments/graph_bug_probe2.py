"""Replays of a captured `(x @ w).amax(1)` return the FIRST replay's result (stale) on this torch / ROCm
build -- under which conditions?  One mode per process: python graph_bug_probe2.py <mode>"""
import sys
import numpy as np, torch
rng = np.random.RandomState(0)
def t(*s): return torch.from_numpy((rng.rand(*s) - 0.5).astype(np.float32)).cuda()
w0 = t(3, 64)
w1 = t(64, 64)
mode = sys.argv[1]

def capture(fn, shape):
    static = t(*shape)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.no_grad(): fn(static)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(g): out = fn(static)
    return g, static, out

def check(tag, fn, shape, reps=4):
    g, static, out = capture(fn, shape)
    prev, res = None, []
    for rep in range(reps):
        b = t(*shape); static.copy_(b); g.replay(); torch.cuda.synchronize()
        got = out.clone()
        with torch.no_grad(): ref = fn(b)
        torch.cuda.synchronize()
        ok = bool(torch.allclose(got, ref, rtol=1e-4, atol=1e-5))
        res.append("ok" if ok else ("STALE" if prev is not None and torch.equal(got, prev) else "BAD"))
        prev = got
    print(f"{mode:12s} {tag:28s} {res}", flush=True)
    return g, static, out

keep = []
base = lambda x: (x @ w0).amax(1, keepdim=True)
if mode == "base":
    keep.append(check("mm->amax", base, (1, 3000, 3)))
elif mode == "mmfirst":
    keep.append(check("mm only", lambda x: x @ w0, (1, 3000, 3)))
    keep.append(check("mm->amax", base, (1, 3000, 3)))
elif mode == "amaxfirst":
    keep.append(check("amax only", lambda x: x.amax(1, keepdim=True), (1, 3000, 64)))
    keep.append(check("mm->amax", base, (1, 3000, 3)))
elif mode == "twice":
    keep.append(check("mm->amax #1", base, (1, 3000, 3)))
    keep.append(check("mm->amax #2", base, (1, 3000, 3)))
elif mode == "k64":
    keep.append(check("mm K=64 ->amax", lambda x: (x @ w1).amax(1, keepdim=True), (1, 3000, 64)))
elif mode == "2d":
    keep.append(check("2-D mm->amax(0)", lambda x: (x @ w0).amax(0, keepdim=True), (3000, 3)))
elif mode == "relu":
    keep.append(check("mm->relu", lambda x: torch.relu(x @ w0), (1, 3000, 3)))
elif mode == "mmonly":
    keep.append(check("mm only", lambda x: x @ w0, (1, 3000, 3)))
elif mode == "amaxonly":
    keep.append(check("amax only", lambda x: x.amax(1, keepdim=True), (1, 3000, 64)))
elif mode == "sum":
    keep.append(check("mm->sum", lambda x: (x @ w0).sum(1, keepdim=True), (1, 3000, 3)))
