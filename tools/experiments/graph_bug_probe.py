"""Which half of `(x @ w).amax(1)` goes wrong when replayed from a captured graph (torch 2.10 / ROCm 7)?"""
import sys
import numpy as np, torch
rng = np.random.RandomState(0)
def t(*s): return torch.from_numpy((rng.rand(*s) - 0.5).astype(np.float32)).cuda()
w0 = t(3, 64)
w1 = t(64, 64)

def run(tag, fn, shape, ref_fn=None):
    static = t(*shape)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.no_grad():
        fn(static)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(g):
        out = fn(static)
    res = []
    prev = None
    for rep in range(4):
        b = t(*shape); static.copy_(b); g.replay(); torch.cuda.synchronize()
        got = out.clone()
        with torch.no_grad():
            ref = (ref_fn or fn)(b)
        ok = bool(torch.allclose(got, ref, rtol=1e-4, atol=1e-5))
        stale = prev is not None and bool(torch.equal(got, prev))
        res.append("ok" if ok else ("STALE" if stale else f"BAD(maxdiff {float((got-ref).abs().max()):.2e}, nan {int(torch.isnan(got).sum())})"))
        prev = got
    print(f"{tag:52s} {res}", flush=True)
    return g, out

keep = []
keep.append(run("matmul only (1,3000,3)@(3,64)", lambda x: x @ w0, (1, 3000, 3)))
keep.append(run("amax only (1,3000,64)", lambda x: x.amax(1, keepdim=True), (1, 3000, 64)))
keep.append(run("matmul -> amax", lambda x: (x @ w0).amax(1, keepdim=True), (1, 3000, 3)))
keep.append(run("matmul -> sum", lambda x: (x @ w0).sum(1, keepdim=True), (1, 3000, 3)))
keep.append(run("matmul -> relu (elementwise)", lambda x: torch.relu(x @ w0), (1, 3000, 3)))
keep.append(run("matmul -> clone -> amax", lambda x: (x @ w0).clone().amax(1, keepdim=True), (1, 3000, 3)))
keep.append(run("2-D mm -> amax(0)", lambda x: (x @ w0).amax(0, keepdim=True), (3000, 3)))
keep.append(run("mm K=64 -> amax", lambda x: (x @ w1).amax(1, keepdim=True), (1, 3000, 64)))
keep.append(run("relu -> amax (no gemm)", lambda x: torch.relu(x).amax(1, keepdim=True), (1, 3000, 64)))
keep.append(run("bmm -> sum(0)", lambda x: torch.bmm(x.transpose(1, 2), x).sum(0), (16, 512, 64)))
keep.append(run("mm -> mul -> sum(all)", lambda x: ((x @ w1) * 2.0).sum(), (3000, 64)))
