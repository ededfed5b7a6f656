"""A graph holding GEMM -> reduce goes wrong after an eager call in between replays: which eager op, which BLAS?"""
import os, sys
import numpy as np, torch
rng = np.random.RandomState(0)
def t(*s): return torch.from_numpy((rng.rand(*s) - 0.5).astype(np.float32)).cuda()
w0 = t(3, 64)
def run(tag, interleave, blas=None):
    if blas: torch.backends.cuda.preferred_blas_library(blas)
    fn = lambda x: (x @ w0).amax(1, keepdim=True)
    static = t(1, 3000, 3)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.no_grad(): fn(static)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(g): out = fn(static)
    res = []
    for rep in range(4):
        b = t(1, 3000, 3); static.copy_(b); g.replay(); torch.cuda.synchronize()
        got = out.clone()
        with torch.no_grad():
            ref = fn(b) if interleave == "same" else None
            if interleave == "matmul": _ = b @ w0
            if interleave == "amax": _ = t(1, 3000, 64).amax(1, keepdim=True)
            if interleave == "alloc": _ = torch.empty(1 << 20, device="cuda").zero_()
            torch.cuda.synchronize()
            if ref is None:
                ref = (b.double() @ w0.double()).amax(1, keepdim=True).float()
        res.append(bool(torch.allclose(got, ref, rtol=1e-4, atol=1e-5)))
    print(f"{tag:40s} {res}", flush=True)
    return g, out  # keep the graph alive
keep = []
keep.append(run("interleave: nothing", "none"))
keep.append(run("interleave: big alloc+memset", "alloc"))
keep.append(run("interleave: eager matmul", "matmul"))
keep.append(run("interleave: eager amax", "amax"))
keep.append(run("interleave: eager same fn", "same"))
keep.append(run("rocBLAS ('cublas'), same fn", "same", "cublas"))
keep.append(run("hipblaslt, same fn", "same", "cublaslt"))
