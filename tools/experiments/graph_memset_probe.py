"""Is a hipMemsetAsync captured into a HIP graph executed on every replay?  graph = [memset(buf, 0);
buf += 1]: buf must read 1 after every replay.  (torch's reduce kernels zero their semaphores with
cudaMemsetAsync: a memset node that does not run on replay leaves the reduction's output unwritten.)"""
import ctypes, os, sys
import torch
libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
hip = ctypes.CDLL(os.path.join(libdir, "libamdhip64.so"))
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int
hip.hipMemsetD32Async.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetD32Async.restype = ctypes.c_int

def run(tag, nfloats, use_d32=False):
    buf = torch.full((nfloats,), 5.0, device="cuda")
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    def body():
        st = torch.cuda.current_stream().cuda_stream
        if use_d32:
            rc = hip.hipMemsetD32Async(buf.data_ptr(), 0, nfloats, st)
        else:
            rc = hip.hipMemsetAsync(buf.data_ptr(), 0, nfloats * 4, st)
        assert rc == 0, rc
        buf.add_(1.0)
    with torch.cuda.stream(s):
        body()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    res = []
    for rep in range(4):
        g.replay(); torch.cuda.synchronize()
        res.append((float(buf.min()), float(buf.max())))
    print(f"{tag:40s} {res}", flush=True)
    return g

keep = [run("memset 16 B", 4), run("memset 256 B", 64), run("memset 4 KB", 1024), run("memset 1 MB", 1 << 18),
        run("memsetD32 256 B", 64, True), run("memsetD32 1 MB", 1 << 18, True)]
# and torch's own zero_ (fill kernel or memset?)
buf = torch.full((64,), 5.0, device="cuda")
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    buf.zero_(); buf.add_(1.0)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    buf.zero_(); buf.add_(1.0)
res = []
for rep in range(4):
    g.replay(); torch.cuda.synchronize(); res.append(float(buf.max()))
print("torch zero_ + add_", res)
