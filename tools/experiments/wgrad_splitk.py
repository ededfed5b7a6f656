"""Weight-gradient GEMMs of the RFNet graph: x^T g with x (R, cin), g (R, cout), R = batch * points up
to 524288 and a tiny (cin, cout) output.  The library runs `x.t() @ g` on a handful of workgroups (one
per output tile, each looping over all R); splitting R into S batches of a bmm and summing the S
partial products fills the chip.  Prints ms for S = 1 (plain) .. 512 per shape."""
import sys, time
import torch


def bench(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


dev = "cuda"
for R in (524288, 96000, 32768):
    for cin, cout in ((3, 256), (16, 128), (64, 64), (128, 128), (128, 256), (256, 256), (384, 256), (512, 128)):
        x = torch.randn(R, cin, device=dev)
        g = torch.randn(R, cout, device=dev)
        ref = x.t() @ g
        line = [f"R={R:6d} {cin:3d}x{cout:3d}: plain {bench(lambda: x.t() @ g):7.3f}"]
        for S in (16, 32, 64, 128, 256, 512):
            if R % S or R // S < 256:
                continue
            f = lambda: torch.bmm(x.view(S, R // S, cin).transpose(1, 2), g.view(S, R // S, cout)).sum(0)
            err = float((f() - ref).abs().max() / ref.abs().max())
            line.append(f"S{S} {bench(f):6.3f}")
        line.append(f"(rel err {err:.1e})")
        print("  ".join(line), flush=True)
