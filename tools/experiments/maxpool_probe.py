"""Scratch timing probe (needs a GPU); run as a script, never imported or collected."""
import time

import torch


def main():
    x=torch.randn(32,16384,128,device='cuda').relu_().requires_grad_(True)
    def t(fn,n=10):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
    def fb(f):
        def g():
            x.grad=None
            f(x).sum().backward()
        return g
    print("max  fwd", t(lambda: x.max(1).values), "fwd+bwd", t(fb(lambda v: v.max(1).values)))
    print("amax fwd", t(lambda: x.amax(1)), "fwd+bwd", t(fb(lambda v: v.amax(1))))
    y=torch.randn(32*16384,128,device='cuda'); gr=torch.randn_like(y)
    print("threshold_backward", t(lambda: torch.threshold_backward(gr,y,0.0)), "mul mask", t(lambda: gr*(y>0).to(gr.dtype)))


if __name__ == "__main__":
    main()
