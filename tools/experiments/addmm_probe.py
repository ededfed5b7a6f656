"""Scratch timing probe (needs a GPU); run as a script, never imported or collected."""
import time

import torch


def main():
    x=torch.randn(32*16384,128,device='cuda'); w=torch.randn(128,128,device='cuda'); b=torch.randn(128,device='cuda')
    def t(fn,n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
    print("linear+relu", t(lambda: torch.relu(torch.nn.functional.linear(x,w.t(),b))))
    print("linear only", t(lambda: torch.nn.functional.linear(x,w.t(),b)))
    try:
        y=torch._addmm_activation(b,x,w)
        print("addmm_act", t(lambda: torch._addmm_activation(b,x,w)), torch.allclose(y, torch.relu(torch.addmm(b,x,w)),atol=1e-4))
    except Exception as e: print("addmm_activation failed", e)
    x3=x.view(32,16384,128)
    print("max over points", t(lambda: x3.max(1)))
    print("amax over points", t(lambda: x3.amax(1)))
    print("relu_ inplace", t(lambda: torch.relu_(torch.nn.functional.linear(x,w.t(),b))))


if __name__ == "__main__":
    main()
